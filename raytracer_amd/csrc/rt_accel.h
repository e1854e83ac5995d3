// rt_accel.h — host-side builder of the library's own acceleration data (used by rt_scene_create).
//
// The reference keeps every shape that straddles a split plane in the INNER group of its
// `divide` tree (groups.rs:198-299), so a ray that enters a group tests all of that group's own
// shapes linearly (groups.rs:118-120).  For the ~871 k-triangle dragon that is thousands of
// Möller–Trumbore tests per ray.  The device scene keeps the reference tree verbatim (its box
// tests decide what the reference would have tested — SURVEY.md Q6/Q7), and hangs one *auxiliary*
// BVH under every group whose own shapes are many identity-transform triangles.  The auxiliary
// BVH only skips triangles whose test could not have succeeded, so the set of intersections found
// — and with it min(t, rank) and the n1/n2 list — is unchanged (DESIGN.md "Auxiliary BVH":
// padding rule, per-ray guard, and the ACCEL == NO_ACCEL full-frame test).
//
// Layout (device, 64 B per node, 16 B aligned):
//   aux[root]      header: {cx, cy, cz, radius} {lmax, amax, 0, 0} {unused} {unused}
//                  (centre/radius of the padded boxes, longest edge, largest |e1|*|e2|)
//   aux[root + 1]  top node; every node holds the boxes of BOTH children:
//     {lmin.xyz, left}  {lmax.xyz, lcount}  {rmin.xyz, right}  {rmax.xyz, rcount}
//     count == 0: child is the node aux[left]; count > 0: child is the leaf
//     aux_tris[left .. left+count), 3 float4 per triangle: {p1.xyz, k} {e1.xyz, 0} {e2.xyz, 0}
//     where k (uint bits) is the triangle's offset inside its group: shape = first_shape + k.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rt_abi.h"

namespace rtaccel {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMinTris = 12;    // groups with fewer own triangles keep the linear loop
constexpr uint32_t kLeafTris = 4;    // max triangles per leaf
constexpr int kMaxDepth = 22;        // device stack holds kMaxDepth + 2 entries
constexpr float kPadRel = 0.01f;     // static box padding relative to the triangle's largest extent
constexpr float kPadAbs = 4e-6f;     // ... plus this times the largest |coordinate|

struct Node {
  float lmin[3];
  uint32_t left;
  float lmax[3];
  uint32_t lcount;
  float rmin[3];
  uint32_t right;
  float rmax[3];
  uint32_t rcount;
};
static_assert(sizeof(Node) == 64, "aux node is 64 B");
struct Tri {
  float p1[3];
  uint32_t k;
  float e1[3], pad1;
  float e2[3], pad2;
};
static_assert(sizeof(Tri) == 48, "aux triangle is 48 B");

struct Result {
  std::vector<Node> nodes;       // all trees back to back
  std::vector<Tri> tris;         // leaf triangles in leaf order
  std::vector<uint32_t> root_of; // per reference node: header index or kNone
};

struct Item {
  float lo[3], hi[3], c[3];
  uint32_t k;
};

struct TreeBuilder {
  const RtTriGeom* geom;  // geometry of the group's triangles (first_tri based)
  std::vector<Item> items;
  std::vector<Node> nodes;  // local; [0] header, [1] top
  std::vector<Tri> tris;
  float lmax = 0.f, amax = 0.f;
  int depth_seen = 0;

  static void merge(float* lo, float* hi, const Item& it) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], it.lo[a]);
      hi[a] = std::max(hi[a], it.hi[a]);
    }
  }
  // returns child reference for [b,e): (index, count); writes the box
  void build(uint32_t b, uint32_t e, int depth, uint32_t* ref, uint32_t* count, float* lo, float* hi) {
    depth_seen = std::max(depth_seen, depth);
    for (int a = 0; a < 3; a++) lo[a] = INFINITY, hi[a] = -INFINITY;
    for (uint32_t i = b; i < e; i++) merge(lo, hi, items[i]);
    uint32_t n = e - b;
    if (n <= kLeafTris) {
      *ref = (uint32_t)tris.size();
      *count = n;
      for (uint32_t i = b; i < e; i++) {
        const RtTriGeom& g = geom[items[i].k];
        Tri t{};
        std::memcpy(t.p1, g.p1, 12);
        std::memcpy(t.e1, g.e1, 12);
        std::memcpy(t.e2, g.e2, 12);
        t.k = items[i].k;
        tris.push_back(t);
      }
      return;
    }
    // object-median split on the axis of largest centroid extent: balanced, depth = ceil(log2(n/leaf))
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = b; i < e; i++)
      for (int a = 0; a < 3; a++) {
        clo[a] = std::min(clo[a], items[i].c[a]);
        chi[a] = std::max(chi[a], items[i].c[a]);
      }
    int axis = 0;
    for (int a = 1; a < 3; a++)
      if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
    uint32_t mid = b + n / 2;
    std::nth_element(items.begin() + b, items.begin() + mid, items.begin() + e, [axis](const Item& x, const Item& y) {
      return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.k < y.k);
    });
    uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    *ref = me;
    *count = 0;
    Node nd{};
    build(b, mid, depth + 1, &nd.left, &nd.lcount, nd.lmin, nd.lmax);
    build(mid, e, depth + 1, &nd.right, &nd.rcount, nd.rmin, nd.rmax);
    nodes[me] = nd;
  }

  // false: this group keeps the linear loop
  bool run(const RtTriGeom* g, uint32_t n) {
    geom = g;
    items.resize(n);
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k = 0; k < n; k++) {
      Item& it = items[k];
      it.k = k;
      float ext = 0.f, mag = 0.f, edge = 0.f;
      for (int a = 0; a < 3; a++) {
        float p1 = g[k].p1[a], p2 = p1 + g[k].e1[a], p3 = p1 + g[k].e2[a];
        if (!std::isfinite(p1) || !std::isfinite(p2) || !std::isfinite(p3)) return false;
        it.lo[a] = std::min(p1, std::min(p2, p3));
        it.hi[a] = std::max(p1, std::max(p2, p3));
        ext = std::max(ext, it.hi[a] - it.lo[a]);
        mag = std::max(mag, std::max(std::fabs(it.lo[a]), std::fabs(it.hi[a])));
      }
      float l1 = 0.f, l2 = 0.f, l3 = 0.f;
      for (int a = 0; a < 3; a++) {
        float d = g[k].e2[a] - g[k].e1[a];
        l1 += g[k].e1[a] * g[k].e1[a], l2 += g[k].e2[a] * g[k].e2[a], l3 += d * d;
      }
      edge = std::sqrt(std::max(l1, std::max(l2, l3)));
      lmax = std::max(lmax, edge);
      amax = std::max(amax, std::sqrt(l1) * std::sqrt(l2) * 1.000001f);
      float pad = kPadRel * std::max(ext, edge) + kPadAbs * mag + 1e-30f;
      for (int a = 0; a < 3; a++) {
        it.lo[a] -= pad;
        it.hi[a] += pad;
        it.c[a] = 0.5f * (it.lo[a] + it.hi[a]);
        blo[a] = std::min(blo[a], it.lo[a]);
        bhi[a] = std::max(bhi[a], it.hi[a]);
      }
    }
    nodes.clear();
    tris.clear();
    tris.reserve(n);
    nodes.emplace_back();  // header
    uint32_t ref, count;
    float lo[3], hi[3];
    build(0, n, 1, &ref, &count, lo, hi);
    if (count != 0 || ref != 1 || depth_seen > kMaxDepth) return false;
    Node& h = nodes[0];
    std::memset(&h, 0, sizeof(h));
    float r2 = 0.f;
    for (int a = 0; a < 3; a++) {
      h.lmin[a] = 0.5f * (blo[a] + bhi[a]);
      float half = 0.5f * (bhi[a] - blo[a]);
      r2 += half * half;
    }
    float radius = std::sqrt(r2) * 1.0001f;
    std::memcpy(&h.left, &radius, 4);
    h.lmax[0] = lmax;
    h.lmax[1] = amax;
    return std::isfinite(radius) && std::isfinite(lmax) && std::isfinite(amax);
  }
};

inline Result build(const RtScene* s) {
  Result r;
  r.root_of.assign(s->n_nodes, kNone);
  std::vector<uint32_t> todo;
  for (uint32_t i = 0; i < s->n_nodes; i++) {
    const RtNode& n = s->nodes[i];
    if ((n.flags & RT_NODE_ALL_TRIS) && n.n_shapes >= kMinTris) todo.push_back(i);
  }
  if (todo.empty()) return r;
  // build the trees in parallel, then concatenate with rebased indices
  struct Out {
    std::vector<Node> nodes;
    std::vector<Tri> tris;
    bool ok = false;
  };
  std::vector<Out> outs(todo.size());
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (todo.size() < 64) nt = 1;
  auto work = [&](unsigned t) {
    for (size_t j = t; j < todo.size(); j += nt) {
      const RtNode& n = s->nodes[todo[j]];
      TreeBuilder tb;
      outs[j].ok = tb.run(s->tri_geom + n.first_tri, n.n_shapes);
      if (outs[j].ok) {
        outs[j].nodes = std::move(tb.nodes);
        outs[j].tris = std::move(tb.tris);
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for (auto& t : th) t.join();
  for (size_t j = 0; j < todo.size(); j++) {
    if (!outs[j].ok) continue;
    uint64_t nb = r.nodes.size(), tbase = r.tris.size();
    if (nb + outs[j].nodes.size() >= 0x1FFFFFFFull || tbase + outs[j].tris.size() >= 0x1FFFFFFFull) break;
    r.root_of[todo[j]] = (uint32_t)nb;
    for (size_t q = 0; q < outs[j].nodes.size(); q++) {
      Node nd = outs[j].nodes[q];
      if (q > 0) {
        nd.left += (uint32_t)(nd.lcount ? tbase : nb);
        nd.right += (uint32_t)(nd.rcount ? tbase : nb);
      }
      r.nodes.push_back(nd);
    }
    r.tris.insert(r.tris.end(), outs[j].tris.begin(), outs[j].tris.end());
  }
  return r;
}

}  // namespace rtaccel
