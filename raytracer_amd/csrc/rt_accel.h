// rt_accel.h — host-side builder of the library's own acceleration data (used by rt_scene_create).
//
// The reference keeps every shape that straddles a split plane in the INNER group of its
// `divide` tree (groups.rs:198-299), so a ray that enters a group tests all of that group's own
// shapes linearly (groups.rs:118-120), and it re-tests one bounding box per tree level.  For the
// ~871 k-triangle dragon that is thousands of Möller–Trumbore tests per ray.
//
// The device scene keeps the reference tree verbatim, and for every maximal subtree that consists
// only of identity-transform groups holding identity-transform triangles (what `divide` makes of an
// OBJ group) it adds ONE auxiliary BVH over all triangles of the subtree.  A ray that reaches the
// subtree root R (R's own box test is still the reference's) traverses the auxiliary BVH instead of
// the subtree.  Two rules keep the result identical to the reference's (DESIGN.md "Auxiliary BVH"):
//   1. culling is conservative: boxes are widened per ray by more than the f32 slop of the
//      reference's triangle test, so only triangles that test would reject are skipped;
//   2. a triangle the test accepts still counts only if the reference would have REACHED it, i.e.
//      if the box test of its owning group G and of every group between G and R passes
//      (groups.rs:111-113).  Inside such a subtree boxes are unions of their contents
//      (groups.rs:176-188), so they nest, and for nested finite boxes "G passes" implies "every
//      ancestor passes" (monotonicity of the correctly rounded slab arithmetic, for rays with at
//      least one |d| >= 1e-4 component).  gate_next[] lists the groups whose test is NOT implied.
//
// Layout (device, 128 B per node, 16 B aligned): a 4-wide tree, made by folding every other level of the binary
// SAH tree into its parent (a ray's traversal is a chain of dependent fetches, one per level: half the levels, half
// the chain, and the per-step costs of the wave-cooperative traversal are paid half as often).
//   aux[root]      entry record of one instance: {cx, cy, cz, radius} {lmax, amax, top node index, first shape}
//                  (centre/radius of the padded boxes, longest edge, largest |e1|*|e2|)
//   aux[top]       top node of the (possibly shared) tree; every node holds the boxes of its 2-4 children and its own
//                  normal cone, as eight float4:
//     {min0.xyz, ref0} {max0.xyz, ref1} {min1.xyz, ref2} {max1.xyz, ref3}
//     {min2.xyz, a.x}  {max2.xyz, a.y}  {min3.xyz, a.z}  {max3.xyz, cos w}
//     cone: every triangle below the node has its unit normal within angle w of a/|a|, and
//     1/sin(angle between its edges e1, e2) <= 1/|a|; cos w < -1 means "no bound".  The traversal turns
//     that into a bound on K = |e1||e2||d|/|det| = 1/(sin(e1,e2) |cos(d, normal)|) for the node, which
//     sets how far its child boxes must be widened for this ray (rule 1).
//     ref = index << 4 | count.  count == 0: the child is the node aux[index]; count > 0: the child is the leaf
//     aux_tris[index .. index+count); 0xFFFFFFFF: no such child (its box is empty: min = +inf, max = -inf).
//     3 float4 per triangle:
//       {p1.xyz, shape index - first shape of the instance} {e1.xyz, owning node index - root node}
//       {e2.xyz, shape kind_flags}   (as uint bits)
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <thread>
#include <vector>

#include "../../include/rt_abi.h"

namespace rtaccel {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMinTris = 12;    // subtrees with fewer triangles keep the reference walk
#ifndef RT_LEAF_TRIS
#define RT_LEAF_TRIS 6
#endif
static_assert(RT_LEAF_TRIS >= 1 && RT_LEAF_TRIS <= 15, "leaf count is a 4-bit field");
constexpr uint32_t kLeafTris = RT_LEAF_TRIS;    // max triangles per leaf
constexpr int kMaxDepth = 20;        // of the binary tree: 10 levels when folded, 3 stack entries a level (RT_AUX_STACK 32)
constexpr float kPadRel = 0.01f;     // static box padding relative to the triangle's largest extent
constexpr float kPadAbs = 4e-6f;     // ... plus this times the largest |coordinate|

struct Node {  // node of the binary tree the builder makes; folded into Node4 for the device
  float lmin[3];
  uint32_t left;
  float lmax[3];
  uint32_t lcount;
  float rmin[3];
  uint32_t right;
  float rmax[3];
  uint32_t rcount;
  float cone[4];
};
static_assert(sizeof(Node) == 80, "binary build node is 80 B");
// the device node (see Layout): b[2c] = {min of child c, word}, b[2c+1] = {max of child c, word};
// words 0-3: child references, words 4-7: the cone
struct Node4 {
  struct Q {
    float v[3];
    uint32_t w;
  } b[8];
};
static_assert(sizeof(Node4) == 128, "aux node is 128 B");
struct Tri {
  float p1[3];
  uint32_t shape;
  float e1[3];
  uint32_t owner;
  float e2[3];
  uint32_t kind_flags;
};
static_assert(sizeof(Tri) == 48, "aux triangle is 48 B");
// resize() leaves new elements as they are: the leaf triangles are written in pieces on threads, and zeroing 42 MB of
// them first on one thread took as long as writing them
template <class T>
struct RawAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    typedef RawAlloc<U> other;
  };
  RawAlloc() = default;
  template <class U>
  RawAlloc(const RawAlloc<U>&) {}
  template <class U>
  void construct(U* p) {
    ::new ((void*)p) U;
  }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
typedef std::vector<Tri, RawAlloc<Tri>> TriVec;

struct Result {
  std::vector<Node4> nodes;        // all trees back to back, and the entry records
  TriVec tris;           // leaf triangles in leaf order
  std::vector<uint32_t> root_of;   // per reference node: header index of the subtree's BVH (roots only) or kNone
  std::vector<uint32_t> gate_next; // per reference node inside a subtree: next group whose box test is not implied
};

struct Item {
  float lo[3], hi[3], c[3];
  uint32_t tri;    // index into RtScene::tri_geom
  uint32_t shape;  // index into RtScene::shapes
  uint32_t owner;  // node index
  float nrm[3];    // unit normal (zero if degenerate)
  float kalpha;    // 1 / sin(angle(e1, e2)), inf if degenerate
};

struct TreeBuilder {
  const RtTriGeom* geom = nullptr;  // RtScene::tri_geom
  const RtShape* shapes_ = nullptr;  // RtScene::shapes
  std::vector<Item> items;
  std::vector<Node> nodes;  // local; [0] header, [1] top
  TriVec tris;
  float lmax = 0.f, amax = 0.f;
  int depth_seen = 0;
  int par_levels = 0;  // top levels of the tree whose halves are built by separate threads (2^par_levels threads)
  float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};

  bool add(uint32_t tri, uint32_t shape, uint32_t owner) {
    const RtTriGeom& g = geom[tri];
    Item it;
    it.tri = tri, it.shape = shape, it.owner = owner;
    float ext = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) {
      float p1 = g.p1[a], p2 = p1 + g.e1[a], p3 = p1 + g.e2[a];
      if (!std::isfinite(p1) || !std::isfinite(p2) || !std::isfinite(p3)) return false;
      it.lo[a] = std::min(p1, std::min(p2, p3));
      it.hi[a] = std::max(p1, std::max(p2, p3));
      ext = std::max(ext, it.hi[a] - it.lo[a]);
      mag = std::max(mag, std::max(std::fabs(it.lo[a]), std::fabs(it.hi[a])));
    }
    float l1 = 0.f, l2 = 0.f, l3 = 0.f;
    for (int a = 0; a < 3; a++) {
      float d = g.e2[a] - g.e1[a];
      l1 += g.e1[a] * g.e1[a], l2 += g.e2[a] * g.e2[a], l3 += d * d;
    }
    {
      double e1[3] = {g.e1[0], g.e1[1], g.e1[2]}, e2[3] = {g.e2[0], g.e2[1], g.e2[2]};
      double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      double cl = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      double sina = cl / (std::sqrt((double)l1) * std::sqrt((double)l2));
      if (cl > 0 && std::isfinite(sina) && sina > 1e-5) {
        for (int a = 0; a < 3; a++) it.nrm[a] = (float)(c[a] / cl);
        it.kalpha = (float)(1.0 / sina);
      } else {
        it.nrm[0] = it.nrm[1] = it.nrm[2] = 0.f;
        it.kalpha = INFINITY;
      }
    }
    float edge = std::sqrt(std::max(l1, std::max(l2, l3)));
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
    items.push_back(it);
    return true;
  }

  // levels a median-split tree needs below a node that holds n triangles
  static int levels_for(uint32_t n) {
    int l = 0;
    while (n > kLeafTris) {
      n = (n + 1) / 2;
      l++;
    }
    return l;
  }
  struct Out {
    std::vector<Node> nodes;
    TriVec tris;
    int depth_seen = 0;
  };
  // child reference for [b,e): (index, count) and its box.  Nodes and leaf triangles go to `out` (indices local to it).
  // par > 0: the two halves of a large range are built by two threads into outputs of their own and appended, left
  // then right -- the same pre-order layout as the serial build.
  void build(uint32_t b, uint32_t e, int depth, uint32_t* ref, uint32_t* count, float* lo, float* hi, Out& out, int par) {
    std::vector<Node>& nodes = out.nodes;
    TriVec& tris = out.tris;
    out.depth_seen = std::max(out.depth_seen, depth);
    for (int a = 0; a < 3; a++) lo[a] = INFINITY, hi[a] = -INFINITY;
    for (uint32_t i = b; i < e; i++)
      for (int a = 0; a < 3; a++) {
        lo[a] = std::min(lo[a], items[i].lo[a]);
        hi[a] = std::max(hi[a], items[i].hi[a]);
      }
    uint32_t n = e - b;
    if (n <= kLeafTris) {
      *ref = (uint32_t)tris.size();
      *count = n;
      for (uint32_t i = b; i < e; i++) {
        const RtTriGeom& g = geom[items[i].tri];
        Tri t{};
        std::memcpy(t.p1, g.p1, 12);
        std::memcpy(t.e1, g.e1, 12);
        std::memcpy(t.e2, g.e2, 12);
        t.shape = items[i].shape;
        t.owner = items[i].owner;
        t.kind_flags = shapes_[items[i].shape].kind_flags;
        tris.push_back(t);
      }
      return;
    }
    // Split: binned surface-area heuristic over the centroids (16 bins per axis); the object median on the axis of
    // largest centroid extent when SAH finds nothing or when its split would leave a child too many triangles for
    // the levels that remain (a median tree below depth d with n triangles needs `levels_for(n)` more levels, and the
    // traversal stack is sized for kMaxDepth).
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = b; i < e; i++)
      for (int a = 0; a < 3; a++) {
        clo[a] = std::min(clo[a], items[i].c[a]);
        chi[a] = std::max(chi[a], items[i].c[a]);
      }
    uint32_t mid = 0;
    {
      constexpr int kBins = 16;
      double best = INFINITY;
      int best_axis = -1, best_split = 0;
      for (int axis = 0; axis < 3; axis++) {
        const float ext = chi[axis] - clo[axis];
        if (!(ext > 0.f)) continue;
        const float scale = kBins / ext;
        uint32_t cnt[kBins] = {0};
        float blo_[kBins][3], bhi_[kBins][3];
        for (int k = 0; k < kBins; k++)
          for (int a = 0; a < 3; a++) blo_[k][a] = INFINITY, bhi_[k][a] = -INFINITY;
        for (uint32_t i = b; i < e; i++) {
          int k = std::min(kBins - 1, std::max(0, (int)((items[i].c[axis] - clo[axis]) * scale)));
          cnt[k]++;
          for (int a = 0; a < 3; a++) {
            blo_[k][a] = std::min(blo_[k][a], items[i].lo[a]);
            bhi_[k][a] = std::max(bhi_[k][a], items[i].hi[a]);
          }
        }
        // sweep from the right, then from the left
        double ra[kBins];
        uint32_t rn[kBins];
        float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t acc = 0;
        auto area = [](const float* l, const float* h) {
          double dx = (double)h[0] - l[0], dy = (double)h[1] - l[1], dz = (double)h[2] - l[2];
          return dx * dy + dy * dz + dz * dx;
        };
        for (int k = kBins - 1; k > 0; k--) {
          for (int a = 0; a < 3; a++) l3[a] = std::min(l3[a], blo_[k][a]), h3[a] = std::max(h3[a], bhi_[k][a]);
          acc += cnt[k];
          ra[k] = acc ? area(l3, h3) : 0.0;
          rn[k] = acc;
        }
        for (int a = 0; a < 3; a++) l3[a] = INFINITY, h3[a] = -INFINITY;
        acc = 0;
        for (int k = 0; k + 1 < kBins; k++) {  // split after bin k
          for (int a = 0; a < 3; a++) l3[a] = std::min(l3[a], blo_[k][a]), h3[a] = std::max(h3[a], bhi_[k][a]);
          acc += cnt[k];
          if (acc == 0 || rn[k + 1] == 0) continue;
          if (depth + 1 + levels_for(acc) > kMaxDepth || depth + 1 + levels_for(rn[k + 1]) > kMaxDepth) continue;
          double cost = area(l3, h3) * acc + ra[k + 1] * rn[k + 1];
          if (cost < best) best = cost, best_axis = axis, best_split = k;
        }
      }
      if (best_axis >= 0) {
        const int axis = best_axis;
        const float scale = kBins / (chi[axis] - clo[axis]);
        const float lo_c = clo[axis];
        const int split = best_split;
        auto it = std::stable_partition(items.begin() + b, items.begin() + e, [=](const Item& x) {
          return std::min(kBins - 1, std::max(0, (int)((x.c[axis] - lo_c) * scale))) <= split;
        });
        mid = (uint32_t)(it - items.begin());
      }
    }
    if (mid <= b || mid >= e) {
      int axis = 0;
      for (int a = 1; a < 3; a++)
        if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
      mid = b + n / 2;
      std::nth_element(items.begin() + b, items.begin() + mid, items.begin() + e, [axis](const Item& x, const Item& y) {
        return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.shape < y.shape);
      });
    }
    uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    *ref = me;
    *count = 0;
    Node nd{};
    {  // normal cone of [b,e): axis = normalised sum of the unit normals, half-angle = largest deviation
      double ax[3] = {0, 0, 0}, kmax = 0;
      bool ok = true;
      for (uint32_t i = b; i < e && ok; i++) {
        ok = std::isfinite(items[i].kalpha);
        kmax = std::max(kmax, (double)items[i].kalpha);
        for (int a = 0; a < 3; a++) ax[a] += items[i].nrm[a];
      }
      double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      nd.cone[0] = nd.cone[1] = nd.cone[2] = 0.f;
      nd.cone[3] = -2.f;
      if (ok && al > 1e-6 * n) {
        double cmin = 1.0;
        for (uint32_t i = b; i < e; i++)
          cmin = std::min(cmin, (ax[0] * items[i].nrm[0] + ax[1] * items[i].nrm[1] + ax[2] * items[i].nrm[2]) / al);
        double w = std::acos(std::max(-1.0, std::min(1.0, cmin))) + 2e-3;  // widened a little: f32 normals
        if (w < 1.4) {                                                     // < 80 degrees, else no useful bound
          double inv_k = 1.0 / (kmax * 1.001);
          for (int a = 0; a < 3; a++) nd.cone[a] = (float)(ax[a] / al * inv_k);
          nd.cone[3] = (float)std::cos(w);
        }
      }
    }
    if (par > 0 && n >= 32768) {
      Out L, R;
      std::thread th([&]() { build(b, mid, depth + 1, &nd.left, &nd.lcount, nd.lmin, nd.lmax, L, par - 1); });
      build(mid, e, depth + 1, &nd.right, &nd.rcount, nd.rmin, nd.rmax, R, par - 1);
      th.join();
      auto append = [&](Out& sub, uint32_t* cref, uint32_t ccount) {
        const uint32_t noff = (uint32_t)nodes.size(), toff = (uint32_t)tris.size();
        for (Node q : sub.nodes) {
          q.left += q.lcount ? toff : noff;
          q.right += q.rcount ? toff : noff;
          nodes.push_back(q);
        }
        tris.insert(tris.end(), sub.tris.begin(), sub.tris.end());
        *cref += ccount ? toff : noff;
        out.depth_seen = std::max(out.depth_seen, sub.depth_seen);
      };
      append(L, &nd.left, nd.lcount);
      append(R, &nd.right, nd.rcount);
    } else {
      build(b, mid, depth + 1, &nd.left, &nd.lcount, nd.lmin, nd.lmax, out, 0);
      build(mid, e, depth + 1, &nd.right, &nd.rcount, nd.rmin, nd.rmax, out, 0);
    }
    nodes[me] = nd;
  }

  // false: this subtree keeps the reference walk
  bool finish() {
    uint32_t n = (uint32_t)items.size();
    if (n < kMinTris) return false;
    Out out;
    out.tris.reserve(n);
    out.nodes.emplace_back();  // header
    uint32_t ref, count;
    float lo[3], hi[3];
    build(0, n, 1, &ref, &count, lo, hi, out, par_levels);
    nodes.swap(out.nodes);
    tris.swap(out.tris);
    depth_seen = out.depth_seen;
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

inline bool box_finite(const RtNode& n) {
  for (int a = 0; a < 3; a++)
    if (!std::isfinite(n.bmin[a]) || !std::isfinite(n.bmax[a])) return false;
  return true;
}
inline bool box_inside(const RtNode& c, const RtNode& p) {
  for (int a = 0; a < 3; a++)
    if (!(c.bmin[a] >= p.bmin[a] && c.bmax[a] <= p.bmax[a])) return false;
  return true;
}

// `device_tree`: builds the binary tree of the subtree under a reference node into a TreeBuilder, as
// TreeBuilder::finish leaves it, or returns false (rt_accel_device.h: the same tree, built on the GPU); subtrees of fewer
// than kDeviceMinTris triangles, and those it declines, are built here on host threads.
using DeviceTree = std::function<bool(const RtScene*, uint32_t, TreeBuilder&)>;
constexpr uint32_t kDeviceMinTris = 16384;
inline Result build(const RtScene* s, const DeviceTree& device_tree = nullptr) {
  Result r;
  const uint32_t nn = s->n_nodes;
  // RT_TIMING=1 in the environment: stage times on stderr
  const bool timing = std::getenv("RT_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "rt_accel build: %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  r.root_of.assign(nn, kNone);
  r.gate_next.assign(nn, kNone);
  if (nn == 0) return r;
  // sub_ok[i]: node i and everything below it are identity-transform groups (i itself may carry a
  // transform) whose own shapes are identity-transform triangles, with finite boxes.
  std::vector<uint8_t> sub_ok(nn, 0);
  for (uint32_t i = nn; i-- > 0;) {
    const RtNode& n = s->nodes[i];
    bool ok = (n.n_shapes == 0 || (n.flags & RT_NODE_ALL_TRIS)) && box_finite(n);
    for (uint32_t c = i + 1; ok && c < n.skip; c = s->nodes[c].skip)  // direct children
      ok = sub_ok[c] && s->nodes[c].xform == 0;
    sub_ok[i] = ok;
  }
  std::vector<uint32_t> roots;
  for (uint32_t i = 0; i < nn; i++) {
    const RtNode& n = s->nodes[i];
    bool inside_parent = n.parent >= 0 && sub_ok[n.parent] && n.xform == 0;
    if (sub_ok[i] && !inside_parent) roots.push_back(i);
  }
  // A subtree with more triangles than a tree of kMaxDepth levels can hold (the builder falls back to median splits
  // where SAH would go deeper, so the count alone decides: 6 * 2^19 = 3.1 M) is covered piecewise: its child groups
  // become roots of their own, and its own shapes keep the reference's linear tests.  RT_AUX_MAX_TRIS in the
  // environment lowers the limit (tests).
  {
    uint64_t max_tris = ~0ull;
    if (const char* e = std::getenv("RT_AUX_MAX_TRIS")) max_tris = std::strtoull(e, nullptr, 10);
    std::vector<uint64_t> below(nn, 0);  // triangles in the subtree of node i
    for (uint32_t i = nn; i-- > 0;) {
      below[i] += s->nodes[i].n_shapes;
      if (s->nodes[i].parent >= 0) below[(uint32_t)s->nodes[i].parent] += below[i];
    }
    std::vector<uint32_t> fit;
    while (!roots.empty()) {
      const uint32_t R = roots.back();
      roots.pop_back();
      const uint64_t n = below[R];
      const bool too_big = n > max_tris || n > 0xFFFFFFF0ull || 1 + TreeBuilder::levels_for((uint32_t)n) > kMaxDepth;
      if (!too_big) {
        fit.push_back(R);
        continue;
      }
      for (uint32_t c = R + 1; c < s->nodes[R].skip; c = s->nodes[c].skip) roots.push_back(c);  // direct children
    }
    std::sort(fit.begin(), fit.end());
    roots.swap(fit);
  }
  // The first root's tree is needed whatever the instance detection below finds (it is nobody's copy): with a device
  // builder it is started now and built on the GPU while this thread hashes and compares.
  struct Early {
    bool ok = false;
    TreeBuilder tb;
  };
  std::future<Early> early;
  if (device_tree && !roots.empty()) {
    uint64_t n0 = 0;
    for (uint32_t i = roots[0]; i < s->nodes[roots[0]].skip; i++) n0 += s->nodes[i].n_shapes;
    if (n0 >= kDeviceMinTris)
      early = std::async(std::launch::async, [s, &device_tree, R0 = roots[0]] {
        Early e;
        e.tb.geom = s->tri_geom;
        e.tb.shapes_ = s->shapes;
        e.ok = device_tree(s, R0, e.tb);
        return e;
      });
  }
  // Instancing: subtrees with the same triangles in the same arrangement (the reference re-parses and duplicates an
  // OBJ file for every instance, scene.rs:593-599) share one tree.  Leaf triangles hold the shape index relative to
  // the instance's first shape and the owning group relative to the instance's root; every root gets its own
  // 128-B entry record {centre, radius}{lmax, amax, index of the tree's top node, first shape of the instance}.
  struct Inst {
    uint32_t n_items = 0, shape_base = 0;
    uint64_t hash = 0;
    size_t same_as = 0;  // index into roots of the instance whose tree this one uses
  };
  auto for_items = [&](uint32_t R, auto&& f) {
    for (uint32_t i = R; i < s->nodes[R].skip; i++) {
      const RtNode& n = s->nodes[i];
      for (uint32_t k = 0; k < n.n_shapes; k++) f(n.first_tri + k, n.first_shape + k, i);
    }
  };
  // Two subtrees are the same instance when their items agree one by one: the triangle (see below), the shape index
  // relative to the instance's first shape, the owning group relative to the root, the kind and flags.  "The same
  // triangle" is decided in two rounds.  First by index: this repo's flattener shares one copy of the triangle data
  // between the instances of a mesh, so equal indices settle it for them (dragons.yaml: five of six roots) without
  // looking at a triangle.  Roots still on their own after that are compared by CANONICAL triangle ids -- the
  // smallest index with the same 48 bytes of geometry -- so that instances are also recognised for a host that
  // flattens the reference's duplicated World as it is (one hash table over all triangles, built only then).
  std::vector<Inst> inst(roots.size());
  for (size_t j = 0; j < roots.size(); j++) {
    Inst& I = inst[j];
    I.same_as = j;
    I.shape_base = 0xFFFFFFFFu;
    for (uint32_t i = roots[j]; i < s->nodes[roots[j]].skip; i++) {
      const RtNode& n = s->nodes[i];
      if (n.n_shapes == 0) continue;
      I.n_items += n.n_shapes;
      I.shape_base = std::min(I.shape_base, n.first_shape);
    }
  }
  auto hash_root = [&](size_t j, const uint32_t* canon) {
    Inst& I = inst[j];
    uint64_t h = 1469598103934665603ull;
    for_items(roots[j], [&](uint32_t tri, uint32_t shape, uint32_t owner) {
      const uint64_t a = ((uint64_t)(canon ? canon[tri] : tri) << 32) | (shape - I.shape_base);
      const uint64_t b = ((uint64_t)(owner - roots[j]) << 32) | s->shapes[shape].kind_flags;
      h = (h ^ a) * 0x100000001B3ull;
      h ^= h >> 29;
      h = (h ^ b) * 0x100000001B3ull;
      h ^= h >> 29;
    });
    I.hash = h;
  };
  // item by item, two cursors (groups without shapes of their own do not count)
  auto same_items = [&](size_t j1, size_t j2, const uint32_t* canon) {
    if (inst[j1].n_items != inst[j2].n_items) return false;
    const uint32_t R1 = roots[j1], R2 = roots[j2], E2 = s->nodes[R2].skip;
    uint32_t i2 = R2, k2 = 0;
    for (uint32_t i1 = R1; i1 < s->nodes[R1].skip; i1++) {
      const RtNode& a = s->nodes[i1];
      for (uint32_t k1 = 0; k1 < a.n_shapes; k1++) {
        while (i2 < E2 && k2 >= s->nodes[i2].n_shapes) i2++, k2 = 0;
        if (i2 >= E2) return false;
        const RtNode& b = s->nodes[i2];
        const uint32_t t1 = a.first_tri + k1, t2 = b.first_tri + k2, s1 = a.first_shape + k1, s2 = b.first_shape + k2;
        if ((canon ? canon[t1] != canon[t2] : t1 != t2) || s1 - inst[j1].shape_base != s2 - inst[j2].shape_base ||
            i1 - R1 != i2 - R2 || s->shapes[s1].kind_flags != s->shapes[s2].kind_flags)
          return false;
        k2++;
      }
    }
    return true;
  };
  // one root per thread (six dragons are 5.2 M items): hash, then compare with the first root of the same hash; what
  // that leaves undecided (a root that differs from the first of its hash) is settled in order afterwards
  auto round = [&](const std::vector<size_t>& which, const uint32_t* canon) {
    const unsigned nt_h = which.size() >= 2 ? std::max(1u, std::min<unsigned>(8u, std::min<unsigned>((unsigned)which.size(), std::thread::hardware_concurrency()))) : 1u;
    auto on_threads = [&](auto&& f) {
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt_h; t++)
        pool.emplace_back([&, t] {
          for (size_t q = t; q < which.size(); q += nt_h) f(q);
        });
      for (size_t q = 0; q < which.size(); q += nt_h) f(q);
      for (auto& th : pool) th.join();
    };
    on_threads([&](size_t q) { hash_root(which[q], canon); });
    std::vector<int> first_same(which.size(), -1);  // position in `which` of the first root with this hash, if it is the same
    std::vector<uint8_t> undecided(which.size(), 0);
    on_threads([&](size_t q) {
      for (size_t p = 0; p < q; p++)
        if (inst[which[p]].hash == inst[which[q]].hash) {
          if (same_items(which[p], which[q], canon))
            first_same[q] = (int)p;
          else
            undecided[q] = 1;
          return;
        }
    });
    for (size_t q = 0; q < which.size(); q++) {
      const size_t j = which[q];
      if (first_same[q] >= 0) {
        inst[j].same_as = inst[which[(size_t)first_same[q]]].same_as;  // (the first of a hash is settled before its followers)
      } else if (undecided[q]) {
        for (size_t p = 0; p < q; p++)
          if (inst[which[p]].same_as == which[p] && inst[which[p]].hash == inst[j].hash && same_items(which[p], j, canon)) {
            inst[j].same_as = which[p];
            break;
          }
      }
    }
  };
  {
    std::vector<size_t> all(roots.size());
    for (size_t j = 0; j < roots.size(); j++) all[j] = j;
    round(all, nullptr);
  }
  stage("roots, instances by index");
  {
    std::vector<size_t> alone;
    for (size_t j = 0; j < roots.size(); j++)
      if (inst[j].same_as == j) alone.push_back(j);
    if (alone.size() >= 2) {
      std::vector<uint32_t> canon(s->n_tris);
      size_t cap = 16;
      while (cap < (size_t)s->n_tris * 2 + 16) cap <<= 1;
      std::vector<uint32_t> slots(cap, kNone);
      for (uint32_t t = 0; t < s->n_tris; t++) {
        uint64_t h = 1469598103934665603ull;
        uint64_t w[6];
        std::memcpy(w, &s->tri_geom[t], 48);
        for (uint64_t x : w) {
          h = (h ^ x) * 0x100000001B3ull;
          h ^= h >> 29;
        }
        size_t i = h & (cap - 1);
        while (slots[i] != kNone && std::memcmp(&s->tri_geom[slots[i]], &s->tri_geom[t], 48) != 0) i = (i + 1) & (cap - 1);
        if (slots[i] == kNone) slots[i] = t;
        canon[t] = slots[i];
      }
      round(alone, canon.data());
      for (size_t j = 0; j < roots.size(); j++) inst[j].same_as = inst[inst[j].same_as].same_as;  // followers of a root that has just found its own
    }
  }
  stage("instances");
  struct Out {
    std::vector<Node> nodes;
    TriVec tris;
    bool ok = false;
    uint32_t top = 0;  // index of the top node in Result::nodes
  };
  std::vector<Out> outs(roots.size());
  std::vector<size_t> todo;
  for (size_t j = 0; j < roots.size(); j++)
    if (inst[j].same_as == j) todo.push_back(j);
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (todo.size() < 2) nt = 1;
  auto work = [&](unsigned t) {
    for (size_t w = t; w < todo.size(); w += nt) {
      const size_t j = todo[w];
      uint32_t R = roots[j];
      TreeBuilder tb;
      tb.geom = s->tri_geom;
      tb.shapes_ = s->shapes;
      Early e;
      if (j == 0 && early.valid()) e = early.get();
      if (e.ok) {
        tb = std::move(e.tb);
        outs[j].ok = true;
      } else if (!(j == 0 && e.tb.geom) && device_tree && inst[j].n_items >= kDeviceMinTris && device_tree(s, R, tb)) {
        outs[j].ok = true;
      } else {
        tb = TreeBuilder();
        tb.geom = s->tri_geom;
        tb.shapes_ = s->shapes;
        while (((size_t)todo.size() << tb.par_levels) < 16) tb.par_levels++;  // about 16 threads in all
        bool ok = true;
        for (uint32_t i = R; ok && i < s->nodes[R].skip; i++) {
          const RtNode& n = s->nodes[i];
          for (uint32_t k = 0; ok && k < n.n_shapes; k++) ok = tb.add(n.first_tri + k, n.first_shape + k, i);
        }
        outs[j].ok = ok && tb.finish();
      }
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
  stage("binary trees (threads)");
  // the trees, without their headers; leaf triangles made relative to the instance
  for (size_t j : todo) {
    if (!outs[j].ok) continue;
    uint64_t nb = r.nodes.size(), tbase = r.tris.size();
    // references are (index << 4 | count): indices stay below 2^28, leaves below 16 triangles
    if (nb + outs[j].nodes.size() + roots.size() >= 0x0FFFFFFFull || tbase + outs[j].tris.size() >= 0x0FFFFFFFull) {
      outs[j].ok = false;
      continue;
    }
    outs[j].top = (uint32_t)nb;
    // fold the binary tree (local indices, header at 0, top at 1) into 4-wide nodes, parents before children
    const std::vector<Node>& bn = outs[j].nodes;
    struct Fold {
      const std::vector<Node>& bn;
      std::vector<Node4>& out;
      uint64_t tbase;
      uint32_t emit(uint32_t q) {
        const uint32_t me = (uint32_t)out.size();
        out.emplace_back();
        struct Child {
          const float *mn, *mx;
          uint32_t ref, count;
        } ch[4];
        int n = 0;
        const Node& N = bn[q];
        auto side = [&](const float* mn, const float* mx, uint32_t ref, uint32_t count) {
          if (count == 0) {  // an inner child: its two children take its place
            const Node& C = bn[ref];
            ch[n++] = Child{C.lmin, C.lmax, C.left, C.lcount};
            ch[n++] = Child{C.rmin, C.rmax, C.right, C.rcount};
          } else {
            ch[n++] = Child{mn, mx, ref, count};
          }
        };
        side(N.lmin, N.lmax, N.left, N.lcount);
        side(N.rmin, N.rmax, N.right, N.rcount);
        Node4 nd;
        for (int c = 0; c < 4; c++) {
          for (int a = 0; a < 3; a++) {
            nd.b[2 * c].v[a] = c < n ? ch[c].mn[a] : INFINITY;
            nd.b[2 * c + 1].v[a] = c < n ? ch[c].mx[a] : -INFINITY;
          }
          uint32_t ref = kNone;
          if (c < n) ref = ch[c].count ? (uint32_t)(((tbase + ch[c].ref) << 4) | ch[c].count) : (emit(ch[c].ref) << 4);
          nd.b[c].w = ref;
          std::memcpy(&nd.b[4 + c].w, &N.cone[c], 4);
        }
        out[me] = nd;
        return me;
      }
    } fold{bn, r.nodes, tbase};
    fold.emit(1);
    {  // leaf triangles: relative to the instance, in pieces on threads (871 k of them for a dragon)
      TriVec& src = outs[j].tris;
      const uint32_t sb = inst[j].shape_base, rb = roots[j];
      const size_t n = src.size();
      const unsigned ntc = n >= (1u << 16) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
      const bool moved = r.tris.empty();
      if (moved)
        r.tris = std::move(src);
      else
        r.tris.resize(tbase + n);
      Tri* dst = r.tris.data() + tbase;
      const Tri* from = moved ? dst : src.data();
      auto piece = [&](unsigned t) {
        for (size_t k = n * t / ntc; k < n * (t + 1) / ntc; k++) {
          Tri x = from[k];
          x.shape -= sb;
          x.owner -= rb;
          dst[k] = x;
        }
      };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < ntc; t++) pool.emplace_back(piece, t);
      piece(0);
      for (auto& th : pool) th.join();
    }
  }
  stage("fold to 4-wide nodes, leaves");
  // one entry record per root
  for (size_t j = 0; j < roots.size(); j++) {
    const Out& o = outs[inst[j].same_as];
    if (!o.ok) continue;
    uint32_t R = roots[j];
    const Node& h2 = o.nodes[0];  // {centre, radius}{lmax, amax, .., ..} from TreeBuilder::finish
    Node4 h{};
    std::memcpy(h.b[0].v, h2.lmin, 12);
    h.b[0].w = h2.left;
    h.b[1].v[0] = h2.lmax[0], h.b[1].v[1] = h2.lmax[1];
    std::memcpy(&h.b[1].v[2], &o.top, 4);
    h.b[1].w = inst[j].shape_base;
    r.root_of[R] = (uint32_t)r.nodes.size();
    r.nodes.push_back(h);
    // gate chains: gate_next[c] = nearest proper ancestor A below R whose test is not implied by the
    // tests already made on the way up, i.e. the node just below A on the path is not nested in A.
    for (uint32_t c = R + 1; c < s->nodes[R].skip; c++) {
      uint32_t p = (uint32_t)s->nodes[c].parent;
      if (p == R)
        r.gate_next[c] = kNone;
      else if (box_inside(s->nodes[c], s->nodes[p]))
        r.gate_next[c] = r.gate_next[p];
      else
        r.gate_next[c] = p;
    }
  }
  stage("entry records, gate chains");
  return r;
}


// ---------------------------------------------------------------------------------------------------------
// Cull spheres for bounded analytic shapes (cubes, spheres).  The reference tests every ray against every shape
// of a group / of the world: ray into object space (shapes.rs:204-246), then the local test.  Before that the kernel
// tests the ray, in the space it is in, against a sphere around the shape; a miss skips the transform and the test.
// A ray the sphere test rejects must be one the reference's own f32 test rejects:
//  * cube (cube.rs:67-114): an axis with |d| < 1e-4 is treated as parallel, which lets the accepted line drift by up to
//    1e-4 |t| on that axis, |t| <= sqrt(3) (|o - c| + r) / |d| at the box (bounded through the largest direction
//    component, which is never "parallel" unless all are, and then the test only asks whether the origin is inside).
//    In the outer space, A = linear part of the inverse transform: drift |d| <= sqrt(3) 1e-4 |A^-1|^2 (|A| |o-c| + r_l).
//  * sphere (sphere.rs:64-80): disc = b^2 - 4ac = 4a (1 - rho^2), rho = distance of the local line from the origin;
//    its f32 evaluation errs by < 32 u a |o'|^2 (u = 2^-24), so the sign is certain for rho^2 > 1 + 2e-6 |o'|^2;
//    |o'| <= |A| |o - c|.
//  * rounding of the f32 transform itself: below 2e-6 |A^-1| (|A| (|o - c| + |c|) + |t| + r_l).
// The kernel rejects when  |(c - o) x d| > |d| (R + q1 |o-c| + q0 + kq |o-c|^2) + k1 |o-c| + k0  (times 1 + 1e-4),
// R = radius of the shape's image (farthest cube corner; largest singular value of A^-1 for the sphere), the
// coefficients below with Frobenius norms as upper bounds and factors of 2-5 to spare.  Non-finite => never rejected.
// 12 floats per record: {c.xyz, R} {q1, q0, kq, k1} {k0, 0, 0, 0}.
// ---------------------------------------------------------------------------------------------------------
inline double sym3_max_eig(double m[3][3]) {  // Jacobi rotations on a symmetric 3x3
  double a[3][3];
  std::memcpy(a, m, sizeof(a));
  for (int sweep = 0; sweep < 32; sweep++) {
    double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (std::fabs(a[p][q]) < 1e-300) continue;
        double th = 0.5 * std::atan2(2.0 * a[p][q], a[q][q] - a[p][p]);
        double c = std::cos(th), sn = std::sin(th);
        double r[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        r[p][p] = c, r[q][q] = c, r[p][q] = sn, r[q][p] = -sn;
        double t[3][3] = {}, u[3][3] = {};
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) t[i][j] += r[k][i] * a[k][j];  // R^T a
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) u[i][j] += t[i][k] * r[k][j];  // (R^T a) R
        std::memcpy(a, u, sizeof(a));
      }
  }
  double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
  return std::max(a[0][0], std::max(a[1][1], a[2][2])) + off;  // Gershgorin: never below the true maximum
}
// false: no record (the shape is never culled)
inline bool shape_cull_record(const RtScene* s, const RtShape& sh, float out[12]) {
  const uint32_t kind = sh.kind_flags & RT_SHAPE_KIND_MASK;
  if (kind != RT_CUBE && kind != RT_SPHERE) return false;
  if (sh.xform >= s->n_xforms) return false;
  const float* m = s->xforms[sh.xform].m;
  double A[3][3], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) A[r][c] = m[4 * r + c];
    t[r] = m[4 * r + 3];
  }
  double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
               A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  if (!std::isfinite(det) || !(std::fabs(det) > 1e-300)) return false;
  double B[3][3];  // A^-1: object -> outer space
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;
      B[r][c] = (A[r1][c1] * A[r2][c2] - A[r1][c2] * A[r2][c1]) / det;
    }
  double nA = 0, nB = 0, nt = 0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) nA += A[r][c] * A[r][c], nB += B[r][c] * B[r][c];
    nt += t[r] * t[r];
  }
  nA = std::sqrt(nA), nB = std::sqrt(nB), nt = std::sqrt(nt);
  double cp[3];  // image of the object-space origin: A^-1 (0 - t)
  for (int r = 0; r < 3; r++) cp[r] = -(B[r][0] * t[0] + B[r][1] * t[1] + B[r][2] * t[2]);
  const double ncp = std::sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
  double R = 0, rl = 0, k1 = 0, k0 = 0, kq = 0;
  if (kind == RT_CUBE) {
    rl = std::sqrt(3.0);
    for (int k = 0; k < 8; k++) {
      double q[3] = {k & 1 ? 1.0 : -1.0, k & 2 ? 1.0 : -1.0, k & 4 ? 1.0 : -1.0};
      double d2 = 0;
      for (int r = 0; r < 3; r++) {
        double v = B[r][0] * q[0] + B[r][1] * q[1] + B[r][2] * q[2];
        d2 += v * v;
      }
      R = std::max(R, std::sqrt(d2));
    }
    k1 = 4e-4 * nB * nB * nA;
    k0 = 4e-4 * nB * nB * rl;
  } else {
    rl = 1.0;
    double btb[3][3] = {};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) btb[i][j] += B[k][i] * B[k][j];
    R = std::sqrt(sym3_max_eig(btb)) * (1.0 + 1e-9);
    kq = 1e-5 * nA * nA * nB;  // with the constant part (|o'| < 1: the error is relative to 1, not to |o'|^2) in q0
  }
  const double q1 = 4e-6 * nB * nA;
  const double q0 = 4e-6 * nB * (nA * ncp + nt + rl) + 1e-6 * (ncp + R)  // last term: c itself is rounded to f32
                    + (kind == RT_SPHERE ? 1e-5 * nB : 0.0);
  const double v[12] = {cp[0], cp[1], cp[2], R * (1.0 + 1e-4), q1, q0, kq, k1, k0, 0, 0, 0};
  for (double x : v)
    if (!std::isfinite(x) || std::fabs(x) > 1e30) return false;
  for (int i = 0; i < 3; i++) out[i] = (float)v[i];
  for (int i = 3; i < 9; i++) out[i] = std::nextafter((float)v[i], INFINITY);
  out[9] = out[10] = out[11] = 0.f;
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// Cull records for whole subtrees under a transformed group.  The reference enters every group by transforming the
// ray into its space and testing its box (groups.rs:127-133, :110-113), for every ray and every top-level group.  A
// transformed group gets a record, in the space its PARENT's ray lives in, such that a ray the record rejects cannot be
// accepted by the f32 test of ANY shape below the group; such a ray skips the subtree without the transform and the
// box test.  Whatever the reference's own (possibly too small or shifted) box says is then irrelevant: it
// decides which shapes are reached, and none of them would have produced an intersection.
// Same record layout and device test as shape_cull_record, plus two coefficients:
//   reject  <=>  |(c - o) x d| > ( |d| (R + q1 l + kq l^2 + |d| (kd1 l + kd0)) + k1 l + k0 ) (1 + 1e-4),   l = |o - c|
//   12 floats: {c.xyz, R} {q1, 0, kq, k1} {k0, kd1, kd0, 0}
// Per shape below the group, with M = the chain of inverse transforms from the record's space to the shape's space
// (A its linear part, B = A^-1, Frobenius norms |.|), the bound on the distance from the shape's centre beyond which a
// line is certainly rejected:
//   sphere, cube      the bounds of shape_cull_record with the composite M;
//   closed or open cylinder with finite min, max (cylinder.rs:95-156): a computed root t satisfies the f32 quadratic
//                     up to a perturbation of c by < 16u |o'|^2, and the point at that t has min < y < max up to 3u
//                     (|o'| + |min| + |max|): accepted wall hits lie within radius sqrt(1 + 2e-6 |o'|^2) of the axis, cap
//                     hits likewise (cylinder.rs:89-93) -- a sphere around the finite cylinder, inflated by
//                     1e-5 (|o'|^2 + 1): the sphere's kq form;
//   triangles         (their own transform identity, collected per space): accepted hits lie within
//                     1.5 * 16u * K (|s| + L) of the triangle, K <= 1e4 |e1||e2||d'| (DESIGN.md "Auxiliary BVH"): a sphere
//                     around the set's box, inflated by 2.4e-2 amax |d'| (|o' - c| + r + lmax): the kd terms;
//   plane, cone, unbounded cylinder, anything non-finite: no record for any group above it.
// Rounding of the k chained f32 transforms: k * 8e-6 * (cond (l + |c| + R) + |B| |t|), cond = max |B||A| along the chain.
// ---------------------------------------------------------------------------------------------------------
struct Aff {
  double A[3][3], t[3];
};
inline Aff aff_identity() {
  Aff r{};
  r.A[0][0] = r.A[1][1] = r.A[2][2] = 1.0;
  return r;
}
inline Aff aff_from(const RtXform& x) {
  Aff r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.A[i][j] = x.m[4 * i + j];
    r.t[i] = x.m[4 * i + 3];
  }
  return r;
}
inline Aff aff_after(const Aff& second, const Aff& first) {  // x -> second(first(x))
  Aff r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.A[i][j] = second.A[i][0] * first.A[0][j] + second.A[i][1] * first.A[1][j] + second.A[i][2] * first.A[2][j];
    r.t[i] = second.A[i][0] * first.t[0] + second.A[i][1] * first.t[1] + second.A[i][2] * first.t[2] + second.t[i];
  }
  return r;
}
inline double frob3(const double A[3][3]) {
  double s = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) s += A[i][j] * A[i][j];
  return std::sqrt(s);
}
inline bool inv3(const double A[3][3], double B[3][3]) {
  double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
               A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  if (!std::isfinite(det) || !(std::fabs(det) > 1e-300)) return false;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;
      B[r][c] = (A[r1][c1] * A[r2][c2] - A[r1][c2] * A[r2][c1]) / det;
    }
  return true;
}
// one bounded thing below a group, as the coefficients of its rejection bound around its own centre (record space)
struct CullItem {
  double c[3], R, q1, q0, kq, k1, k0, kd1, kd0;
};
// chain: M (record space -> item space), number of f32 transform steps, and the worst norms met along the chain
struct Chain {
  Aff M = aff_identity();
  int steps = 0;
  double cond = 1.0, nB_nt = 0.0;  // max |B_j||A_j|, max |B_j||t_j| over the partial products
  bool ok = true;
};
inline Chain chain_then(const Chain& c, const RtXform& x) {
  Chain r = c;
  r.M = aff_after(aff_from(x), c.M);
  r.steps = c.steps + 1;
  double B[3][3];
  if (!inv3(r.M.A, B)) {
    r.ok = false;
    return r;
  }
  const double nA = frob3(r.M.A), nB = frob3(B);
  const double nt = std::sqrt(r.M.t[0] * r.M.t[0] + r.M.t[1] * r.M.t[1] + r.M.t[2] * r.M.t[2]);
  r.cond = std::max(c.cond, nA * nB);
  r.nB_nt = std::max(c.nB_nt, nB * nt);
  r.ok = c.ok && std::isfinite(r.cond) && std::isfinite(r.nB_nt);
  return r;
}
// kind: RT_SPHERE / RT_CUBE / RT_CYLINDER (yc, half height h) / RT_TRIANGLE (a set: centre cT, radius rT, amax, lmax)
inline bool cull_item(const Chain& ch, uint32_t kind, const double centre_obj[3], double r_obj, double amax, double lmax,
                      CullItem* out) {
  if (!ch.ok) return false;
  const Aff& M = ch.M;
  double B[3][3];
  if (!inv3(M.A, B)) return false;
  const double nA = frob3(M.A), nB = frob3(B);
  // centre in record space: B (centre_obj - t)
  double d[3] = {centre_obj[0] - M.t[0], centre_obj[1] - M.t[1], centre_obj[2] - M.t[2]};
  CullItem it{};
  for (int r = 0; r < 3; r++) it.c[r] = B[r][0] * d[0] + B[r][1] * d[1] + B[r][2] * d[2];
  const double ncp = std::sqrt(it.c[0] * it.c[0] + it.c[1] * it.c[1] + it.c[2] * it.c[2]);
  const double nco = std::sqrt(centre_obj[0] * centre_obj[0] + centre_obj[1] * centre_obj[1] + centre_obj[2] * centre_obj[2]);
  double btb[3][3] = {};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) btb[i][j] += B[k][i] * B[k][j];
  const double smax = std::sqrt(sym3_max_eig(btb)) * (1.0 + 1e-9);  // largest stretch of B
  if (kind == RT_CUBE) {
    double R = 0;
    for (int k = 0; k < 8; k++) {
      double q[3] = {k & 1 ? 1.0 : -1.0, k & 2 ? 1.0 : -1.0, k & 4 ? 1.0 : -1.0};
      double d2 = 0;
      for (int r = 0; r < 3; r++) {
        double v = B[r][0] * q[0] + B[r][1] * q[1] + B[r][2] * q[2];
        d2 += v * v;
      }
      R = std::max(R, std::sqrt(d2));
    }
    it.R = R;
    it.k1 = 4e-4 * nB * nB * nA;
    it.k0 = 4e-4 * nB * nB * std::sqrt(3.0);
  } else if (kind == RT_SPHERE || kind == RT_CYLINDER) {
    it.R = smax * r_obj;
    // |o'|^2 <= 2 |A|^2 l^2 + 2 |centre_obj|^2  (o' - centre_obj = A (o - c))
    it.kq = 2e-5 * nA * nA * nB;
    it.q0 = nB * (2e-5 * nco * nco + 2e-5);
  } else {  // triangle set
    it.R = smax * r_obj;
    it.kd1 = 2.4e-2 * amax * nB * nA * nA;
    it.kd0 = 2.4e-2 * amax * nB * nA * (r_obj + lmax);
  }
  // rounding of the chained f32 transforms (and of the record's own centre)
  const double k = std::max(1, ch.steps);
  it.q1 += k * 8e-6 * ch.cond;
  it.q0 += k * 8e-6 * (ch.cond * (ncp + it.R) + ch.nB_nt) + 1e-6 * (ncp + it.R);
  it.R *= 1.0 + 1e-4;
  const double v[] = {it.c[0], it.c[1], it.c[2], it.R, it.q1, it.q0, it.kq, it.k1, it.k0, it.kd1, it.kd0};
  for (double x : v)
    if (!std::isfinite(x) || std::fabs(x) > 1e30) return false;
  *out = it;
  return true;
}
struct NodeCull {
  std::vector<float> recs;        // 12 floats per record
  std::vector<uint32_t> rec_of;   // per node: record index or kNone
};
inline NodeCull build_node_cull(const RtScene* s) {
  NodeCull out;
  const uint32_t nn = s->n_nodes;
  out.rec_of.assign(nn, kNone);
  if (nn == 0) return out;
  // identity triangles collected per space (the nearest transformed node at or above their group; nn = world space)
  struct TriAgg {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double amax = 0, lmax = 0;
    uint64_t n = 0;
    bool ok = true;
  };
  std::vector<uint32_t> xf_nodes;  // transformed nodes, ascending
  for (uint32_t i = 0; i < nn; i++)
    if (s->nodes[i].xform) xf_nodes.push_back(i);
  if (xf_nodes.empty()) return out;
  std::vector<TriAgg> agg(nn);  // indexed by the space's node
  std::vector<uint32_t> loose;  // shapes that are not identity triangles, ascending (grouped shapes only)
  const uint32_t n_grouped = s->n_shapes - s->n_loose_shapes;
  for (uint32_t i = 0; i < nn; i++) {
    const RtNode& n = s->nodes[i];
    for (uint32_t k = 0; k < n.n_shapes; k++) {
      const uint32_t si = n.first_shape + k;
      const RtShape& sh = s->shapes[si];
      const uint32_t kind = sh.kind_flags & RT_SHAPE_KIND_MASK;
      if (kind >= RT_TRIANGLE && sh.xform == 0 && n.next_xf >= 0) {
        TriAgg& a = agg[n.next_xf];
        const RtTriGeom& g = s->tri_geom[sh.prim];
        double l1 = 0, l2 = 0, l3 = 0;
        for (int c = 0; c < 3; c++) {
          const double p1 = g.p1[c], p2 = p1 + g.e1[c], p3 = p1 + g.e2[c];
          if (!std::isfinite(p1) || !std::isfinite(p2) || !std::isfinite(p3)) a.ok = false;
          a.lo[c] = std::min(a.lo[c], std::min(p1, std::min(p2, p3)));
          a.hi[c] = std::max(a.hi[c], std::max(p1, std::max(p2, p3)));
          const double dd = (double)g.e2[c] - g.e1[c];
          l1 += (double)g.e1[c] * g.e1[c], l2 += (double)g.e2[c] * g.e2[c], l3 += dd * dd;
        }
        a.amax = std::max(a.amax, std::sqrt(l1) * std::sqrt(l2));
        a.lmax = std::max(a.lmax, std::sqrt(std::max(l1, std::max(l2, l3))));
        a.n++;
      } else if (n.next_xf >= 0) {
        loose.push_back(si);
      }
    }
  }
  (void)n_grouped;
  uint64_t budget = 4000000;  // chain evaluations over all records: beyond it the remaining groups go without
  for (uint32_t i : xf_nodes) {
    const RtNode& root = s->nodes[i];
    std::vector<CullItem> items;
    bool ok = true;
    // chains of the transformed nodes of the subtree, in ascending (= DFS) order: a node's chain extends that of the
    // nearest transformed node above it
    const auto xb = std::lower_bound(xf_nodes.begin(), xf_nodes.end(), i);
    const auto xe = std::lower_bound(xf_nodes.begin(), xf_nodes.end(), root.skip);
    std::vector<Chain> chains(xe - xb);
    auto chain_of = [&](uint32_t j) -> const Chain& { return chains[std::lower_bound(xb, xe, j) - xb]; };
    for (auto it = xb; it != xe && ok; ++it) {
      const uint32_t j = *it;
      if (budget == 0) ok = false;
      else budget--;
      const RtNode& nj = s->nodes[j];
      Chain base;  // record space
      if (j != i) {
        const int32_t up = nj.parent >= 0 ? s->nodes[nj.parent].next_xf : -1;
        if (up < (int32_t)i) {
          ok = false;
          break;
        }
        base = chain_of((uint32_t)up);
      }
      Chain& c = chains[it - xb];
      c = chain_then(base, s->xforms[nj.xform]);
      ok = ok && c.ok;
      const TriAgg& a = agg[j];
      if (ok && a.n) {
        if (!a.ok) {
          ok = false;
          break;
        }
        double cT[3], r2 = 0;
        for (int q = 0; q < 3; q++) {
          cT[q] = 0.5 * (a.lo[q] + a.hi[q]);
          const double h = 0.5 * (a.hi[q] - a.lo[q]);
          r2 += h * h;
        }
        CullItem ci;
        ok = cull_item(c, RT_TRIANGLE, cT, std::sqrt(r2), a.amax, a.lmax, &ci);
        if (ok) items.push_back(ci);
      }
    }
    // the other shapes of the subtree, one by one
    if (ok) {
      const uint32_t s_lo = root.first_shape;
      const uint32_t s_hi = root.skip < nn ? s->nodes[root.skip].first_shape : s->n_shapes - s->n_loose_shapes;
      for (auto it = std::lower_bound(loose.begin(), loose.end(), s_lo); ok && it != loose.end() && *it < s_hi; ++it) {
        if (budget == 0) ok = false;
        else budget--;
        const RtShape& sh = s->shapes[*it];
        const uint32_t kind = sh.kind_flags & RT_SHAPE_KIND_MASK;
        if (sh.parent < 0 || s->nodes[sh.parent].next_xf < (int32_t)i) {
          ok = false;
          break;
        }
        Chain c = chain_of((uint32_t)s->nodes[sh.parent].next_xf);
        if (sh.xform) c = chain_then(c, s->xforms[sh.xform]);
        CullItem ci;
        const double zero[3] = {0, 0, 0};
        if (kind == RT_SPHERE) {
          ok = cull_item(c, RT_SPHERE, zero, 1.0, 0, 0, &ci);
        } else if (kind == RT_CUBE) {
          ok = cull_item(c, RT_CUBE, zero, std::sqrt(3.0), 0, 0, &ci);
        } else if (kind == RT_CYLINDER && std::isfinite(sh.mn) && std::isfinite(sh.mx) && sh.mn <= sh.mx) {
          const double yc[3] = {0, 0.5 * ((double)sh.mn + sh.mx), 0};
          const double h = 0.5 * ((double)sh.mx - sh.mn);
          ok = cull_item(c, RT_CYLINDER, yc, std::sqrt(1.0 + h * h), 0, 0, &ci);
        } else if (kind >= RT_TRIANGLE) {
          const RtTriGeom& g = s->tri_geom[sh.prim];
          double lo[3], hi[3], cT[3], r2 = 0, l1 = 0, l2 = 0, l3 = 0;
          for (int q = 0; q < 3; q++) {
            const double p1 = g.p1[q], p2 = p1 + g.e1[q], p3 = p1 + g.e2[q];
            lo[q] = std::min(p1, std::min(p2, p3)), hi[q] = std::max(p1, std::max(p2, p3));
            cT[q] = 0.5 * (lo[q] + hi[q]);
            r2 += 0.25 * (hi[q] - lo[q]) * (hi[q] - lo[q]);
            const double dd = (double)g.e2[q] - g.e1[q];
            l1 += (double)g.e1[q] * g.e1[q], l2 += (double)g.e2[q] * g.e2[q], l3 += dd * dd;
          }
          ok = cull_item(c, RT_TRIANGLE, cT, std::sqrt(r2), std::sqrt(l1) * std::sqrt(l2), std::sqrt(std::max(l1, std::max(l2, l3))), &ci);
        } else {
          ok = false;  // plane, cone, unbounded cylinder
        }
        if (ok) items.push_back(ci);
      }
    }
    if (!ok) continue;
    // one record around all items
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const CullItem& it : items)
      for (int q = 0; q < 3; q++) lo[q] = std::min(lo[q], it.c[q] - it.R), hi[q] = std::max(hi[q], it.c[q] + it.R);
    double c[3] = {0, 0, 0}, R = 0, q1 = 0, kq = 0, k1 = 0, k0 = 0, kd1 = 0, kd0 = 0;
    if (!items.empty())
      for (int q = 0; q < 3; q++) c[q] = 0.5 * (lo[q] + hi[q]);
    for (const CullItem& it : items) {
      const double e = std::sqrt((it.c[0] - c[0]) * (it.c[0] - c[0]) + (it.c[1] - c[1]) * (it.c[1] - c[1]) + (it.c[2] - c[2]) * (it.c[2] - c[2]));
      R = std::max(R, e + it.R + it.q1 * e + it.q0 + it.kq * e * e);
      q1 = std::max(q1, it.q1 + 2.0 * it.kq * e);
      kq = std::max(kq, it.kq);
      k1 = std::max(k1, it.k1);
      k0 = std::max(k0, it.k1 * e + it.k0);
      kd1 = std::max(kd1, it.kd1);
      kd0 = std::max(kd0, it.kd1 * e + it.kd0);
    }
    if (items.empty()) continue;
    const double v[12] = {c[0], c[1], c[2], R, q1, 0, kq, k1, k0, kd1, kd0, 0};
    bool fin = true;
    for (double x : v) fin = fin && std::isfinite(x) && std::fabs(x) < 1e30;
    if (!fin) continue;
    out.rec_of[i] = (uint32_t)(out.recs.size() / 12);
    for (int q = 0; q < 3; q++) out.recs.push_back((float)v[q]);
    for (int q = 3; q < 12; q++) out.recs.push_back(v[q] < 0 ? (float)v[q] : std::nextafter((float)v[q], INFINITY));
  }
  return out;
}

}  // namespace rtaccel
