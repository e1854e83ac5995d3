// rt_host.cpp — librt_host.so: host side of the MI355X render path.
//
// Mirrors the reference's World-building interface (include/rt_builder.h, prefix rth_) in C++
// because the reference is compiled code and there is no Rust toolchain here, and FLATTENS the
// world into the device scene of include/rt_abi.h.  Rendering itself only happens in
// librt_hip.so (HIP kernels); there is no CPU render path in this library.
//
// Host arithmetic that decides device inputs is restated exactly (f32, reference op order):
//   * Matrix<4,4>::inverse by cofactor expansion (src/matrices.rs:146-198);
//   * Transform op composition `op * acc` (src/transformations.rs:60-92), view_transform (:122-134);
//   * BoundingBox add/transform/split/contains (src/bounds.rs:36-153);
//   * Group::make_bounds with its null-parent quirk, partition and divide (src/groups.rs:166-299);
//   * Camera::new (src/camera.rs:19-38), Triangle::new (src/shapes/triangle.rs:32-48).
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (see __graft_entry__).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#define RTB_PREFIX(name) rth_##name
#include "../../../include/rt_abi.h"
#include "../../../include/rt_builder.h"

namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------------------------------
// 4x4 matrices, row-major flat storage
// ---------------------------------------------------------------------------------------
struct M4 {
  float a[16];
  float& at(int r, int c) { return a[4 * r + c]; }
  float at(int r, int c) const { return a[4 * r + c]; }
};
M4 m4_identity() {
  M4 m{};
  m.a[0] = m.a[5] = m.a[10] = m.a[15] = 1.f;
  return m;
}
bool m4_is_identity(const M4& m) {
  M4 id = m4_identity();
  for (int i = 0; i < 16; i++)
    if (!(m.a[i] == id.a[i])) return false;
  return true;
}
// src/matrices.rs:68-80: each entry is the k-ordered sum starting from 0.
M4 m4_mul(const M4& x, const M4& y) {
  M4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float s = 0.f;
      for (int k = 0; k < 4; k++) s += x.at(i, k) * y.at(k, j);
      r.at(i, j) = s;
    }
  return r;
}
// Determinant of the n x n matrix obtained from `m` (stride 4) by keeping rows[]/cols[]:
// Laplace expansion along the first kept row, exactly as the reference nests
// Matrix<4,4> -> Matrix<3,3> -> Matrix<2,2> (src/matrices.rs:100-183).
float minor_det(const M4& m, const int* rows, const int* cols, int n) {
  if (n == 2) return m.at(rows[0], cols[0]) * m.at(rows[1], cols[1]) - m.at(rows[1], cols[0]) * m.at(rows[0], cols[1]);
  float det = 0.f;
  int sub_cols[3];
  for (int j = 0; j < n; j++) {
    int w = 0;
    for (int c = 0; c < n; c++)
      if (c != j) sub_cols[w++] = cols[c];
    float minor = minor_det(m, rows + 1, sub_cols, n - 1);
    float cof = (j % 2 == 0) ? minor : -minor;
    det += m.at(rows[0], cols[j]) * cof;
  }
  return det;
}
float m4_cofactor(const M4& m, int row, int col) {  // src/matrices.rs:147-175
  int rows[3], cols[3];
  for (int i = 0, w = 0; i < 4; i++)
    if (i != row) rows[w++] = i;
  for (int j = 0, w = 0; j < 4; j++)
    if (j != col) cols[w++] = j;
  float minor = minor_det(m, rows, cols, 3);
  return ((row + col) % 2 == 0) ? minor : -minor;
}
bool m4_inverse(const M4& m, M4* out) {  // src/matrices.rs:177-197
  float det = 0.f;
  for (int j = 0; j < 4; j++) det += m.at(0, j) * m4_cofactor(m, 0, j);
  if (det == 0.f) return false;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) out->at(j, i) = m4_cofactor(m, i, j) / det;
  return true;
}
struct P3 {
  float x, y, z;
};
// Matrix * Point (src/matrices.rs:200-227): rows 0..2, w = 1.
P3 m4_point(const M4& m, P3 p) {
  float v[4] = {p.x, p.y, p.z, 1.f};
  float r[3];
  for (int i = 0; i < 3; i++) {
    float s = 0.f;
    for (int k = 0; k < 4; k++) s += m.at(i, k) * v[k];
    r[i] = s;
  }
  return P3{r[0], r[1], r[2]};
}
M4 op_matrix(const RtbXfOp& o) {  // src/transformations.rs:6-58
  M4 m = m4_identity();
  switch (o.op) {
    case RTB_XF_TRANSLATE:
      m.at(0, 3) = o.a[0];
      m.at(1, 3) = o.a[1];
      m.at(2, 3) = o.a[2];
      break;
    case RTB_XF_SCALE:
      m.at(0, 0) = o.a[0];
      m.at(1, 1) = o.a[1];
      m.at(2, 2) = o.a[2];
      break;
    case RTB_XF_ROTATE_X:
      m.at(1, 1) = cosf(o.a[0]);
      m.at(1, 2) = -sinf(o.a[0]);
      m.at(2, 1) = sinf(o.a[0]);
      m.at(2, 2) = cosf(o.a[0]);
      break;
    case RTB_XF_ROTATE_Y:
      m.at(0, 0) = cosf(o.a[0]);
      m.at(0, 2) = sinf(o.a[0]);
      m.at(2, 0) = -sinf(o.a[0]);
      m.at(2, 2) = cosf(o.a[0]);
      break;
    case RTB_XF_ROTATE_Z:
      m.at(0, 0) = cosf(o.a[0]);
      m.at(0, 1) = -sinf(o.a[0]);
      m.at(1, 0) = sinf(o.a[0]);
      m.at(1, 1) = cosf(o.a[0]);
      break;
    case RTB_XF_SHEAR:
      m.at(0, 1) = o.a[0];
      m.at(0, 2) = o.a[1];
      m.at(1, 0) = o.a[2];
      m.at(1, 2) = o.a[3];
      m.at(2, 0) = o.a[4];
      m.at(2, 1) = o.a[5];
      break;
    default: std::memcpy(m.a, o.m, sizeof(m.a)); break;
  }
  return m;
}
M4 compose(const RtbXfOp* ops, int n) {  // src/transformations.rs:60-92
  M4 acc = m4_identity();
  for (int i = 0; i < n; i++) acc = m4_mul(op_matrix(ops[i]), acc);
  return acc;
}
P3 sub3(P3 a, P3 b) { return P3{a.x - b.x, a.y - b.y, a.z - b.z}; }
P3 cross3(P3 a, P3 b) { return P3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
P3 normalize3(P3 a) {  // src/tuples.rs:232-238 with w = 0
  float mag = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + 0.f * 0.f);
  return P3{a.x / mag, a.y / mag, a.z / mag};
}
M4 view_matrix(P3 from, P3 to, P3 up) {  // src/transformations.rs:122-134
  P3 forward = normalize3(sub3(to, from));
  P3 upn = normalize3(up);
  P3 left = cross3(forward, upn);
  P3 true_up = cross3(left, forward);
  M4 o{};
  o.at(0, 0) = left.x, o.at(0, 1) = left.y, o.at(0, 2) = left.z;
  o.at(1, 0) = true_up.x, o.at(1, 1) = true_up.y, o.at(1, 2) = true_up.z;
  o.at(2, 0) = -forward.x, o.at(2, 1) = -forward.y, o.at(2, 2) = -forward.z;
  o.at(3, 3) = 1.f;
  RtbXfOp t{};
  t.op = RTB_XF_TRANSLATE;
  t.a[0] = -from.x, t.a[1] = -from.y, t.a[2] = -from.z;
  return m4_mul(o, op_matrix(t));
}

// ---------------------------------------------------------------------------------------
// Boxes (src/bounds.rs)
// ---------------------------------------------------------------------------------------
struct Box {
  P3 lo{kInf, kInf, kInf}, hi{-kInf, -kInf, -kInf};
};
void box_add(Box& b, P3 p) {  // :55-74
  if (p.x < b.lo.x) b.lo.x = p.x;
  if (p.y < b.lo.y) b.lo.y = p.y;
  if (p.z < b.lo.z) b.lo.z = p.z;
  if (p.x > b.hi.x) b.hi.x = p.x;
  if (p.y > b.hi.y) b.hi.y = p.y;
  if (p.z > b.hi.z) b.hi.z = p.z;
}
void box_merge(Box& b, const Box& o) {  // :76-79
  box_add(b, o.lo);
  box_add(b, o.hi);
}
Box box_transform(const Box& b, const M4& t) {  // :36-53, corner order as in the reference
  Box r;
  const float xs[2] = {b.lo.x, b.hi.x}, ys[2] = {b.lo.y, b.hi.y}, zs[2] = {b.lo.z, b.hi.z};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++) box_add(r, m4_point(t, P3{xs[i], ys[j], zs[k]}));
  return r;
}
bool box_has_point(const Box& b, P3 p) {  // :81-85
  return b.lo.x <= p.x && p.x <= b.hi.x && b.lo.y <= p.y && p.y <= b.hi.y && b.lo.z <= p.z && p.z <= b.hi.z;
}
bool box_has_box(const Box& b, const Box& o) { return box_has_point(b, o.lo) && box_has_point(b, o.hi); }  // :87-89
inline bool tot_gt(float a, float b) { return (a > b) || (std::isnan(a) && !std::isnan(b)); }
inline bool tot_eq(float a, float b) { return std::isnan(a) ? std::isnan(b) : a == b; }
void box_split(const Box& b, Box* l, Box* r) {  // :125-153
  float d[3] = {std::fabs(b.hi.x - b.lo.x), std::fabs(b.hi.y - b.lo.y), std::fabs(b.hi.z - b.lo.z)};
  float g = tot_gt(d[0], d[1]) ? d[0] : d[1];
  g = tot_gt(g, d[2]) ? g : d[2];
  P3 mid_lo = b.lo, mid_hi = b.hi;
  if (tot_eq(g, d[0])) {
    mid_hi.x = b.lo.x + d[0] / 2.f;
    mid_lo.x = b.lo.x + d[0] / 2.f;
  } else if (tot_eq(g, d[1])) {
    mid_hi.y = b.lo.y + d[1] / 2.f;
    mid_lo.y = b.lo.y + d[1] / 2.f;
  } else {
    mid_hi.z = b.lo.z + d[2] / 2.f;
    mid_lo.z = b.lo.z + d[2] / 2.f;
  }
  l->lo = b.lo;
  l->hi = mid_hi;
  r->lo = mid_lo;
  r->hi = b.hi;
}

// ---------------------------------------------------------------------------------------
// World model (index based arenas)
// ---------------------------------------------------------------------------------------
struct HUv {
  int kind = 0;
  float width = 0, height = 0;
  float c[5][3] = {};
  int image = -1;
};
struct HImage {
  uint32_t w = 0, h = 0;
  std::vector<float> rgb;
};
struct HPattern {
  int kind = 0;
  float a[3] = {}, b[3] = {};
  M4 xf, inv;
  int uv_mapping = 0;
  int uv[6] = {-1, -1, -1, -1, -1, -1};
};
// resize() of the big arrays leaves new elements of trivial types as they are (every one is written afterwards, in
// pieces on threads): zeroing 5 M shape records on one thread first was a sixth of the flatten, 871 k triangles of a
// mesh two thirds of making them
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
struct HTri {
  P3 p1, p2, p3, e1, e2;
  P3 n1, n2, n3;  // flat: n1 = normal
};
// A triangle mesh kept ONCE however many groups instance it.  The reference re-parses the OBJ file and builds new
// Triangle shapes for every `obj:` use (scene.rs:593-599); their values are equal, so everything derived from the
// values alone -- Triangle::new's edges and normal, parent_space_bounds, the value class of the geometry -- is computed
// once here, and groups hold triangle INDICES into the mesh (HGroup::mtris) instead of shapes.
struct HMesh {
  uint64_t hash = 0;               // of the inputs, to recognise the same mesh handed in again
  std::vector<float> in_pos, in_nrm;
  std::vector<int32_t> in_vidx, in_nidx;
  std::vector<HTri, RawAlloc<HTri>> tris;
  std::vector<uint8_t> smooth;     // RTB_SMOOTH_TRIANGLE or RTB_TRIANGLE per triangle (obj.rs:24-36)
  std::vector<Box> box;            // parent_space_bounds of each triangle (identity transform, shapes.rs:318-320)
  std::vector<uint32_t> gid;       // value class of the triangle's geometry (TriValueTable), RT_VCLASS_NEVER if it holds a NaN
};
struct HShape {
  uint8_t kind = RTB_SPHERE;
  bool shadow = true, closed = false;
  float mn = -kInf, mx = kInf;
  M4 xf = m4_identity(), inv = m4_identity();
  bool identity = true;
  RtbMaterial mat{};
  int tri = -1;
  int parent = -1;
  Box psb;  // parent_space_bounds(), constant for the life of the shape
};
struct HGroup {
  M4 xf = m4_identity(), inv = m4_identity();
  std::vector<uint32_t> shapes;
  std::vector<int> children;
  int parent = -1;
  bool cached = false;
  Box bounds;
  // triangles of a mesh, by index, in insertion order (a group holds either these or `shapes`, never both: adding a
  // shape to a mesh group first turns its triangles into shapes).  All share the material `mmat`.
  int mesh = -1;
  std::vector<uint32_t> mtris;
  RtbMaterial mmat{};
  bool fresh_mesh = false;  // exactly what group_add_mesh made: every triangle of the mesh in order, nothing else
  // mtris is the copy of node `tmpl_node` of divide template `tmpl_id` (-1: not, or changed since): the flattener
  // finds the triangle run of such a group by this key instead of hashing the index sequence
  int tmpl_id = -1, tmpl_node = -1;
  // the group and everything below it still hold exactly the triangles of node `sub_node` of template `sub_id` (its
  // copy, possibly divided further or materialised: the same triangles), all transforms below the identity: its bounds
  // are that node's `sub` box (make_bounds: the union of 871 k boxes, once per template instead of once per instance)
  int sub_id = -1, sub_node = -1;
  // this group is node 0 of an untouched copy of template `inst_tmpl`: every group below is what `instantiate` made
  // (nothing added, moved, divided again or given a transform since), so the flattener emits the subtree from the template
  int inst_tmpl = -1;
  // ... and the groups below it have not been made at all (RtbWorld::realize makes them when anything needs them:
  // nothing does when the copy goes to the flattener as it is)
  bool lazy = false;
};
// The result of `divide` on a fresh mesh group (groups.rs:284-299), kept so that the next instance of the same mesh
// with the same threshold is a copy of the tree instead of a second partition of 871 k triangles: the partition
// depends on the triangles' boxes only, never on the instance's material or on any transform above it.
struct DivNode {
  std::vector<uint32_t> mtris;
  std::vector<int> children;  // indices into DivTemplate::nodes
  Box sub;                    // union of the boxes of every triangle at or below the node
  Box own;                    // ... of the node's own triangles (mtris), merged in their order
};
struct DivTemplate {
  std::vector<DivNode> nodes;  // [0] = the mesh group itself
  int id = -1;
};
struct HCsg {
  int op = 0;
  int lkind = 0, l = -1, rkind = 0, r = -1;
};
struct HLight {
  RtLight l{};
};

RtbMaterial default_material() {  // src/materials.rs:19-33
  RtbMaterial m{};
  m.color[0] = m.color[1] = m.color[2] = 1.f;
  m.ambient = 0.1f;
  m.diffuse = 0.9f;
  m.specular = 0.9f;
  m.shininess = 200.f;
  m.reflective = 0.f;
  m.transparency = 0.f;
  m.refractive_index = 1.f;
  m.pattern = -1;
  return m;
}

// Keys for value classes: raw bits with -0 folded onto +0; NaN poisons the key.
struct KeyBuilder {
  // the value of a shape / material / pattern as bytes (floats with -0 folded into +0, NaN noted): equal keys <=>
  // the reference's derived `==`.  Fixed buffer: a key is built for every one of millions of triangles.
  char bytes[320];
  uint32_t len = 0;
  bool nan = false;
  void put(const void* p) {
    if (len + 4 > sizeof(bytes)) std::abort();  // longest key: smooth triangle, 4 + 16 + 2 + 15 + 9 words
    std::memcpy(bytes + len, p, 4);
    len += 4;
  }
  void f(float v) {
    if (std::isnan(v)) nan = true;
    if (v == 0.f) v = 0.f;
    put(&v);
  }
  void fs(const float* v, int n) {
    for (int i = 0; i < n; i++) f(v[i]);
  }
  void u(uint32_t v) { put(&v); }
};
// Value classes: ids in first-seen order.  Open addressing over an arena of key bytes (an unordered_map of strings
// spent ~10 s on the 5.2 M triangles of dragons.yaml).
struct ClassTable {
  std::vector<char> arena;
  struct Keyed {
    uint64_t where;  // offset << 16 | length of the key in the arena
    uint32_t id;
  };
  std::vector<Keyed> keyed;     // the classes that have a key (fresh() hands out ids without one: millions of them for a mesh)
  std::vector<uint32_t> slots;  // index into keyed + 1, 0 = empty
  uint32_t next = 0;
  static uint64_t hash(const char* p, uint32_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (uint32_t i = 0; i + 4 <= n; i += 4) {
      uint32_t w;
      std::memcpy(&w, p + i, 4);
      h = (h ^ w) * 0xFF51AFD7ED558CCDull;
      h ^= h >> 29;
    }
    return h;
  }
  void grow() {
    size_t cap = slots.empty() ? 1024 : slots.size() * 2;
    std::vector<uint32_t> ns(cap, 0);
    for (uint32_t q = 0; q < keyed.size(); q++) {
      const uint64_t w = keyed[q].where;
      size_t i = hash(arena.data() + (w >> 16), (uint32_t)(w & 0xFFFF)) & (cap - 1);
      while (ns[i]) i = (i + 1) & (cap - 1);
      ns[i] = q + 1;
    }
    slots.swap(ns);
  }
  uint32_t fresh() { return next++; }  // a new class id that no key maps to (the caller keeps its own index to it)
  uint32_t fresh_n(uint32_t n) {       // n of them in a row; returns the first
    next += n;
    return next - n;
  }
  uint32_t get(const KeyBuilder& k) {
    if (k.nan) return RT_VCLASS_NEVER;
    if ((keyed.size() + 1) * 2 > slots.size()) grow();
    const size_t mask = slots.size() - 1;
    size_t i = hash(k.bytes, k.len) & mask;
    while (slots[i]) {
      const Keyed& e = keyed[slots[i] - 1];
      if ((e.where & 0xFFFF) == k.len && std::memcmp(arena.data() + (e.where >> 16), k.bytes, k.len) == 0) return e.id;
      i = (i + 1) & mask;
    }
    keyed.push_back(Keyed{((uint64_t)arena.size() << 16) | k.len, next});
    arena.insert(arena.end(), k.bytes, k.bytes + k.len);
    slots[i] = (uint32_t)keyed.size();
    return next++;
  }
};

// Value classes of triangle GEOMETRY: ids in first-seen order; two triangles get the same id iff the reference's
// derived `==` on Triangle / SmoothTriangle holds for them apart from material, transform and shadow (p1, p2, p3, e1, e2
// and normal / n1..n3, triangle.rs:14-27; a flat and a smooth triangle are different enum variants, never equal).
// Entries point at the triangle itself instead of copying its bytes.
struct TriValueTable {
  std::vector<const HTri*> rep;    // per id: a triangle with that value
  std::vector<uint8_t> rep_smooth;
  std::vector<uint64_t> rep_hash;
  std::vector<uint32_t> slots;     // id + 1, 0 = empty
  static void words(const HTri& t, bool smooth, uint32_t* w, int* n, bool* nan) {
    const P3* ps[8] = {&t.p1, &t.p2, &t.p3, &t.e1, &t.e2, &t.n1, &t.n2, &t.n3};
    const int np = smooth ? 8 : 6;
    *nan = false;
    int k = 0;
    for (int i = 0; i < np; i++) {
      float v[3] = {ps[i]->x, ps[i]->y, ps[i]->z};
      for (float x : v) {
        if (std::isnan(x)) *nan = true;
        if (x == 0.f) x = 0.f;  // -0 == +0
        std::memcpy(&w[k++], &x, 4);
      }
    }
    w[k++] = smooth ? 1u : 0u;
    *n = k;
  }
  static uint64_t hash(const uint32_t* w, int n) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; i++) {
      h = (h ^ w[i]) * 0xFF51AFD7ED558CCDull;
      h ^= h >> 29;
    }
    return h;
  }
  void grow() {
    size_t cap = slots.empty() ? 4096 : slots.size() * 2;
    std::vector<uint32_t> ns(cap, 0);
    for (uint32_t id = 0; id < rep.size(); id++) {
      size_t i = rep_hash[id] & (cap - 1);
      while (ns[i]) i = (i + 1) & (cap - 1);
      ns[i] = id + 1;
    }
    slots.swap(ns);
  }
  void reserve(size_t more) {
    while ((rep.size() + more + 1) * 2 > slots.size()) grow();
  }
  // hash of a triangle's value, or 0 with *nan set (computed ahead, in parallel, for whole meshes)
  static uint64_t value_hash(const HTri& t, bool smooth, bool* nan) {
    uint32_t w[25];
    int n;
    words(t, smooth, w, &n, nan);
    return hash(w, n);
  }
  // `t` must stay where it is for the life of the table (mesh triangles and the world's triangle arena do)
  uint32_t get(const HTri& t, bool smooth) {
    bool nan;
    const uint64_t h = value_hash(t, smooth, &nan);
    return get_hashed(t, smooth, h, nan);
  }
  uint32_t get_hashed(const HTri& t, bool smooth, uint64_t h, bool nan) {
    if (nan) return RT_VCLASS_NEVER;
    if ((rep.size() + 1) * 2 > slots.size()) grow();
    const size_t mask = slots.size() - 1;
    uint32_t w[25];
    int n = 0;
    size_t i = h & mask;
    while (slots[i]) {
      const uint32_t id = slots[i] - 1;
      if (rep_hash[id] == h && (rep_smooth[id] != 0) == smooth) {
        uint32_t w2[25];
        int n2;
        bool nan2;
        if (n == 0) words(t, smooth, w, &n, &nan2);
        words(*rep[id], smooth, w2, &n2, &nan2);
        if (std::memcmp(w, w2, 4 * (size_t)n) == 0) return id;
      }
      i = (i + 1) & mask;
    }
    rep.push_back(&t);
    rep_hash.push_back(h);
    rep_smooth.push_back(smooth ? 1 : 0);
    slots[i] = (uint32_t)rep.size();
    return (uint32_t)rep.size() - 1;
  }
  // The classes of a whole mesh at once, on threads: the ids get_hashed() would give triangle after triangle (a value
  // seen before keeps its id, new values are numbered in the order of their first triangles).  Every triangle finds
  // the slot of its value in the open-addressed table, claiming an empty one with a compare-and-swap; a claimed slot
  // holds the SMALLEST index of the triangles with that value (0x80000000 | index until the ids are known), so that
  // the numbering does not depend on who came first.  Slots only ever go from empty to taken, and triangles of one
  // value probe the same sequence: two of them cannot end up in two slots.
  static bool same_value(const HTri& a, const HTri& b, bool smooth) {
    uint32_t wa[25], wb[25];
    int na, nb;
    bool nan;
    words(a, smooth, wa, &na, &nan);
    words(b, smooth, wb, &nb, &nan);
    return std::memcmp(wa, wb, 4 * (size_t)na) == 0;
  }
  template <class Tris>
  void get_batch(const Tris& tris, const uint8_t* kind, const uint64_t* hs, const uint8_t* nans, uint32_t n, uint32_t* gid, unsigned nt) {
    reserve(n);
    const size_t mask = slots.size() - 1;
    uint32_t* S = slots.data();
    std::vector<uint32_t> slot_of(n), new_id(n, 0);
    auto on_threads = [&](auto&& f) {
      std::vector<std::thread> pool;
      for (unsigned k = 1; k < nt; k++) pool.emplace_back([&, k] { f((uint32_t)((uint64_t)n * k / nt), (uint32_t)((uint64_t)n * (k + 1) / nt)); });
      f(0u, (uint32_t)((uint64_t)n / nt));
      for (auto& th : pool) th.join();
    };
    on_threads([&](uint32_t b, uint32_t e) {
      for (uint32_t t = b; t < e; t++) {
        if (nans[t]) continue;
        const bool smooth = kind[t] == RTB_SMOOTH_TRIANGLE;
        const uint32_t mine = 0x80000000u | t;
        size_t i = hs[t] & mask;
        while (true) {
          uint32_t v = __atomic_load_n(&S[i], __ATOMIC_ACQUIRE);
          if (v == 0u) {
            if (__atomic_compare_exchange_n(&S[i], &v, mine, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) break;
            // (v now holds what another thread put there)
          }
          bool same;
          if (v & 0x80000000u) {
            const uint32_t o = v & 0x7FFFFFFFu;
            same = hs[o] == hs[t] && (kind[o] == RTB_SMOOTH_TRIANGLE) == smooth && same_value(tris[t], tris[o], smooth);
          } else {
            const uint32_t id = v - 1u;
            same = rep_hash[id] == hs[t] && (rep_smooth[id] != 0) == smooth && same_value(tris[t], *rep[id], smooth);
          }
          if (same) {
            while ((v & 0x80000000u) && mine < v && !__atomic_compare_exchange_n(&S[i], &v, mine, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
            }
            break;
          }
          i = (i + 1) & mask;
        }
        slot_of[t] = (uint32_t)i;
      }
    });
    // the first triangle of every new value, in order: its id
    uint32_t next = (uint32_t)rep.size();
    for (uint32_t t = 0; t < n; t++)
      if (!nans[t] && S[slot_of[t]] == (0x80000000u | t)) new_id[t] = next++;
    const uint32_t first_new = (uint32_t)rep.size();
    rep.resize(next);
    rep_smooth.resize(next);
    rep_hash.resize(next);
    on_threads([&](uint32_t b, uint32_t e) {
      for (uint32_t t = b; t < e; t++) {
        if (nans[t]) {
          gid[t] = RT_VCLASS_NEVER;
          continue;
        }
        const uint32_t v = S[slot_of[t]];
        if (v & 0x80000000u) {
          const uint32_t o = v & 0x7FFFFFFFu;
          gid[t] = new_id[o];
          if (o == t) rep[new_id[t]] = &tris[t], rep_smooth[new_id[t]] = kind[t] == RTB_SMOOTH_TRIANGLE, rep_hash[new_id[t]] = hs[t];
        } else {
          gid[t] = v - 1u;
        }
      }
    });
    on_threads([&](uint32_t b, uint32_t e) {  // (after everybody has read the provisional entries)
      for (uint32_t t = b; t < e; t++)
        if (!nans[t] && S[slot_of[t]] == (0x80000000u | t)) S[slot_of[t]] = new_id[t] + 1u;
    });
    (void)first_new;
  }
};

struct Flat {
  std::vector<RtXform> xforms;
  std::vector<RtShape, RawAlloc<RtShape>> shapes;
  std::vector<RtTriGeom, RawAlloc<RtTriGeom>> tri_geom;
  std::vector<RtTriNormals, RawAlloc<RtTriNormals>> tri_normals;
  std::vector<RtNode> nodes;
  std::vector<RtCsgNode> csg_nodes;
  std::vector<RtCsgRoot> csg_roots;
  std::vector<RtMaterial> materials;
  std::vector<RtPattern> patterns;
  std::vector<RtUvPattern> uvpatterns;
  std::vector<RtImage> images;
  std::vector<float> texels;
  std::vector<RtLight> lights;
  RtScene view{};
};

}  // namespace

struct RtbWorld {
  std::vector<HShape> shapes;  // arena: detached shapes and every placed copy
  std::deque<HTri> tris;       // (a deque: TriValueTable keeps pointers to its elements)
  std::vector<uint32_t> tri_gid;  // per entry of `tris`: value class of its geometry
  std::vector<std::unique_ptr<HMesh>> meshes;
  TriValueTable tri_values;
  std::map<std::pair<int, int>, DivTemplate> div_cache;  // (mesh, threshold) -> the divided tree
  std::vector<const DivTemplate*> div_by_id;              // ... by DivTemplate::id (map nodes do not move)
  // g gets something new (a shape, a child, a transform): it and every group above it stop being plain copies
  void no_longer_template_bounds(int g) {
    for (int a = g; a >= 0; a = groups[a].parent) groups[a].sub_id = groups[a].sub_node = groups[a].inst_tmpl = -1;
  }
  // g changes in a way that leaves the triangles below its ancestors where they are (a transform, another divide)
  void no_longer_template_copy(int g) {
    for (int a = g; a >= 0; a = groups[a].parent) groups[a].inst_tmpl = -1;
  }
  std::vector<HGroup> groups;
  std::vector<HCsg> csgs;
  std::vector<HUv> uvs;
  std::vector<HImage> images;
  std::vector<HPattern> patterns;
  std::vector<HLight> lights;
  std::vector<uint32_t> world_shapes;
  std::vector<int> world_groups;
  std::vector<int> world_csgs;
  int recursion_limit = 4;
  std::string error;

  bool dirty = true;
  std::unique_ptr<Flat> flat;
  RtContext* ctx = nullptr;
  RtDeviceScene* dscene = nullptr;
  int device = 0;
  // further GPUs that render a share of every frame (rth_set_devices; rt_render_multi): context and resident scene each
  struct MoreGpu {
    int device = 0;
    RtContext* ctx = nullptr;
    RtDeviceScene* dscene = nullptr;
  };
  std::vector<MoreGpu> more;

  // The device context of a world that gets a big mesh is made on a thread of its own while the caller goes on building
  // the world: the first context of a process loads the library's code object and takes 0.15 s, as long as building and
  // flattening dragons.yaml.  (No GPU: the attempt fails as it would have later, and rth_device_scene reports it.)
  std::future<std::pair<int, RtContext*>> ctx_job;
  void ctx_early() {
    if (ctx || ctx_job.valid()) return;
    const int dev = device;
    ctx_job = std::async(std::launch::async, [dev] {
      RtContext* c = nullptr;
      const int rc = rt_context_create(dev, &c);
      return std::make_pair(rc, rc == RT_OK ? c : (RtContext*)nullptr);
    });
  }
  void ctx_collect() {  // what ctx_early started, if anything, is ours now
    if (!ctx_job.valid()) return;
    const std::pair<int, RtContext*> r = ctx_job.get();
    if (r.second) {
      if (ctx)
        rt_context_destroy(r.second);
      else
        ctx = r.second;
    }
  }
  ~RtbWorld() { drop_device(); }
  void drop_device() {
    ctx_collect();
    if (dscene) rt_scene_destroy(ctx, dscene);
    dscene = nullptr;
    if (ctx) rt_context_destroy(ctx);
    ctx = nullptr;
    for (MoreGpu& m : more) {
      if (m.dscene) rt_scene_destroy(m.ctx, m.dscene);
      if (m.ctx) rt_context_destroy(m.ctx);
      m.dscene = nullptr;
      m.ctx = nullptr;
    }
  }
  void touch() {
    dirty = true;
    if (dscene) {
      rt_scene_destroy(ctx, dscene);
      dscene = nullptr;
    }
    for (MoreGpu& m : more)
      if (m.dscene) {
        rt_scene_destroy(m.ctx, m.dscene);
        m.dscene = nullptr;
      }
  }

  // ---- shapes ---------------------------------------------------------------------------
  Box local_bounds(const HShape& s) const {
    Box b;
    switch (s.kind) {
      case RTB_SPHERE:
      case RTB_CUBE: b.lo = P3{-1.f, -1.f, -1.f}, b.hi = P3{1.f, 1.f, 1.f}; break;     // sphere.rs:86-90, cube.rs:135-139
      case RTB_PLANE: b.lo = P3{-kInf, 0.f, -kInf}, b.hi = P3{kInf, 0.f, kInf}; break;  // plane.rs:79-83
      case RTB_CYLINDER: b.lo = P3{-1.f, s.mn, -1.f}, b.hi = P3{1.f, s.mx, 1.f}; break;  // cylinder.rs:169-173
      case RTB_CONE: {                                                                    // cone.rs:182-193
        float a = std::fabs(s.mn), c = std::fabs(s.mx);
        float lim = (std::isinf(a) || std::isinf(c)) ? kInf : (tot_gt(a, c) ? a : c);
        b.lo = P3{-lim, s.mn, -lim}, b.hi = P3{lim, s.mx, lim};
        break;
      }
      default: {  // triangle.rs:117-123
        const HTri& t = tris[s.tri];
        box_add(b, t.p1);
        box_add(b, t.p2);
        box_add(b, t.p3);
      }
    }
    return b;
  }
  // What a shape really covers in its parent's space, for the flags that let the device cull by distance. The
  // reference's box is not that for a cone: a ray parallel to the cone's side (|a| < EPSILON) gets t = -c / 2b
  // without the min < y < max test (cone.rs:131-141), a hit anywhere on the infinite cone; and check_cap accepts
  // x^2 + z^2 <= |y| (cone.rs:90-94), caps of radius sqrt|y|, wider than the box where |y| < 1. A cone is unbounded.
  Box true_psb(const HShape& s) const {
    if (s.kind != RTB_CONE) return s.psb;
    Box b;
    b.lo = P3{-kInf, -kInf, -kInf}, b.hi = P3{kInf, kInf, kInf};
    return b;
  }
  uint32_t clone_shape(uint32_t id) {
    shapes.push_back(shapes[id]);
    return (uint32_t)shapes.size() - 1;
  }

  // ---- groups (src/groups.rs) -----------------------------------------------------------
  int group_len(int g) const {
    const HGroup& gr = groups[g];
    const size_t kids = gr.lazy ? div_by_id[gr.inst_tmpl]->nodes[0].children.size() : gr.children.size();
    return (int)(gr.shapes.size() + gr.mtris.size() + kids);
  }
  // a mesh group's triangles as ordinary shapes (what the reference holds anyway), for the rare group that mixes a
  // mesh with other shapes; keeps the insertion order: the triangles came first
  void materialise(int g) {
    HGroup& gr = groups[g];
    if (gr.mtris.empty()) return;
    no_longer_template_copy(g);
    const HMesh& m = *meshes[gr.mesh];
    std::vector<uint32_t> made;
    made.reserve(gr.mtris.size());
    for (uint32_t t : gr.mtris) {
      tris.push_back(m.tris[t]);
      tri_gid.push_back(m.gid[t]);
      HShape s;
      s.kind = m.smooth[t];
      s.mat = gr.mmat;
      s.tri = (int)tris.size() - 1;
      s.parent = g;
      s.psb = m.box[t];
      shapes.push_back(s);
      made.push_back((uint32_t)shapes.size() - 1);
    }
    HGroup& g2 = groups[g];
    made.insert(made.end(), g2.shapes.begin(), g2.shapes.end());
    g2.shapes.swap(made);
    g2.mtris.clear();
    g2.tmpl_id = g2.tmpl_node = -1;
    g2.mesh = -1;
    g2.fresh_mesh = false;
  }
  Box group_bounds(int g) const {  // :166-174
    if (groups[g].cached) return groups[g].bounds;
    return make_bounds(g);
  }
  Box make_bounds(int g) const {  // :176-188
    const HGroup& gr = groups[g];
    if (gr.sub_id >= 0 && gr.parent >= 0) return div_by_id[gr.sub_id]->nodes[gr.sub_node].sub;
    Box bb;
    for (uint32_t s : gr.shapes) box_merge(bb, shapes[s].psb);
    if (!gr.mtris.empty()) {
      const HMesh& m = *meshes[gr.mesh];
      for (uint32_t t : gr.mtris) box_merge(bb, m.box[t]);
    }
    if (gr.lazy) {  // (the children that realize() would make have parents: their bounds are their template nodes')
      const DivTemplate& t = *div_by_id[gr.inst_tmpl];
      for (int k : t.nodes[0].children) {
        if (gr.parent < 0) bb = box_transform(bb, gr.xf);
        box_merge(bb, t.nodes[k].sub);
      }
      return bb;
    }
    for (int c : gr.children) {
      if (gr.parent < 0) bb = box_transform(bb, gr.xf);  // the reference's root-with-children quirk
      box_merge(bb, group_bounds(c));
    }
    return bb;
  }
  void attach_child(int g, int c) {  // :64-72
    groups[g].children.push_back(c);
    groups[c].parent = g;
  }
  int new_group() {
    groups.emplace_back();
    return (int)groups.size() - 1;
  }
  void partition(int g, int* lo_out, int* hi_out) {  // :198-282
    Box lb, rb;
    box_split(group_bounds(g), &lb, &rb);
    std::vector<uint32_t> all;
    all.swap(groups[g].shapes);
    std::vector<uint32_t> ls, rs, keep;
    for (uint32_t s : all) {
      if (box_has_box(lb, shapes[s].psb))
        ls.push_back(s);
      else if (box_has_box(rb, shapes[s].psb))
        rs.push_back(s);
      else
        keep.push_back(s);
    }
    groups[g].shapes = std::move(keep);
    // the triangles of a mesh group, by the same rule on their boxes
    std::vector<uint32_t> lt, rt;
    const int mesh = groups[g].mesh;
    const RtbMaterial mmat = groups[g].mmat;
    if (!groups[g].mtris.empty()) {
      const HMesh& m = *meshes[mesh];
      std::vector<uint32_t> allt, keept;
      allt.swap(groups[g].mtris);
      for (uint32_t t : allt) {
        if (box_has_box(lb, m.box[t]))
          lt.push_back(t);
        else if (box_has_box(rb, m.box[t]))
          rt.push_back(t);
        else
          keept.push_back(t);
      }
      groups[g].mtris = std::move(keept);
      groups[g].tmpl_id = groups[g].tmpl_node = -1;
      groups[g].fresh_mesh = false;
    }
    std::vector<int> kids;
    kids.swap(groups[g].children);
    std::vector<int> lk, rk, keepk;
    for (int c : kids) {
      Box cb = group_bounds(c);
      if (box_has_box(lb, cb))
        lk.push_back(c);
      else if (box_has_box(rb, cb))
        rk.push_back(c);
      else
        keepk.push_back(c);
    }
    groups[g].children = std::move(keepk);
    int L = new_group();
    groups[L].shapes = std::move(ls);
    groups[L].mtris = std::move(lt);
    groups[L].mesh = mesh;
    groups[L].mmat = mmat;
    groups[L].children = lk;
    for (uint32_t s : groups[L].shapes) shapes[s].parent = L;
    for (int c : lk) groups[c].parent = L;
    int R = new_group();
    groups[R].shapes = std::move(rs);
    groups[R].mtris = std::move(rt);
    groups[R].mesh = mesh;
    groups[R].mmat = mmat;
    groups[R].children = rk;
    for (uint32_t s : groups[R].shapes) shapes[s].parent = R;
    for (int c : rk) groups[c].parent = R;
    *lo_out = L;
    *hi_out = R;
  }
  int snapshot(int g, DivTemplate& t) const {
    const int me = (int)t.nodes.size();
    t.nodes.emplace_back();
    t.nodes[me].mtris = groups[g].mtris;
    for (int c : groups[g].children) {
      const int k = snapshot(c, t);
      t.nodes[me].children.push_back(k);
    }
    return me;
  }
  // The copy of a template under group g.  The template's nodes are in pre-order and so are the groups a recursive copy
  // would make (a child is created, then everything below it): node k > 0 becomes group base + k - 1, and the copy is
  // one growth of the table and a fill in pieces on threads.
  void instantiate(int g, const DivTemplate& t) {
    HGroup& root = groups[g];
    root.mtris = t.nodes[0].mtris;
    root.tmpl_id = t.id, root.tmpl_node = 0;
    root.sub_id = t.id, root.sub_node = 0;
    root.fresh_mesh = false;
    root.inst_tmpl = t.id;
    root.lazy = t.nodes.size() > 1;
  }
  // the groups below a lazy copy, made now (before anything that looks at them or changes the copy)
  void realize(int g) {
    if (!groups[g].lazy) return;
    groups[g].lazy = false;
    const DivTemplate& t = *div_by_id[groups[g].inst_tmpl];
    if (groups.capacity() < groups.size() + t.nodes.size())  // one growth for the whole copy, with room for the next
      groups.reserve(std::max(groups.capacity() * 2, groups.size() + 2 * t.nodes.size()));
    const size_t n = t.nodes.size();
    const int base = (int)groups.size();
    groups.resize(groups.size() + n - 1);
    const int mesh = groups[g].mesh;
    const RtbMaterial mmat = groups[g].mmat;
    auto group_of = [&](size_t k) { return k == 0 ? g : base + (int)k - 1; };
    auto fill = [&](size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; k++) {
        HGroup& gr = groups[group_of(k)];
        const DivNode& nd = t.nodes[k];
        gr.mtris = nd.mtris;
        gr.tmpl_id = t.id, gr.tmpl_node = (int)k;
        gr.sub_id = t.id, gr.sub_node = (int)k;
        gr.fresh_mesh = false;
        gr.mesh = mesh;
        gr.mmat = mmat;
        gr.children.resize(nd.children.size());
        for (size_t c = 0; c < nd.children.size(); c++) {
          gr.children[c] = group_of((size_t)nd.children[c]);
          groups[gr.children[c]].parent = group_of(k);  // (every node is the child of exactly one)
        }
      }
    };
    const size_t nt = n >= 4096 ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    std::vector<std::thread> th;
    for (size_t i = 1; i < nt; i++) th.emplace_back(fill, n * i / nt, n * (i + 1) / nt);
    fill(0, n / nt);
    for (auto& x : th) x.join();
  }
  // `divide` (groups.rs:284-299) of a group that holds only triangles `tris` of mesh `m`, as a template: partition if
  // threshold <= len (groups.rs:285), attach the non-empty halves, divide them.  Every group met on the way has no
  // children when it is partitioned, so its bounds are the union of its triangles' boxes.  The halves are independent:
  // the top levels run on separate threads (par), each into a template of its own, appended left then right.
  void divide_template(const HMesh& m, std::vector<uint32_t> tris, int threshold, DivTemplate& t, int par) const {
    const int me = (int)t.nodes.size();
    t.nodes.emplace_back();
    {
      Box all;
      for (uint32_t q : tris) box_merge(all, m.box[q]);
      t.nodes[me].sub = all;
    }
    if (threshold <= (int)tris.size()) {
      const Box bb = t.nodes[me].sub;
      Box lb, rb;
      box_split(bb, &lb, &rb);
      std::vector<uint32_t> lt, rt, keep;
      for (uint32_t q : tris) {
        if (box_has_box(lb, m.box[q]))
          lt.push_back(q);
        else if (box_has_box(rb, m.box[q]))
          rt.push_back(q);
        else
          keep.push_back(q);
      }
      tris.swap(keep);
      std::vector<uint32_t>().swap(keep);
      auto child = [&](std::vector<uint32_t>& sub, DivTemplate& into, int p) {
        const int k = (int)into.nodes.size();
        divide_template(m, std::move(sub), threshold, into, p);
        return k;
      };
      if (par > 0 && lt.size() + rt.size() >= 65536 && !lt.empty() && !rt.empty()) {
        DivTemplate tl, tr;
        std::thread th([&]() { child(lt, tl, par - 1); });
        child(rt, tr, par - 1);
        th.join();
        for (DivTemplate* sub : {&tl, &tr}) {
          const int off = (int)t.nodes.size();
          t.nodes[me].children.push_back(off);
          for (DivNode& n : sub->nodes) {
            for (int& c : n.children) c += off;
            t.nodes.push_back(std::move(n));
          }
        }
      } else {
        if (!lt.empty()) {
          const int k = child(lt, t, 0);
          t.nodes[me].children.push_back(k);
        }
        if (!rt.empty()) {
          const int k = child(rt, t, 0);
          t.nodes[me].children.push_back(k);
        }
      }
    }
    {
      Box own;
      for (uint32_t q : tris) box_merge(own, m.box[q]);
      t.nodes[me].own = own;
    }
    t.nodes[me].mtris = std::move(tris);
  }
  void divide(int g, int threshold) {  // :284-299
    // a mesh group exactly as group_add_mesh made it, already attached (so make_bounds' null-parent quirk cannot
    // apply anywhere below): its divided tree depends on (mesh, threshold) only -- built once as a template
    // (divide_template, the same rule on the same boxes), copied for every instance
    if (groups[g].fresh_mesh && groups[g].parent >= 0 && groups[g].shapes.empty() && groups[g].children.empty()) {
      const std::pair<int, int> key(groups[g].mesh, threshold);
      const bool timing = std::getenv("RT_TIMING") != nullptr;
      auto t0 = std::chrono::steady_clock::now();
      auto it = div_cache.find(key);
      if (it == div_cache.end()) {
        DivTemplate t;
        divide_template(*meshes[groups[g].mesh], groups[g].mtris, threshold, t, 4);
        t.id = (int)div_cache.size();
        it = div_cache.emplace(key, std::move(t)).first;
        div_by_id.push_back(&it->second);
        if (timing) std::fprintf(stderr, "divide: template of %zu groups %.1f ms\n", it->second.nodes.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        t0 = std::chrono::steady_clock::now();
      }
      instantiate(g, it->second);
      if (timing) std::fprintf(stderr, "divide: instance %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      return;
    }
    realize(g);
    no_longer_template_copy(g);
    if (threshold <= group_len(g)) {
      int L, R;
      partition(g, &L, &R);
      if (group_len(L) != 0) attach_child(g, L);
      if (group_len(R) != 0) attach_child(g, R);
    }
    std::vector<int> kids = groups[g].children;
    for (int c : kids) divide(c, threshold);
  }

  // ---- flatten (World -> RtScene) -----------------------------------------------------
  bool flatten();
};

namespace {

int fail(RtbWorld* w, const std::string& msg, int code = -1) {
  w->error = msg;
  return code;
}

struct Flattener {
  RtbWorld& w;
  Flat& f;
  std::map<std::vector<uint32_t>, uint32_t> xform_ids;
  ClassTable uv_classes, pattern_classes, material_classes, shape_classes;
  std::unordered_map<uint32_t, uint32_t> material_slot;  // material class -> materials[] index
  std::vector<int> uv_class_of, pattern_class_of;
  std::vector<Box> true_box;  // per node, conservative box of the subtree in the node's space
  // Value classes of triangles: (kind, transform, material class, shadow) is interned once per group, the geometry's
  // class (TriValueTable) indexes a table per such triple.  Equal <=> the reference's Shape == Shape, as for every other
  // shape, without hashing the 100-byte value of each of millions of mesh triangles per instance.
  ClassTable triple_classes;
  std::vector<std::vector<uint32_t>> triple_map;
  std::vector<size_t> triple_set;  // per triple: how many entries of its map hold a class
  // A triple whose classes so far are those of ONE untouched template copy met while its table was empty: class of
  // geometry g = base + rank[g] (TmplFlat::rank_*), and no table is made for it -- unless somebody else turns up with the
  // same triple (map_of).
  struct TripleLazy {
    const uint32_t* rank = nullptr;
    uint32_t base = 0;
  };
  std::vector<TripleLazy> triple_lazy;
  uint32_t* map_of(uint32_t tid) {
    std::vector<uint32_t>& mp = triple_map[tid];
    const size_t n_gid = w.tri_values.rep.size();
    if (mp.size() < n_gid) mp.resize(n_gid, kUnset);
    if (triple_lazy[tid].rank) {
      const TripleLazy tl = triple_lazy[tid];
      triple_lazy[tid] = TripleLazy{};
      for (size_t g = 0; g < n_gid; g++)
        if (tl.rank[g] != kUnset) mp[g] = tl.base + tl.rank[g];
    }
    return mp.data();
  }
  static constexpr uint32_t kUnset = 0xFFFFFFFEu;
  // Triangle data already emitted, by (mesh, sequence of triangle indices): instances of one mesh, divided the same
  // way, share their tri_geom / tri_normals entries (a node's own triangles are a contiguous run of them).
  struct Run {
    int mesh;
    const std::vector<uint32_t>* seq;
    uint32_t first_tri;
    Box box;  // union of the triangles' parent_space_bounds
  };
  std::unordered_map<uint64_t, std::vector<Run>> runs;
  std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> mesh_triples;  // material class -> (flat, smooth) triple ids

  double t_loop = 0, t_run = 0;
  Flattener(RtbWorld& w_, Flat& f_) : w(w_), f(f_) {}
  // What the walk over the groups leaves for later: the shape records of a mesh group's triangles and the triangle
  // data of a run, both plain copies into places fixed by the walk -- done in pieces on threads once the walk is over
  // (run_fills).  What decides a number (value classes, indices) is done by the walk itself, in its order.
  struct ShapeFill {
    size_t base;
    int mesh;
    const std::vector<uint32_t>* mtris;
    uint32_t node, first_tri, mat_index;
    uint32_t tid_flat, tid_smooth;  // RT_VCLASS_NEVER: no value class for any of them
  };
  struct TriFill {
    uint32_t first;
    int mesh;
    const std::vector<uint32_t>* seq;
  };
  std::vector<ShapeFill> shape_fills;
  std::vector<TriFill> tri_fills;
  void fill_shapes(const ShapeFill& sf) {
    const std::vector<uint32_t>& mtris = *sf.mtris;
    const HMesh& m = *w.meshes[sf.mesh];
    // (run_fills: nothing changes the tables any more -- a triple is still lazy or has its table)
    const uint32_t* map_flat = nullptr;
    const uint32_t* map_smooth = nullptr;
    uint32_t add_flat = 0, add_smooth = 0;
    if (sf.tid_flat != RT_VCLASS_NEVER) {
      const TripleLazy &lf = triple_lazy[sf.tid_flat], &ls = triple_lazy[sf.tid_smooth];
      map_flat = lf.rank ? lf.rank : triple_map[sf.tid_flat].data(), add_flat = lf.rank ? lf.base : 0u;
      map_smooth = ls.rank ? ls.rank : triple_map[sf.tid_smooth].data(), add_smooth = ls.rank ? ls.base : 0u;
    }
    for (size_t k = 0; k < mtris.size(); k++) {
      const uint32_t t = mtris[k];
      RtShape r{};
      r.kind_flags = m.smooth[t] | RT_SHAPE_SHADOW;
      r.xform = 0;
      r.material = sf.mat_index;
      r.parent = (int32_t)sf.node;
      const uint32_t gid = m.gid[t];
      const bool smooth = m.smooth[t] == RTB_SMOOTH_TRIANGLE;
      r.vclass = (map_flat && gid != RT_VCLASS_NEVER) ? (smooth ? map_smooth : map_flat)[gid] + (smooth ? add_smooth : add_flat) : RT_VCLASS_NEVER;
      r.prim = sf.first_tri + (uint32_t)k;
      r.mn = -kInf;
      r.mx = kInf;
      f.shapes[sf.base + k] = r;
    }
  }
  void fill_tris(const TriFill& tf) {
    const HMesh& m = *w.meshes[tf.mesh];
    for (size_t k = 0; k < tf.seq->size(); k++) fill_tri(m.tris[(*tf.seq)[k]], &f.tri_geom[tf.first + k], &f.tri_normals[tf.first + k]);
  }
  void run_fills() {
    size_t total = 0;
    for (const ShapeFill& sf : shape_fills) total += sf.mtris->size();
    for (const TriFill& tf : tri_fills) total += 2 * tf.seq->size();
    const unsigned nt = total >= (1u << 16) ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    // pieces of about equal weight, in task order (neighbouring tasks write neighbouring memory)
    auto piece = [&](unsigned i) {
      const size_t lo = total * i / nt, hi = total * (i + 1) / nt;
      size_t at = 0;
      for (const ShapeFill& sf : shape_fills) {
        if (at >= lo && at < hi) fill_shapes(sf);
        at += sf.mtris->size();
      }
      for (const TriFill& tf : tri_fills) {
        if (at >= lo && at < hi) fill_tris(tf);
        at += 2 * tf.seq->size();
      }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; i++) th.emplace_back(piece, i);
    piece(0);
    for (auto& x : th) x.join();
  }

  uint32_t triple_id(uint32_t kind, const M4& xf, uint32_t mclass, bool shadow) {
    if (mclass == RT_VCLASS_NEVER) return RT_VCLASS_NEVER;
    KeyBuilder k;
    k.u(kind), k.fs(xf.a, 16), k.u(mclass), k.u(shadow ? 1 : 0);
    const uint32_t tid = triple_classes.get(k);
    if (tid != RT_VCLASS_NEVER && tid >= triple_map.size()) triple_map.resize(tid + 1), triple_set.resize(tid + 1, 0), triple_lazy.resize(tid + 1);
    return tid;
  }
  uint32_t tri_vclass(uint32_t tid, uint32_t gid) {
    if (tid == RT_VCLASS_NEVER || gid == RT_VCLASS_NEVER) return RT_VCLASS_NEVER;
    uint32_t* mp = map_of(tid);
    if (mp[gid] == kUnset) mp[gid] = shape_classes.fresh(), triple_set[tid]++;
    return mp[gid];
  }
  static void fill_tri(const HTri& t, RtTriGeom* g, RtTriNormals* n) {
    *g = RtTriGeom{};
    *n = RtTriNormals{};
    g->p1[0] = t.p1.x, g->p1[1] = t.p1.y, g->p1[2] = t.p1.z;
    g->e1[0] = t.e1.x, g->e1[1] = t.e1.y, g->e1[2] = t.e1.z;
    g->e2[0] = t.e2.x, g->e2[1] = t.e2.y, g->e2[2] = t.e2.z;
    n->n1[0] = t.n1.x, n->n1[1] = t.n1.y, n->n1[2] = t.n1.z;
    n->n2[0] = t.n2.x, n->n2[1] = t.n2.y, n->n2[2] = t.n2.z;
    n->n3[0] = t.n3.x, n->n3[1] = t.n3.y, n->n3[2] = t.n3.z;
  }
  // A divide template as the flattener needs it, made when its first copy is met: the run of every node (all of them
  // at once, in the template's pre-order = the order a walk over a copy meets them), and what a walk over an untouched
  // copy would compute for every node -- parents, depths, subtree ends, shape offsets, boxes, flags.
  struct TmplFlat {
    std::vector<Run> run;
    std::vector<uint32_t> size_before;  // tri_geom.size() when the node's run was asked for (first_tri of a node without triangles)
    std::vector<int> parent, depth, end;
    std::vector<uint32_t> prefix;       // triangles of the nodes before k in pre-order
    std::vector<Box> box;               // merge of the node's run box and its children's boxes, in that order
    std::vector<uint32_t> flags;
    // the distinct (geometry class, smooth) of the whole template numbered in the order a walk first meets them
    std::vector<uint32_t> rank_flat, rank_smooth;  // per geometry class, kUnset: not in the template
    uint32_t n_first_met = 0, n_flat = 0, n_smooth = 0;
    bool first_met_done = false;
  };
  std::unordered_map<int, TmplFlat> tmpl_flat;
  TmplFlat& template_flat(int tmpl_id, int mesh) {
    auto it = tmpl_flat.find(tmpl_id);
    if (it != tmpl_flat.end()) return it->second;
    const DivTemplate& t = *w.div_by_id[tmpl_id];
    const size_t n = t.nodes.size();
    const auto T0 = std::chrono::steady_clock::now();
    TmplFlat tf;
    tf.run.reserve(n);
    tf.size_before.resize(n);
    tf.parent.assign(n, -1);
    tf.depth.assign(n, 0);
    tf.end.assign(n, 0);
    tf.prefix.assign(n + 1, 0);
    tf.box.resize(n);
    tf.flags.resize(n);
    for (size_t k = 0; k < n; k++) {
      tf.size_before[k] = (uint32_t)f.tri_geom.size();
      // (the node's triangle data goes behind what is there: template runs are not looked up among the runs of
      // other groups, nor entered there)
      const size_t cnt = t.nodes[k].mtris.size();
      if (cnt) {
        f.tri_geom.resize(tf.size_before[k] + cnt);
        f.tri_normals.resize(tf.size_before[k] + cnt);
        tri_fills.push_back(TriFill{tf.size_before[k], mesh, &t.nodes[k].mtris});
      }
      tf.run.push_back(Run{mesh, &t.nodes[k].mtris, tf.size_before[k], cnt ? t.nodes[k].own : Box{}});
      tf.prefix[k + 1] = tf.prefix[k] + (uint32_t)t.nodes[k].mtris.size();
      for (int c : t.nodes[k].children) tf.parent[c] = (int)k, tf.depth[c] = tf.depth[k] + 1;
    }
    for (size_t k = n; k-- > 0;) {
      const DivNode& nd = t.nodes[k];
      tf.end[k] = nd.children.empty() ? (int)k + 1 : tf.end[nd.children.back()];
      Box bb;  // emit_group's bb and tb of a group with a parent and identity children: the same merges
      if (!nd.mtris.empty()) box_merge(bb, tf.run[k].box);
      for (int c : nd.children) box_merge(bb, tf.box[c]);
      tf.box[k] = bb;
      const bool encloses = bb.lo.x >= bb.lo.x && bb.lo.y >= bb.lo.y && bb.lo.z >= bb.lo.z && bb.hi.x <= bb.hi.x &&
                            bb.hi.y <= bb.hi.y && bb.hi.z <= bb.hi.z;  // false if any NaN
      const bool empty = bb.lo.x > bb.hi.x && bb.lo.y > bb.hi.y && bb.lo.z > bb.hi.z;
      tf.flags[k] = (!nd.mtris.empty() ? RT_NODE_ALL_TRIS : 0u) | ((encloses || empty) ? RT_NODE_PRUNABLE : 0u);
    }
    if (std::getenv("RT_TIMING")) std::fprintf(stderr, "flatten: template of %zu nodes prepared in %.1f ms\n", n, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
    return tmpl_flat.emplace(tmpl_id, std::move(tf)).first->second;
  }
  const Run& geometry_run(int mesh, const std::vector<uint32_t>& seq, int tmpl_id, int tmpl_node) {
    if (tmpl_id >= 0) return template_flat(tmpl_id, mesh).run[tmpl_node];
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)mesh;
    for (uint32_t t : seq) {
      h = (h ^ t) * 0xFF51AFD7ED558CCDull;
      h ^= h >> 29;
    }
    std::vector<Run>& bucket = runs[h];
    for (const Run& r : bucket)
      if (r.mesh == mesh && r.seq->size() == seq.size() && std::memcmp(r.seq->data(), seq.data(), seq.size() * 4) == 0)
        return r;
    const HMesh& m = *w.meshes[mesh];
    const uint32_t first = (uint32_t)f.tri_geom.size();
    f.tri_geom.resize(first + seq.size());
    f.tri_normals.resize(first + seq.size());
    Box mb;
    for (size_t k = 0; k < seq.size(); k++) box_merge(mb, m.box[seq[k]]);
    tri_fills.push_back(TriFill{first, mesh, &seq});
    bucket.push_back(Run{mesh, &seq, first, mb});
    return bucket.back();
  }
  uint32_t xform_index(const M4& inv, bool identity) {
    if (identity) return 0;
    std::vector<uint32_t> key(12);
    std::memcpy(key.data(), inv.a, 48);
    auto it = xform_ids.find(key);
    if (it != xform_ids.end()) return it->second;
    RtXform x;
    std::memcpy(x.m, inv.a, 48);
    f.xforms.push_back(x);
    uint32_t id = (uint32_t)f.xforms.size() - 1;
    xform_ids.emplace(std::move(key), id);
    return id;
  }

  void tables() {
    RtXform id{};
    id.m[0] = id.m[5] = id.m[10] = 1.f;
    f.xforms.push_back(id);
    // images
    for (auto& im : w.images) {
      RtImage r{im.w, im.h, (uint64_t)f.texels.size()};
      f.texels.insert(f.texels.end(), im.rgb.begin(), im.rgb.end());
      f.images.push_back(r);
    }
    // uv patterns (value classes feed pattern equality)
    for (auto& u : w.uvs) {
      RtUvPattern r{};
      r.kind = u.kind;
      r.width = u.width;
      r.height = u.height;
      r.image = u.image;
      std::memcpy(r.c, u.c, sizeof(r.c));
      f.uvpatterns.push_back(r);
      KeyBuilder k;
      k.u(u.kind);
      if (u.kind == 0) {
        k.f(u.width), k.f(u.height), k.fs(u.c[0], 6);
      } else if (u.kind == 1) {
        k.fs(u.c[0], 15);
      } else {
        k.u((uint32_t)u.image);
      }
      uv_class_of.push_back((int)uv_classes.get(k));
    }
    for (auto& p : w.patterns) {
      RtPattern r{};
      r.kind = p.kind;
      r.xform = xform_index(p.inv, m4_is_identity(p.xf));
      r.uv_mapping = p.uv_mapping;
      std::memcpy(r.a, p.a, 12);
      std::memcpy(r.b, p.b, 12);
      for (int i = 0; i < 6; i++) r.uv[i] = p.uv[i];
      f.patterns.push_back(r);
      KeyBuilder k;
      k.u(p.kind);
      k.fs(p.xf.a, 16);
      if (p.kind == RTB_PAT_TEXTURE_MAP) {
        k.u(p.uv_mapping);
        k.u((uint32_t)uv_class_of[p.uv[0]]);
        if ((uint32_t)uv_class_of[p.uv[0]] == RT_VCLASS_NEVER) k.nan = true;
      } else if (p.kind == RTB_PAT_CUBE_MAP) {
        for (int i = 0; i < 6; i++) {
          k.u((uint32_t)uv_class_of[p.uv[i]]);
          if ((uint32_t)uv_class_of[p.uv[i]] == RT_VCLASS_NEVER) k.nan = true;
        }
      } else if (p.kind != RTB_PAT_XYZ_RGB) {
        k.fs(p.a, 3), k.fs(p.b, 3);
      }
      pattern_class_of.push_back((int)pattern_classes.get(k));
    }
  }

  // material -> (materials[] index, value class)
  RtbMaterial last_mat{};
  uint32_t last_index = 0, last_class = 0;
  bool have_last = false;
  void material(const RtbMaterial& m, uint32_t* index, uint32_t* vclass) {
    // the same material as the call before (every sub-group of a divided mesh): same answer, unless it holds a NaN
    if (have_last && last_class != RT_VCLASS_NEVER && std::memcmp(&m, &last_mat, sizeof(m)) == 0) {
      *index = last_index;
      *vclass = last_class;
      return;
    }
    material_slow(m, index, vclass);
    last_mat = m;
    last_index = *index;
    last_class = *vclass;
    have_last = true;
  }
  void material_slow(const RtbMaterial& m, uint32_t* index, uint32_t* vclass) {
    KeyBuilder k;
    k.fs(m.color, 3);
    k.f(m.ambient), k.f(m.diffuse), k.f(m.specular), k.f(m.shininess), k.f(m.reflective), k.f(m.transparency),
        k.f(m.refractive_index);
    if (m.pattern >= 0) {
      uint32_t pc = (uint32_t)pattern_class_of[m.pattern];
      if (pc == RT_VCLASS_NEVER) k.nan = true;
      k.u(1), k.u(pc);
    } else {
      k.u(0), k.u(0);
    }
    uint32_t cls = material_classes.get(k);
    *vclass = cls;
    if (cls != RT_VCLASS_NEVER) {
      auto it = material_slot.find(cls);
      if (it != material_slot.end()) {
        *index = it->second;
        return;
      }
    }
    RtMaterial r{};
    std::memcpy(r.color, m.color, 12);
    r.ambient = m.ambient, r.diffuse = m.diffuse, r.specular = m.specular, r.shininess = m.shininess;
    r.reflective = m.reflective, r.transparency = m.transparency, r.refractive_index = m.refractive_index;
    r.pattern = m.pattern;
    f.materials.push_back(r);
    *index = (uint32_t)f.materials.size() - 1;
    if (cls != RT_VCLASS_NEVER) material_slot.emplace(cls, *index);
  }

  uint32_t emit_shape(uint32_t sid, int parent_node) {
    const HShape& s = w.shapes[sid];
    RtShape r{};
    r.kind_flags = s.kind | (s.shadow ? RT_SHAPE_SHADOW : 0u) | (s.closed ? RT_SHAPE_CLOSED : 0u);
    r.xform = xform_index(s.inv, s.identity);
    uint32_t mclass;
    material(s.mat, &r.material, &mclass);
    r.parent = parent_node;
    r.mn = s.mn;
    r.mx = s.mx;
    KeyBuilder k;
    k.u(s.kind), k.fs(s.xf.a, 16), k.u(mclass), k.u(s.shadow ? 1 : 0);
    if (mclass == RT_VCLASS_NEVER) k.nan = true;
    if (s.kind == RTB_CYLINDER || s.kind == RTB_CONE) {
      // -inf/+inf compare equal to themselves; keep their bits
      k.f(s.mn), k.f(s.mx), k.u(s.closed ? 1 : 0);
    } else if (s.kind >= RTB_TRIANGLE) {
      RtTriGeom g;
      RtTriNormals n;
      fill_tri(w.tris[s.tri], &g, &n);
      r.prim = (uint32_t)f.tri_geom.size();
      f.tri_geom.push_back(g);
      f.tri_normals.push_back(n);
      // Triangle == compares p1,p2,p3,e1,e2 and normal / n1..n3 (triangle.rs:14-27): the geometry's class
      r.vclass = k.nan ? RT_VCLASS_NEVER : tri_vclass(triple_id(s.kind, s.xf, mclass, s.shadow), w.tri_gid[s.tri]);
      f.shapes.push_back(r);
      return (uint32_t)f.shapes.size() - 1;
    }
    r.vclass = shape_classes.get(k);
    f.shapes.push_back(r);
    return (uint32_t)f.shapes.size() - 1;
  }

  // csg: leaves in tie order (shape child before csg child, else left then right; csg.rs:32-46),
  // nodes in post-order.  Returns the node index; *lo/*hi = leaf shape range of the subtree.
  Box csg_bounds(int c) {  // csg.rs:87-105
    const HCsg& n = w.csgs[c];
    Box bb;
    auto add = [&](int kind, int id) {
      if (kind == RTB_CHILD_SHAPE)
        box_merge(bb, w.shapes[id].psb);
      else
        box_merge(bb, csg_bounds(id));
    };
    if (n.lkind == RTB_CHILD_CSG && n.rkind == RTB_CHILD_SHAPE) {
      add(n.rkind, n.r);
      add(n.lkind, n.l);
    } else {
      add(n.lkind, n.l);
      add(n.rkind, n.r);
    }
    return bb;
  }
  int emit_csg(int c, uint32_t* lo, uint32_t* hi) {
    const HCsg n = w.csgs[c];
    RtCsgNode r{};
    r.op = n.op;
    uint32_t llo = 0, lhi = 0, rlo = 0, rhi = 0;
    auto child = [&](int kind, int id, int32_t* out, uint32_t* clo, uint32_t* chi, uint32_t flag) {
      if (kind == RTB_CHILD_SHAPE) {
        uint32_t s = emit_shape((uint32_t)id, -1);
        *out = (int32_t)s;
        *clo = s;
        *chi = s + 1;
        r.flags |= flag;
      } else {
        *out = emit_csg(id, clo, chi);
      }
    };
    if (n.lkind == RTB_CHILD_CSG && n.rkind == RTB_CHILD_SHAPE) {
      child(n.rkind, n.r, &r.right, &rlo, &rhi, RT_CSG_RIGHT_IS_SHAPE);
      child(n.lkind, n.l, &r.left, &llo, &lhi, RT_CSG_LEFT_IS_SHAPE);
    } else {
      child(n.lkind, n.l, &r.left, &llo, &lhi, RT_CSG_LEFT_IS_SHAPE);
      child(n.rkind, n.r, &r.right, &rlo, &rhi, RT_CSG_RIGHT_IS_SHAPE);
    }
    r.left_lo = llo;
    r.left_hi = lhi;
    Box bb = csg_bounds(c);
    r.bmin[0] = bb.lo.x, r.bmin[1] = bb.lo.y, r.bmin[2] = bb.lo.z;
    r.bmax[0] = bb.hi.x, r.bmax[1] = bb.hi.y, r.bmax[2] = bb.hi.z;
    *lo = std::min(llo, rlo);
    *hi = std::max(lhi, rhi);
    f.csg_nodes.push_back(r);
    return (int)f.csg_nodes.size() - 1;
  }

  // material index and the value-class tables of a mesh group's triangles (flat, smooth), sized for the mesh
  void mesh_classes(const RtbMaterial& mmat, uint32_t* mat_index, uint32_t* tid_flat, uint32_t* tid_smooth) {
    uint32_t mclass;
    material(mmat, mat_index, &mclass);
    auto tids = mesh_triples.find(mclass);
    if (tids == mesh_triples.end()) {
      const M4 ident = m4_identity();
      tids = mesh_triples.emplace(mclass, std::make_pair(triple_id(RTB_TRIANGLE, ident, mclass, true),
                                                         triple_id(RTB_SMOOTH_TRIANGLE, ident, mclass, true))).first;
    }
    *tid_flat = tids->second.first, *tid_smooth = tids->second.second;
  }
  // value classes of a run of mesh triangles, in the order the shapes are met (the numbering is part of the output);
  // the records themselves are written later (fill_shapes)
  void class_pass(const HMesh& m, const std::vector<uint32_t>& mtris, uint32_t tid_flat, uint32_t tid_smooth) {
    if (tid_flat == RT_VCLASS_NEVER) return;
    uint32_t* map_flat = map_of(tid_flat);
    uint32_t* map_smooth = map_of(tid_smooth);
    for (size_t k = 0; k < mtris.size(); k++) {
      const uint32_t t = mtris[k];
      const uint32_t gid = m.gid[t];
      if (gid != RT_VCLASS_NEVER) {
        const bool smooth = m.smooth[t] == RTB_SMOOTH_TRIANGLE;
        uint32_t& slot = (smooth ? map_smooth : map_flat)[gid];
        if (slot == kUnset) {
          slot = shape_classes.fresh();
          triple_set[smooth ? tid_smooth : tid_flat]++;
        }
      }
    }
  }
  // ... of every node of an untouched template copy at once (what class_pass would do node after node in pre-order).
  // Into tables nothing has been entered into yet -- a material met for the first time, as with every dragon of
  // dragons.yaml -- the classes are the template's distinct geometries in the order a walk first meets them, numbered
  // from the next free class: that order is kept with the template, and the next copy writes it down instead of
  // looking at 871 k triangles again.
  void class_pass_instance(const HMesh& m, const DivTemplate& t, TmplFlat& tf, uint32_t tid_flat, uint32_t tid_smooth) {
    if (tid_flat == RT_VCLASS_NEVER) return;
    if (triple_set[tid_flat] != 0 || triple_set[tid_smooth] != 0) {
      for (const DivNode& nd : t.nodes) class_pass(m, nd.mtris, tid_flat, tid_smooth);
      return;
    }
    if (!tf.first_met_done) {
      const size_t n_gid = w.tri_values.rep.size();
      tf.rank_flat.assign(n_gid, kUnset);
      tf.rank_smooth.assign(n_gid, kUnset);
      for (const DivNode& nd : t.nodes)
        for (uint32_t q : nd.mtris) {
          const uint32_t gid = m.gid[q];
          if (gid == RT_VCLASS_NEVER) continue;
          const bool smooth = m.smooth[q] == RTB_SMOOTH_TRIANGLE;
          uint32_t& r = (smooth ? tf.rank_smooth : tf.rank_flat)[gid];
          if (r != kUnset) continue;
          r = tf.n_first_met++;
          (smooth ? tf.n_smooth : tf.n_flat)++;
        }
      tf.first_met_done = true;
    }
    const uint32_t base = shape_classes.fresh_n(tf.n_first_met);
    triple_lazy[tid_flat] = TripleLazy{tf.rank_flat.data(), base};
    triple_lazy[tid_smooth] = TripleLazy{tf.rank_smooth.data(), base};
    triple_set[tid_flat] += tf.n_flat;
    triple_set[tid_smooth] += tf.n_smooth;
  }
  // The groups below the root of an untouched template copy (HGroup::inst_tmpl): node k of the template is node
  // idx + k here, its shapes follow the root's by prefix[k]; everything else a walk would have found out about it is
  // in the TmplFlat.  `created`: this copy is the one that made the template's runs (a walk would have seen tri_geom
  // grow node by node).
  void emit_instance_below(int mesh, const DivTemplate& t, const TmplFlat& tf, uint32_t idx, size_t shape_base, bool created,
                           uint32_t mat_index, uint32_t tid_flat, uint32_t tid_smooth) {
    const size_t n = t.nodes.size();
    f.nodes.resize(idx + n);
    true_box.resize(idx + n);
    f.shapes.resize(shape_base + tf.prefix[n]);
    const int32_t next_xf = f.nodes[idx].next_xf;
    const uint32_t idrun = f.nodes[idx].idrun;
    const uint32_t tri_now = (uint32_t)f.tri_geom.size();
    for (size_t k = 1; k < n; k++) {
      const DivNode& nd = t.nodes[k];
      RtNode nn{};
      const Box& bb = tf.box[k];
      nn.bmin[0] = bb.lo.x, nn.bmin[1] = bb.lo.y, nn.bmin[2] = bb.lo.z;
      nn.bmax[0] = bb.hi.x, nn.bmax[1] = bb.hi.y, nn.bmax[2] = bb.hi.z;
      nn.xform = 0;
      nn.flags = tf.flags[k];
      nn.first_shape = (uint32_t)(shape_base + tf.prefix[k]);
      nn.n_shapes = (uint32_t)nd.mtris.size();
      nn.skip = idx + (uint32_t)tf.end[k];
      nn.parent = (int32_t)(idx + (uint32_t)tf.parent[k]);
      nn.first_tri = !nd.mtris.empty() ? tf.run[k].first_tri : created ? tf.size_before[k] : tri_now;
      nn.next_xf = next_xf;
      nn.idrun = idrun + (uint32_t)tf.depth[k];
      nn.n_children = (uint32_t)nd.children.size();
      f.nodes[idx + k] = nn;
      true_box[idx + k] = bb;
      if (!nd.mtris.empty())
        shape_fills.push_back(ShapeFill{shape_base + tf.prefix[k], mesh, &nd.mtris, idx + (uint32_t)k, tf.run[k].first_tri, mat_index, tid_flat, tid_smooth});
    }
  }

  // groups depth-first; returns the node index
  uint32_t emit_group(int g, int parent_node, Box* bounds_out, Box* true_out) {
    uint32_t idx = (uint32_t)f.nodes.size();
    f.nodes.emplace_back();
    true_box.emplace_back();
    const HGroup& gr = w.groups[g];
    bool identity = m4_is_identity(gr.xf);
    uint32_t xf = xform_index(gr.inv, identity);
    uint32_t first_shape = (uint32_t)f.shapes.size();
    uint32_t first_tri = (uint32_t)f.tri_geom.size();
    bool all_tris = !gr.shapes.empty() || !gr.mtris.empty();
    Box tb;   // true (conservative) box
    Box bb;   // the reference's make_bounds (groups.rs:176-188), recomputed with final parent state
    for (uint32_t s : gr.shapes) {
      const HShape& hs = w.shapes[s];
      if (!(hs.kind >= RTB_TRIANGLE && hs.identity)) all_tris = false;
      emit_shape(s, (int)idx);
      box_merge(tb, w.true_psb(hs));
      box_merge(bb, hs.psb);
    }
    // the root of an untouched copy of a divide template: the groups below come from the template (emit_instance_below)
    const DivTemplate* inst = nullptr;
    TmplFlat* inst_flat = nullptr;
    bool inst_created = false;
    if (gr.inst_tmpl >= 0 && gr.tmpl_id == gr.inst_tmpl && gr.tmpl_node == 0 && gr.shapes.empty() && gr.mesh >= 0 &&
        (gr.lazy || gr.children.size() == w.div_by_id[gr.inst_tmpl]->nodes[0].children.size())) {
      inst = w.div_by_id[gr.inst_tmpl];
      inst_created = tmpl_flat.find(gr.inst_tmpl) == tmpl_flat.end();
      inst_flat = &template_flat(gr.inst_tmpl, gr.mesh);
    }
    uint32_t mat_index = 0, tid_flat = RT_VCLASS_NEVER, tid_smooth = RT_VCLASS_NEVER;
    if (inst || !gr.mtris.empty()) mesh_classes(gr.mmat, &mat_index, &tid_flat, &tid_smooth);
    if (inst) {
      auto T0 = std::chrono::steady_clock::now();
      class_pass_instance(*w.meshes[gr.mesh], *inst, *inst_flat, tid_flat, tid_smooth);
      t_loop += std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
      if (std::getenv("RT_TIMING")) std::fprintf(stderr, "flatten: instance classes %.1f ms (created %d)\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count(), (int)inst_created);
    }
    if (!gr.mtris.empty()) {  // the triangles of a mesh group, in bulk (a group holds these or shapes, never both)
      const HMesh& m = *w.meshes[gr.mesh];
      auto T1 = std::chrono::steady_clock::now();
      const Run& run = geometry_run(gr.mesh, gr.mtris, gr.tmpl_id, gr.tmpl_node);
      t_run += std::chrono::duration<double>(std::chrono::steady_clock::now() - T1).count();
      first_tri = run.first_tri;
      const size_t base = f.shapes.size();
      f.shapes.resize(base + gr.mtris.size());
      const Box mb = run.box;
      auto T0 = std::chrono::steady_clock::now();
      if (!inst) class_pass(m, gr.mtris, tid_flat, tid_smooth);
      shape_fills.push_back(ShapeFill{base, gr.mesh, &gr.mtris, idx, first_tri, mat_index, tid_flat, tid_smooth});
      t_loop += std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
      box_merge(tb, mb);
      box_merge(bb, mb);
    }
    uint32_t n_shapes = (uint32_t)(gr.shapes.size() + gr.mtris.size());
    {
      RtNode& n = f.nodes[idx];
      n.xform = xf;
      n.first_shape = first_shape;
      n.n_shapes = n_shapes;
      n.parent = parent_node;
      n.first_tri = first_tri;
      n.n_children = (uint32_t)(inst ? inst->nodes[0].children.size() : gr.children.size());
      if (xf) {
        n.next_xf = (int32_t)idx;
        n.idrun = 0;
      } else {
        n.next_xf = parent_node >= 0 ? f.nodes[parent_node].next_xf : -1;
        n.idrun = 1 + (parent_node >= 0 ? f.nodes[parent_node].idrun : 0);
      }
    }
    if (inst) {
      auto T0 = std::chrono::steady_clock::now();
      emit_instance_below(gr.mesh, *inst, *inst_flat, idx, first_shape, inst_created, mat_index, tid_flat, tid_smooth);
      if (std::getenv("RT_TIMING")) std::fprintf(stderr, "flatten: instance nodes %.1f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
      for (int k : inst->nodes[0].children) {
        if (gr.parent < 0) bb = box_transform(bb, gr.xf);
        box_merge(bb, inst_flat->box[k]);
        box_merge(tb, inst_flat->box[k]);
      }
    } else {
      for (int c : gr.children) {
        Box cb, ctb;
        emit_group(c, (int)idx, &cb, &ctb);
        if (gr.parent < 0) bb = box_transform(bb, gr.xf);
        box_merge(bb, cb);
        // child's true box, expressed in this node's space
        const HGroup& cg = w.groups[c];
        if (m4_is_identity(cg.xf))
          box_merge(tb, ctb);
        else
          box_merge(tb, box_transform(ctb, cg.xf));
      }
    }
    if (gr.cached) bb = gr.bounds;  // cache_bounds() result taken at World-add time (scene.rs:197)
    RtNode& n = f.nodes[idx];
    n.bmin[0] = bb.lo.x, n.bmin[1] = bb.lo.y, n.bmin[2] = bb.lo.z;
    n.bmax[0] = bb.hi.x, n.bmax[1] = bb.hi.y, n.bmax[2] = bb.hi.z;
    n.skip = (uint32_t)f.nodes.size();
    bool encloses = tb.lo.x >= bb.lo.x && tb.lo.y >= bb.lo.y && tb.lo.z >= bb.lo.z && tb.hi.x <= bb.hi.x &&
                    tb.hi.y <= bb.hi.y && tb.hi.z <= bb.hi.z;  // false if any NaN
    bool empty = tb.lo.x > tb.hi.x && tb.lo.y > tb.hi.y && tb.lo.z > tb.hi.z;
    n.flags = (all_tris ? RT_NODE_ALL_TRIS : 0u) | ((encloses || empty) ? RT_NODE_PRUNABLE : 0u);
    true_box[idx] = tb;
    *bounds_out = bb;
    *true_out = tb;
    return idx;
  }

  void run() {
    {
      // (a lazy copy is flattened from its template only as long as it is what instantiate() left: anything else
      // gets its groups first -- no reference into w.groups is held yet)
      for (size_t g = 0; g < w.groups.size(); g++) {
        const HGroup& gr = w.groups[g];
        if (gr.lazy && !(gr.inst_tmpl >= 0 && gr.tmpl_id == gr.inst_tmpl && gr.tmpl_node == 0 && gr.shapes.empty() && gr.mesh >= 0))
          w.realize((int)g);
      }
      size_t n_shapes = w.shapes.size(), n_nodes = w.groups.size();
      for (const HGroup& g : w.groups) {
        n_shapes += g.mtris.size();
        if (g.lazy) {
          const DivTemplate& t = *w.div_by_id[g.inst_tmpl];
          n_nodes += t.nodes.size() - 1;
          for (size_t k = 1; k < t.nodes.size(); k++) n_shapes += t.nodes[k].mtris.size();
        }
      }
      f.shapes.reserve(n_shapes);
      f.nodes.reserve(n_nodes);
      true_box.reserve(n_nodes);
      size_t n_tris = w.tris.size();
      for (const auto& m : w.meshes) n_tris += m->tris.size();
      f.tri_geom.reserve(n_tris);
      f.tri_normals.reserve(n_tris);
      runs.reserve(n_nodes);
    }
    tables();
    for (int c : w.world_csgs) {
      uint32_t first = (uint32_t)f.csg_nodes.size();
      uint32_t lo, hi;
      emit_csg(c, &lo, &hi);
      f.csg_roots.push_back(RtCsgRoot{first, (uint32_t)f.csg_nodes.size() - first});
    }
    auto TG = std::chrono::steady_clock::now();
    for (int g : w.world_groups) {
      Box b, t;
      emit_group(g, -1, &b, &t);
    }
    if (std::getenv("RT_TIMING")) std::fprintf(stderr, "flatten: groups %.3f s (triangle loops %.3f, geometry runs %.3f)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - TG).count(), t_loop, t_run);
    uint32_t loose0 = (uint32_t)f.shapes.size();
    for (uint32_t s : w.world_shapes) emit_shape(s, -1);
    auto TF = std::chrono::steady_clock::now();
    run_fills();
    if (std::getenv("RT_TIMING")) std::fprintf(stderr, "flatten: fills %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - TF).count());
    for (auto& l : w.lights) f.lights.push_back(l.l);
    RtScene& v = f.view;
    v = RtScene{};
    v.abi_version = RT_ABI_VERSION;
    v.recursion_limit = (uint32_t)w.recursion_limit;
    v.n_xforms = (uint32_t)f.xforms.size();
    v.n_shapes = (uint32_t)f.shapes.size();
    v.n_tris = (uint32_t)f.tri_geom.size();
    v.n_nodes = (uint32_t)f.nodes.size();
    v.n_csg_nodes = (uint32_t)f.csg_nodes.size();
    v.n_csg_roots = (uint32_t)f.csg_roots.size();
    v.n_materials = (uint32_t)f.materials.size();
    v.n_patterns = (uint32_t)f.patterns.size();
    v.n_uvpatterns = (uint32_t)f.uvpatterns.size();
    v.n_images = (uint32_t)f.images.size();
    v.n_lights = (uint32_t)f.lights.size();
    v.n_loose_shapes = (uint32_t)f.shapes.size() - loose0;
    v.n_texels = f.texels.size();
    v.xforms = f.xforms.data();
    v.shapes = f.shapes.data();
    v.tri_geom = f.tri_geom.data();
    v.tri_normals = f.tri_normals.data();
    v.nodes = f.nodes.data();
    v.csg_nodes = f.csg_nodes.data();
    v.csg_roots = f.csg_roots.data();
    v.materials = f.materials.data();
    v.patterns = f.patterns.data();
    v.uvpatterns = f.uvpatterns.data();
    v.images = f.images.data();
    v.texels = f.texels.data();
    v.lights = f.lights.data();
  }
};

bool make_rt_camera(const RtbCamera* d, RtCamera* c) {  // src/camera.rs:19-43
  c->hsize = d->hsize;
  c->vsize = d->vsize;
  float half_view = tanf(d->field_of_view / 2.f);
  float aspect = (float)d->hsize / (float)d->vsize;
  if (aspect >= 1.f) {
    c->half_width = half_view;
    c->half_height = half_view / aspect;
  } else {
    c->half_width = half_view * aspect;
    c->half_height = half_view;
  }
  c->pixel_size = (c->half_width * 2.f) / (float)d->hsize;
  M4 t = d->use_view ? view_matrix(P3{d->from[0], d->from[1], d->from[2]}, P3{d->to[0], d->to[1], d->to[2]},
                                   P3{d->up[0], d->up[1], d->up[2]})
                     : compose(d->ops, d->nops);
  M4 inv;
  if (!m4_inverse(t, &inv)) return false;
  std::memcpy(c->inv_transform, inv.a, 48);
  return true;
}

bool set_shape_transform(HShape& s, const RtbXfOp* ops, int nops) {
  s.xf = compose(ops, nops);
  s.identity = m4_is_identity(s.xf);
  return m4_inverse(s.xf, &s.inv);
}

RtRenderOptions to_rt_options(const RtbRenderOptions* o) {
  RtRenderOptions r{};
  r.jitter_mode = (o && o->jitter_mode == RTB_JITTER_COUNTER) ? RT_JITTER_COUNTER : RT_JITTER_CONST;
  r.jitter_value = o ? o->jitter_value : 0.5f;
  r.seed = o ? o->seed : 0;
  r.tile_h = 1;
  r.n_parts = 1;
  r.part = 0;
  r.flags = 0;
  return r;
}

}  // namespace

bool RtbWorld::flatten() {
  if (!dirty && flat) return true;
  flat.reset(new Flat());
  Flattener fl(*this, *flat);
  fl.run();
  dirty = false;
  return true;
}

extern "C" {

RtbWorld* rth_world_new(void) { return new RtbWorld(); }
void rth_world_free(RtbWorld* w) { delete w; }
const char* rth_last_error(const RtbWorld* w) { return w->error.c_str(); }

int rth_uv_checker(RtbWorld* w, float width, float height, const float a[3], const float b[3]) {
  HUv u;
  u.kind = 0;
  u.width = width;
  u.height = height;
  std::memcpy(u.c[0], a, 12);
  std::memcpy(u.c[1], b, 12);
  w->uvs.push_back(u);
  return (int)w->uvs.size() - 1;
}
int rth_uv_align_check(RtbWorld* w, const float c[15]) {
  HUv u;
  u.kind = 1;
  std::memcpy(u.c, c, 60);
  w->uvs.push_back(u);
  return (int)w->uvs.size() - 1;
}
int rth_uv_image(RtbWorld* w, uint32_t width, uint32_t height, const float* rgb) {
  HImage im;
  im.w = width;
  im.h = height;
  im.rgb.assign(rgb, rgb + (size_t)width * height * 3);
  w->images.push_back(std::move(im));
  HUv u;
  u.kind = 2;
  u.image = (int)w->images.size() - 1;
  w->uvs.push_back(u);
  return (int)w->uvs.size() - 1;
}
int rth_pattern_new(RtbWorld* w, int kind, const float colors[6], const RtbXfOp* ops, int nops, int uv_mapping,
                    const int32_t* uvpats) {
  HPattern p;
  p.kind = kind;
  if (colors) {
    std::memcpy(p.a, colors, 12);
    std::memcpy(p.b, colors + 3, 12);
  }
  p.xf = compose(ops, nops);
  if (!m4_inverse(p.xf, &p.inv)) return fail(w, "pattern transform is not invertible");
  p.uv_mapping = uv_mapping;
  int nuv = kind == RTB_PAT_TEXTURE_MAP ? 1 : (kind == RTB_PAT_CUBE_MAP ? 6 : 0);
  for (int i = 0; i < nuv; i++) {
    if (!uvpats || uvpats[i] < 0 || uvpats[i] >= (int)w->uvs.size()) return fail(w, "bad uv pattern handle");
    p.uv[i] = uvpats[i];
  }
  w->patterns.push_back(p);
  return (int)w->patterns.size() - 1;
}

int rth_shape_new(RtbWorld* w, int kind, const RtbXfOp* ops, int nops, const RtbMaterial* material, int shadow,
                  const float* params, int nparams) {
  if (kind < RTB_SPHERE || kind > RTB_SMOOTH_TRIANGLE) return fail(w, "bad shape kind");
  HShape s;
  s.kind = (uint8_t)kind;
  s.shadow = shadow != 0;
  s.mat = material ? *material : default_material();
  if (s.mat.pattern >= (int)w->patterns.size()) return fail(w, "bad pattern handle");
  if (kind == RTB_TRIANGLE || kind == RTB_SMOOTH_TRIANGLE) {
    int need = kind == RTB_TRIANGLE ? 9 : 18;
    if (!params || nparams < need) return fail(w, "triangle params missing");
    HTri t{};
    t.p1 = P3{params[0], params[1], params[2]};
    t.p2 = P3{params[3], params[4], params[5]};
    t.p3 = P3{params[6], params[7], params[8]};
    t.e1 = sub3(t.p2, t.p1);
    t.e2 = sub3(t.p3, t.p1);
    if (kind == RTB_TRIANGLE) {
      t.n1 = normalize3(cross3(t.e2, t.e1));  // triangle.rs:35
    } else {
      t.n1 = P3{params[9], params[10], params[11]};
      t.n2 = P3{params[12], params[13], params[14]};
      t.n3 = P3{params[15], params[16], params[17]};
    }
    w->tris.push_back(t);
    w->tri_gid.push_back(w->tri_values.get(w->tris.back(), kind == RTB_SMOOTH_TRIANGLE));
    s.tri = (int)w->tris.size() - 1;
  } else if ((kind == RTB_CYLINDER || kind == RTB_CONE) && params && nparams >= 3) {
    s.mn = params[0];
    s.mx = params[1];
    s.closed = params[2] != 0.f;
  }
  if (!set_shape_transform(s, ops, nops)) return fail(w, "shape transform is not invertible");
  s.psb = box_transform(w->local_bounds(s), s.xf);  // shapes.rs:318-320
  w->shapes.push_back(s);
  return (int)w->shapes.size() - 1;
}

int rth_group_new(RtbWorld* w, const RtbXfOp* ops, int nops) {
  int g = w->new_group();
  w->groups[g].xf = compose(ops, nops);
  w->no_longer_template_bounds(g);
  if (!m4_inverse(w->groups[g].xf, &w->groups[g].inv)) return fail(w, "group transform is not invertible");
  w->touch();
  return g;
}
int rth_group_set_transform(RtbWorld* w, int g, const RtbXfOp* ops, int nops) {
  if (g < 0 || g >= (int)w->groups.size()) return fail(w, "bad group handle");
  w->groups[g].xf = compose(ops, nops);
  // (the instance root's own transform is its own business: what must stay as it is lies below it)
  if (w->groups[g].parent >= 0) w->no_longer_template_copy(w->groups[g].parent);
  if (!m4_inverse(w->groups[g].xf, &w->groups[g].inv)) return fail(w, "group transform is not invertible");
  w->touch();
  return 0;
}
int rth_group_add_shape(RtbWorld* w, int g, int shape) {
  if (g < 0 || g >= (int)w->groups.size() || shape < 0 || shape >= (int)w->shapes.size())
    return fail(w, "bad handle");
  w->realize(g);
  w->materialise(g);                             // a mesh group that also gets other shapes holds shapes only
  w->no_longer_template_bounds(g);
  uint32_t s = w->clone_shape((uint32_t)shape);  // groups.rs:80-85: the group owns a copy
  w->shapes[s].parent = g;
  w->groups[g].shapes.push_back(s);
  w->touch();
  return 0;
}
int rth_group_add_child(RtbWorld* w, int g, int child) {
  if (g < 0 || g >= (int)w->groups.size() || child < 0 || child >= (int)w->groups.size())
    return fail(w, "bad group handle");
  if (w->groups[child].parent >= 0) w->no_longer_template_copy(w->groups[child].parent);
  w->realize(g);
  w->attach_child(g, child);
  w->no_longer_template_bounds(g);
  w->touch();
  return 0;
}
int rth_group_add_mesh(RtbWorld* w, int g, const RtbMaterial* material, const float* positions, int npositions,
                       const float* normals, int nnormals, const int32_t* vidx, const int32_t* nidx, int ntriangles) {
  if (g < 0 || g >= (int)w->groups.size()) return fail(w, "bad group handle");
  if (npositions < 0 || nnormals < 0 || ntriangles < 0) return fail(w, "negative mesh size");
  if (ntriangles >= 65536) w->ctx_early();
  const bool has_n = normals && nidx;
  // the same arrays handed in again (scene.rs:593-599 re-parses the file for every `obj:` use): one HMesh
  // (hashed and compared in 1 MB pieces on threads: dragons.yaml hands the same 31 MB in six times)
  const bool timing = std::getenv("RT_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto fnv = [](uint64_t h, const void* p, size_t n) {
    const char* q = static_cast<const char*>(p);
    for (size_t i = 0; i + 8 <= n; i += 8) {
      uint64_t v;
      std::memcpy(&v, q + i, 8);
      h = (h ^ v) * 0x100000001B3ull;
      h ^= h >> 31;
    }
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (size_t i = n & ~(size_t)7; i < n; i++) h = (h ^ c[i]) * 0x100000001B3ull;
    return h;
  };
  constexpr size_t kPiece = (size_t)1 << 20;
  auto on_pieces = [](size_t n_pieces, const std::function<void(size_t)>& f) {
    const unsigned nt = n_pieces >= 4 ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
    if (nt == 1) {
      for (size_t k = 0; k < n_pieces; k++) f(k);
      return;
    }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++)
      pool.emplace_back([&, t] {
        for (size_t k = t; k < n_pieces; k += nt) f(k);
      });
    for (auto& th : pool) th.join();
  };
  auto hash_array = [&](uint64_t h, const void* p, size_t n) {
    const size_t n_pieces = (n + kPiece - 1) / kPiece;
    std::vector<uint64_t> part(n_pieces);
    on_pieces(n_pieces, [&](size_t k) {
      part[k] = fnv(1469598103934665603ull + k, static_cast<const char*>(p) + k * kPiece, std::min(kPiece, n - k * kPiece));
    });
    return fnv(h ^ n, part.data(), part.size() * 8);
  };
  auto same_bytes = [&](const void* a, const void* b, size_t n) {
    const size_t n_pieces = (n + kPiece - 1) / kPiece;
    std::vector<uint8_t> differ(n_pieces, 0);
    on_pieces(n_pieces, [&](size_t k) {
      differ[k] = std::memcmp(static_cast<const char*>(a) + k * kPiece, static_cast<const char*>(b) + k * kPiece,
                              std::min(kPiece, n - k * kPiece)) != 0;
    });
    for (uint8_t d : differ)
      if (d) return false;
    return true;
  };
  uint64_t h = 1469598103934665603ull ^ (uint64_t)ntriangles ^ ((uint64_t)npositions << 32);
  h = hash_array(h, positions, (size_t)npositions * 12);
  h = hash_array(h, vidx, (size_t)ntriangles * 12);
  if (has_n) {
    h = hash_array(h, normals, (size_t)nnormals * 12);
    h = hash_array(h, nidx, (size_t)ntriangles * 12);
  }
  auto stage = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "group_add_mesh: %-24s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  stage("hash inputs");
  int mesh = -1;
  for (size_t k = 0; k < w->meshes.size() && mesh < 0; k++) {
    const HMesh& m = *w->meshes[k];
    if (m.hash == h && m.in_pos.size() == (size_t)npositions * 3 && m.in_vidx.size() == (size_t)ntriangles * 3 &&
        m.in_nrm.size() == (has_n ? (size_t)nnormals * 3 : 0) && m.in_nidx.size() == (has_n ? (size_t)ntriangles * 3 : 0) &&
        same_bytes(m.in_pos.data(), positions, m.in_pos.size() * 4) && same_bytes(m.in_vidx.data(), vidx, m.in_vidx.size() * 4) &&
        (!has_n || (same_bytes(m.in_nrm.data(), normals, m.in_nrm.size() * 4) && same_bytes(m.in_nidx.data(), nidx, m.in_nidx.size() * 4))))
      mesh = (int)k;
  }
  if (mesh < 0) {
    std::unique_ptr<HMesh> mp(new HMesh());
    HMesh& m = *mp;
    m.hash = h;
    m.in_pos.assign(positions, positions + (size_t)npositions * 3);
    m.in_vidx.assign(vidx, vidx + (size_t)ntriangles * 3);
    if (has_n) {
      m.in_nrm.assign(normals, normals + (size_t)nnormals * 3);
      m.in_nidx.assign(nidx, nidx + (size_t)ntriangles * 3);
    }
    stage("copy inputs");
    m.tris.resize(ntriangles);
    m.smooth.resize(ntriangles);
    m.box.resize(ntriangles);
    m.gid.resize(ntriangles);
    const M4 ident = m4_identity();
    for (int t = 0; t < ntriangles * 3; t++)
      if (vidx[t] < 0 || vidx[t] >= npositions) return fail(w, "mesh position index out of range");
    if (has_n)
      for (int t = 0; t < ntriangles * 3; t++)
        if (nidx[t] >= nnormals) return fail(w, "mesh normal index out of range");
    const int n_threads = ntriangles >= 65536 ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
    auto work = [&](int t_begin, int t_end) {
    for (int t = t_begin; t < t_end; t++) {
      HTri tr{};
      P3* ps[3] = {&tr.p1, &tr.p2, &tr.p3};
      for (int c = 0; c < 3; c++) {
        int vi = vidx[3 * t + c];
        *ps[c] = P3{positions[3 * vi], positions[3 * vi + 1], positions[3 * vi + 2]};
      }
      tr.e1 = sub3(tr.p2, tr.p1);  // Triangle::new (triangle.rs:32-48)
      tr.e2 = sub3(tr.p3, tr.p1);
      bool smooth = has_n;
      if (smooth) {
        P3* ns[3] = {&tr.n1, &tr.n2, &tr.n3};
        for (int c = 0; c < 3; c++) {
          int ni = nidx[3 * t + c];
          if (ni < 0) {
            smooth = false;
            break;
          }
          *ns[c] = P3{normals[3 * ni], normals[3 * ni + 1], normals[3 * ni + 2]};
        }
      }
      if (!smooth) {
        tr.n1 = normalize3(cross3(tr.e2, tr.e1));
        tr.n2 = tr.n3 = P3{0.f, 0.f, 0.f};
      }
      m.tris[t] = tr;
      m.smooth[t] = smooth ? RTB_SMOOTH_TRIANGLE : RTB_TRIANGLE;
      Box lb;  // triangle.rs:117-123, then shapes.rs:318-320 with the identity transform
      box_add(lb, tr.p1);
      box_add(lb, tr.p2);
      box_add(lb, tr.p3);
      m.box[t] = box_transform(lb, ident);
    }
    };
    if (n_threads == 1) {
      work(0, ntriangles);
    } else {
      std::vector<std::thread> pool;
      const int per = (ntriangles + n_threads - 1) / n_threads;
      for (int k = 0; k < n_threads; k++) pool.emplace_back(work, std::min(ntriangles, k * per), std::min(ntriangles, (k + 1) * per));
      for (auto& th : pool) th.join();
    }
    stage("triangles and boxes");
    w->tri_values.reserve((size_t)ntriangles);
    {
      std::vector<uint64_t> hs(ntriangles);
      std::vector<uint8_t> nans(ntriangles);
      auto hash_work = [&](int b, int e) {
        for (int t = b; t < e; t++) {
          bool nan;
          hs[t] = TriValueTable::value_hash(m.tris[t], m.smooth[t] == RTB_SMOOTH_TRIANGLE, &nan);
          nans[t] = nan;
        }
      };
      if (n_threads == 1) {
        hash_work(0, ntriangles);
      } else {
        std::vector<std::thread> pool;
        const int per = (ntriangles + n_threads - 1) / n_threads;
        for (int k = 0; k < n_threads; k++) pool.emplace_back(hash_work, std::min(ntriangles, k * per), std::min(ntriangles, (k + 1) * per));
        for (auto& th : pool) th.join();
      }
      w->tri_values.get_batch(m.tris, m.smooth.data(), hs.data(), nans.data(), (uint32_t)ntriangles, m.gid.data(), (unsigned)n_threads);
    }
    stage("geometry value classes");
    w->meshes.push_back(std::move(mp));
    mesh = (int)w->meshes.size() - 1;
  }
  w->realize(g);
  int child = w->new_group();
  HGroup& cg = w->groups[child];
  cg.mesh = mesh;
  cg.mmat = material ? *material : default_material();
  if (cg.mmat.pattern >= (int)w->patterns.size()) return fail(w, "bad pattern handle");
  cg.mtris.resize(ntriangles);
  for (int t = 0; t < ntriangles; t++) cg.mtris[t] = (uint32_t)t;
  cg.fresh_mesh = true;
  w->attach_child(g, child);
  w->no_longer_template_bounds(g);
  w->touch();
  return child;
}
int rth_group_divide(RtbWorld* w, int g, int threshold) {
  if (g < 0 || g >= (int)w->groups.size()) return fail(w, "bad group handle");
  w->divide(g, threshold);
  w->touch();
  return 0;
}

int rth_csg_new(RtbWorld* w, int op, int lkind, int l, int rkind, int r) {
  HCsg c;
  c.op = op;
  auto take = [&](int kind, int id, int* out) -> bool {
    if (kind == RTB_CHILD_SHAPE) {
      if (id < 0 || id >= (int)w->shapes.size()) return false;
      *out = (int)w->clone_shape((uint32_t)id);  // csg.rs:14-24: Box<Shape> owned by the node
    } else {
      if (id < 0 || id >= (int)w->csgs.size()) return false;
      *out = id;
    }
    return true;
  };
  c.lkind = lkind;
  c.rkind = rkind;
  if (!take(lkind, l, &c.l) || !take(rkind, r, &c.r)) return fail(w, "bad csg child handle");
  w->csgs.push_back(c);
  return (int)w->csgs.size() - 1;
}

int rth_world_add_shape(RtbWorld* w, int shape) {
  if (shape < 0 || shape >= (int)w->shapes.size()) return fail(w, "bad shape handle");
  uint32_t s = w->clone_shape((uint32_t)shape);
  w->world_shapes.push_back(s);
  w->touch();
  return (int)w->world_shapes.size() - 1;
}
int rth_world_add_group(RtbWorld* w, int g, int cache_bounds) {
  if (g < 0 || g >= (int)w->groups.size()) return fail(w, "bad group handle");
  if (cache_bounds) {  // groups.rs:190-196
    const auto t0 = std::chrono::steady_clock::now();
    w->groups[g].bounds = w->make_bounds(g);
    w->groups[g].cached = true;
    if (std::getenv("RT_TIMING"))
      std::fprintf(stderr, "world_add_group: bounds %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  w->world_groups.push_back(g);
  w->touch();
  return 0;
}
int rth_world_add_csg(RtbWorld* w, int c) {
  if (c < 0 || c >= (int)w->csgs.size()) return fail(w, "bad csg handle");
  w->world_csgs.push_back(c);
  w->touch();
  return 0;
}
int rth_world_add_point_light(RtbWorld* w, const float position[3], const float intensity[3]) {
  HLight l;
  l.l.kind = RT_LIGHT_POINT;
  l.l.usteps = l.l.vsteps = 1;
  std::memcpy(l.l.position, position, 12);
  std::memcpy(l.l.intensity, intensity, 12);
  w->lights.push_back(l);
  w->touch();
  return (int)w->lights.size() - 1;
}
int rth_world_add_area_light(RtbWorld* w, const float corner[3], const float fu[3], int usteps, const float fv[3],
                             int vsteps, const float intensity[3]) {
  if (usteps < 1 || vsteps < 1) return fail(w, "area light steps must be >= 1");
  HLight l;  // lights.rs:83-103
  l.l.kind = RT_LIGHT_AREA;
  l.l.usteps = (uint32_t)usteps;
  l.l.vsteps = (uint32_t)vsteps;
  for (int i = 0; i < 3; i++) {
    l.l.corner[i] = corner[i];
    l.l.position[i] = corner[i] + (fu[i] + fv[i]) / 2.f;
    l.l.uvec[i] = fu[i] / (float)usteps;
    l.l.vvec[i] = fv[i] / (float)vsteps;
    l.l.intensity[i] = intensity[i];
  }
  w->lights.push_back(l);
  w->touch();
  return (int)w->lights.size() - 1;
}
int rth_world_set_recursion_limit(RtbWorld* w, int limit) {
  if (limit < 0 || limit > 255) return fail(w, "recursion_limit is a u8 in the reference");
  w->recursion_limit = limit;
  w->touch();
  return 0;
}

// ---- extras beyond rt_builder.h: the flatten / device plumbing the harness needs ---------
int rth_set_device(RtbWorld* w, int device) {
  if (device != w->device) w->drop_device();
  w->device = device;
  return 0;
}
// Camera::render on several GPUs: devices[0] is the world's device (rt_color_at, aux outputs), the others take a
// share of the rows of every frame rendered without aux outputs (rt_render_multi).  An ordinal may be listed more than
// once (several contexts on one GPU: what a test on a one-GPU box does).
int rth_set_devices(RtbWorld* w, const int* devices, int n) {
  if (!devices || n < 1 || n > (int)RT_MULTI_MAX) return fail(w, "rth_set_devices: 1 .. RT_MULTI_MAX devices");
  w->drop_device();
  w->device = devices[0];
  w->more.clear();
  for (int k = 1; k < n; k++) {
    RtbWorld::MoreGpu m;
    m.device = devices[k];
    w->more.push_back(m);
  }
  return 0;
}
const RtScene* rth_flatten(RtbWorld* w) {
  if (!w->flatten()) return nullptr;
  return &w->flat->view;
}
int rth_make_camera(RtbWorld* w, const RtbCamera* cam, RtCamera* out) {
  if (!make_rt_camera(cam, out)) return fail(w, "camera transform is not invertible");
  return 0;
}
// Ensures the world is flattened and resident on its device; returns the C-ABI handles.
int rth_device_scene(RtbWorld* w, RtContext** ctx, RtDeviceScene** scene) {
  if (!w->flatten()) return fail(w, "flatten failed");
  w->ctx_collect();
  if (!w->ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc = rt_context_create(w->device, &w->ctx);
    if (std::getenv("RT_TIMING")) std::fprintf(stderr, "rt_context_create: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rc != RT_OK) return fail(w, "rt_context_create failed (is a GPU visible?)", rc);
  }
  if (!w->dscene) {
    int rc = rt_scene_create(w->ctx, &w->flat->view, &w->dscene);
    if (rc != RT_OK) return fail(w, std::string("rt_scene_create: ") + rt_last_error(w->ctx), rc);
  }
  for (RtbWorld::MoreGpu& m : w->more) {
    if (!m.ctx) {
      int rc = rt_context_create(m.device, &m.ctx);
      if (rc != RT_OK) return fail(w, "rt_context_create failed for GPU " + std::to_string(m.device), rc);
    }
    if (!m.dscene) {
      int rc = rt_scene_create(m.ctx, &w->flat->view, &m.dscene);
      if (rc != RT_OK) return fail(w, std::string("rt_scene_create: ") + rt_last_error(m.ctx), rc);
    }
  }
  if (ctx) *ctx = w->ctx;
  if (scene) *scene = w->dscene;
  return 0;
}

// Camera::render(&World) -> Canvas (src/camera.rs:66-84): flatten -> upload -> HIP kernels.
int rth_render(RtbWorld* w, const RtbCamera* cam, const RtbRenderOptions* opts, float* out_rgb, RtbAux* aux,
               RtbStats* stats) {
  RtCamera rc;
  if (!make_rt_camera(cam, &rc)) return fail(w, "camera transform is not invertible");
  auto t0 = std::chrono::steady_clock::now();
  int e = rth_device_scene(w, nullptr, nullptr);
  if (e != 0) return e;
  RtRenderOptions ro = to_rt_options(opts);
  RtAux ra{};
  if (aux) {
    ra.hit_mask = aux->hit_mask;
    ra.object_id = aux->object_id;
    ra.hit_t = aux->hit_t;
  }
  RtStats rs{};
  int r;
  if (!w->more.empty() && !aux) {  // every GPU renders its stripes of rows and copies them into out_rgb
    std::vector<RtContext*> ctxs{w->ctx};
    std::vector<const RtDeviceScene*> scenes{w->dscene};
    for (RtbWorld::MoreGpu& m : w->more) {
      ctxs.push_back(m.ctx);
      scenes.push_back(m.dscene);
    }
    std::vector<RtStats> per(ctxs.size());
    r = rt_render_multi(ctxs.data(), scenes.data(), (uint32_t)ctxs.size(), &rc, &ro, out_rgb, per.data());
    for (const RtStats& p : per) {
      rs.primary += p.primary;
      rs.shadow += p.shadow;
      rs.reflect += p.reflect;
      rs.refract += p.refract;
      rs.overflow += p.overflow;
    }
  } else {
    r = rt_render(w->ctx, w->dscene, &rc, &ro, out_rgb, aux ? &ra : nullptr, &rs);
  }
  if (stats) {
    stats->primary = rs.primary;
    stats->shadow = rs.shadow;
    stats->reflect = rs.reflect;
    stats->refract = rs.refract;
    stats->overflow = rs.overflow;
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (r != RT_OK) return fail(w, std::string("rt_render: ") + rt_last_error(w->ctx), r);
  return 0;
}

int rth_color_at(RtbWorld* w, const RtbRenderOptions* opts, const float* origins, const float* directions, int n,
                 int remaining, float* out_rgb, float* hit_t, int32_t* hit_id, float* hit_uv) {
  int e = rth_device_scene(w, nullptr, nullptr);
  if (e != 0) return e;
  RtRenderOptions ro = to_rt_options(opts);
  int r = rt_color_at(w->ctx, w->dscene, &ro, origins, directions, (uint32_t)n, (uint32_t)remaining, out_rgb, hit_t,
                      hit_id, hit_uv, nullptr);
  if (r != RT_OK) return fail(w, std::string("rt_color_at: ") + rt_last_error(w->ctx), r);
  return 0;
}

int rth_ray_for_pixel(RtbWorld* w, const RtbCamera* cam, uint32_t px, uint32_t py, float out[6], float* pixel_size) {
  // Host mirror of Camera::ray_for_pixel (camera.rs:45-64), used by host-logic tests only; the
  // render kernel generates its own rays.
  RtCamera c;
  if (!make_rt_camera(cam, &c)) return fail(w, "camera transform is not invertible");
  float xoffset = ((float)px + 0.5f) * c.pixel_size, yoffset = ((float)py + 0.5f) * c.pixel_size;
  float wx = c.half_width - xoffset, wy = c.half_height - yoffset;
  M4 inv = m4_identity();
  std::memcpy(inv.a, c.inv_transform, 48);
  P3 pixel = m4_point(inv, P3{wx, wy, -1.f}), origin = m4_point(inv, P3{0.f, 0.f, 0.f});
  P3 d = normalize3(sub3(pixel, origin));
  out[0] = origin.x, out[1] = origin.y, out[2] = origin.z;
  out[3] = d.x, out[4] = d.y, out[5] = d.z;
  if (pixel_size) *pixel_size = c.pixel_size;
  return 0;
}

}  // extern "C"
