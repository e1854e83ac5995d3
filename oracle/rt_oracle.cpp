// rt_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never on the product path).
//
// A C++ restatement of the hot path of lerouxrgd/raytracer v1.4.0:
//   Camera::render(&World) -> Canvas and everything it calls per pixel.
// Every function cites the reference file:line it follows (paths relative to the reference
// root).  It deliberately keeps the reference's structure (recursive color_at, sorted
// intersection lists, by-value shape equality, per-level normal renormalisation) instead of
// anything GPU-shaped, so that it is an independent check on raytracer_amd/csrc.
//
// Parity pin: tests/test_oracle_kat.py restates the reference's own unit-test known answers
//  against this file, and tests/test_oracle_golden.py compares full renders with
// the reference's committed renders (samples/rendered/*.png, copied as fixtures to
// tests/golden/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
// (no FMA contraction, no reassociation: rustc never contracts).
//
// Deviations from a literal restatement (all result-preserving):
//  * matrix inverses are computed once when a transform is set instead of per call
//    (src/shapes.rs:205, src/groups.rs:130, src/camera.rs:58) — same bits, same function;
//  * Group::bounds() of un-cached groups (src/groups.rs:166-174) is memoised per render in
//    `frozen_bounds` — the tree is immutable while rendering, so the value is identical;
//  * Intersection holds a pointer to the stored Shape instead of a by-value copy
//    (src/intersections.rs:12-17); equality still compares by value (shape_eq);
//  * area-light jitter: thread_rng() (src/lights.rs:114-120) is unreproducible by design; the
//    oracle offers const / counter-hash / cycling-sequence modes (DESIGN.md "Area lights").

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#define RTB_PREFIX(name) orc_##name
#include "../include/rt_builder.h"

namespace {

constexpr float INF = std::numeric_limits<float>::infinity();
constexpr float F32_MAX = std::numeric_limits<float>::max();
constexpr float PI_F = 3.14159265358979323846f;  // std::f32::consts::PI

// ---------------------------------------------------------------------------------------
// tuples.rs
// ---------------------------------------------------------------------------------------
struct Tuple {
  float v[4];
};
inline Tuple tup(float x, float y, float z, float w) { return Tuple{{x, y, z, w}}; }
inline Tuple point(float x, float y, float z) { return tup(x, y, z, 1.f); }   // tuples.rs:115
inline Tuple vector(float x, float y, float z) { return tup(x, y, z, 0.f); }  // tuples.rs:208
inline Tuple color(float r, float g, float b) { return tup(r, g, b, 0.f); }   // tuples.rs:315
inline Tuple add(Tuple a, Tuple b) {  // tuples.rs:36-46
  return tup(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]);
}
inline Tuple sub(Tuple a, Tuple b) {  // tuples.rs:49-59
  return tup(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]);
}
inline Tuple neg(Tuple a) { return tup(-a.v[0], -a.v[1], -a.v[2], -a.v[3]); }  // tuples.rs:62-67
inline Tuple mul(Tuple a, Tuple b) {                                           // tuples.rs:70-80
  return tup(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]);
}
inline Tuple scale(float s, Tuple a) {  // tuples.rs:83-93 (f32 * Tuple)
  return tup(s * a.v[0], s * a.v[1], s * a.v[2], s * a.v[3]);
}
inline Tuple divs(Tuple a, float s) {  // tuples.rs:96-106
  return tup(a.v[0] / s, a.v[1] / s, a.v[2] / s, a.v[3] / s);
}
inline bool teq(Tuple a, Tuple b) {  // derived PartialEq on [f32;4]
  return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2] && a.v[3] == b.v[3];
}
inline float magnitude(Tuple a) {  // tuples.rs:232-234
  return std::sqrt(a.v[0] * a.v[0] + a.v[1] * a.v[1] + a.v[2] * a.v[2] + a.v[3] * a.v[3]);
}
inline Tuple normalize(Tuple a) { return divs(a, magnitude(a)); }  // tuples.rs:236-238
inline float dot(Tuple a, Tuple b) {                               // tuples.rs:240-242
  return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2] + a.v[3] * b.v[3];
}
inline Tuple cross(Tuple a, Tuple b) {  // tuples.rs:244-250
  return vector(a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2],
                a.v[0] * b.v[1] - a.v[1] * b.v[0]);
}
inline Tuple reflect(Tuple in, Tuple normal) {  // tuples.rs:252-254
  return sub(in, scale(2.f * dot(in, normal), normal));
}
inline Tuple cscale(Tuple c, float s) { return scale(s, c); }  // tuples.rs:420-426 (Color * f32 = rhs * self.0)

// ---------------------------------------------------------------------------------------
// ordered-float 3.7.0 total order: NaN is greatest and equal to itself.
// ---------------------------------------------------------------------------------------
inline int of_cmp(float a, float b) {
  if (a < b) return -1;
  if (a > b) return 1;
  if (a == b) return 0;
  bool an = std::isnan(a), bn = std::isnan(b);
  if (an && bn) return 0;
  return an ? 1 : -1;
}
inline bool of_eq(float a, float b) { return std::isnan(a) ? std::isnan(b) : a == b; }
inline float of_max(float a, float b) { return of_cmp(a, b) > 0 ? a : b; }  // cmp::max: rhs on ties
inline float of_min(float a, float b) { return of_cmp(a, b) > 0 ? b : a; }  // cmp::min: lhs on ties

// ---------------------------------------------------------------------------------------
// matrices.rs
// ---------------------------------------------------------------------------------------
struct Mat4 {
  float m[4][4];
};
inline Mat4 identity() {  // matrices.rs:83-93
  Mat4 r{};
  for (int i = 0; i < 4; i++) r.m[i][i] = 1.f;
  return r;
}
inline Mat4 matmul(const Mat4& a, const Mat4& b) {  // matrices.rs:68-80 (sum from 0, left to right)
  Mat4 r;
  for (int row = 0; row < 4; row++)
    for (int col = 0; col < 4; col++) {
      float s = 0.f;
      for (int k = 0; k < 4; k++) s = s + a.m[row][k] * b.m[k][col];
      r.m[row][col] = s;
    }
  return r;
}
inline Tuple mat_tuple(const Mat4& a, Tuple t) {  // matrices.rs:200-218 (4x4 * 4x1)
  Tuple r;
  for (int row = 0; row < 4; row++) {
    float s = 0.f;
    for (int k = 0; k < 4; k++) s = s + a.m[row][k] * t.v[k];
    r.v[row] = s;
  }
  return r;
}
inline Tuple mat_point(const Mat4& a, Tuple p) {  // matrices.rs:220-227 (w forced to 1)
  Tuple t = mat_tuple(a, p);
  return point(t.v[0], t.v[1], t.v[2]);
}
inline Tuple mat_vector(const Mat4& a, Tuple p) {  // matrices.rs:229-236 (w forced to 0)
  Tuple t = mat_tuple(a, p);
  return vector(t.v[0], t.v[1], t.v[2]);
}
inline Mat4 transpose(const Mat4& a) {  // matrices.rs:24-32
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.m[j][i] = a.m[i][j];
  return r;
}
inline bool mat_eq(const Mat4& a, const Mat4& b) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (!(a.m[i][j] == b.m[i][j])) return false;
  return true;
}
inline float det2(const float m[2][2]) { return m[0][0] * m[1][1] - m[1][0] * m[0][1]; }  // matrices.rs:100-104
inline float minor3(const float m[3][3], int row, int col) {  // matrices.rs:107-127
  float s[2][2];
  for (int i = 0, ii = 0; i < 3; i++) {
    if (i == row) continue;
    for (int j = 0, jj = 0; j < 3; j++) {
      if (j == col) continue;
      s[ii][jj] = m[i][j];
      jj++;
    }
    ii++;
  }
  return det2(s);
}
inline float cofactor3(const float m[3][3], int row, int col) {  // matrices.rs:129-135
  float mi = minor3(m, row, col);
  return ((row + col) % 2 == 0) ? mi : -mi;
}
inline float det3(const float m[3][3]) {  // matrices.rs:137-143
  float det = 0.f;
  for (int j = 0; j < 3; j++) det += m[0][j] * cofactor3(m, 0, j);
  return det;
}
inline float minor4(const Mat4& a, int row, int col) {  // matrices.rs:147-167
  float s[3][3];
  for (int i = 0, ii = 0; i < 4; i++) {
    if (i == row) continue;
    for (int j = 0, jj = 0; j < 4; j++) {
      if (j == col) continue;
      s[ii][jj] = a.m[i][j];
      jj++;
    }
    ii++;
  }
  return det3(s);
}
inline float cofactor4(const Mat4& a, int row, int col) {  // matrices.rs:169-175
  float mi = minor4(a, row, col);
  return ((row + col) % 2 == 0) ? mi : -mi;
}
inline float det4(const Mat4& a) {  // matrices.rs:177-183
  float det = 0.f;
  for (int j = 0; j < 4; j++) det += a.m[0][j] * cofactor4(a, 0, j);
  return det;
}
inline bool inverse(const Mat4& a, Mat4* out) {  // matrices.rs:185-197
  float det = det4(a);
  if (det == 0.f) return false;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) out->m[j][i] = cofactor4(a, i, j) / det;
  return true;
}

// ---------------------------------------------------------------------------------------
// transformations.rs
// ---------------------------------------------------------------------------------------
inline Mat4 translation(float x, float y, float z) {  // :6-12
  Mat4 t = identity();
  t.m[0][3] = x;
  t.m[1][3] = y;
  t.m[2][3] = z;
  return t;
}
inline Mat4 scaling(float x, float y, float z) {  // :14-20
  Mat4 t = identity();
  t.m[0][0] = x;
  t.m[1][1] = y;
  t.m[2][2] = z;
  return t;
}
inline Mat4 rotation_x(float a) {  // :22-29
  Mat4 t = identity();
  t.m[1][1] = cosf(a);
  t.m[1][2] = -sinf(a);
  t.m[2][1] = sinf(a);
  t.m[2][2] = cosf(a);
  return t;
}
inline Mat4 rotation_y(float a) {  // :31-38
  Mat4 t = identity();
  t.m[0][0] = cosf(a);
  t.m[0][2] = sinf(a);
  t.m[2][0] = -sinf(a);
  t.m[2][2] = cosf(a);
  return t;
}
inline Mat4 rotation_z(float a) {  // :40-47
  Mat4 t = identity();
  t.m[0][0] = cosf(a);
  t.m[0][1] = -sinf(a);
  t.m[1][0] = sinf(a);
  t.m[1][1] = cosf(a);
  return t;
}
inline Mat4 shearing(float xy, float xz, float yx, float yz, float zx, float zy) {  // :49-58
  Mat4 t = identity();
  t.m[0][1] = xy;
  t.m[0][2] = xz;
  t.m[1][0] = yx;
  t.m[1][2] = yz;
  t.m[2][0] = zx;
  t.m[2][1] = zy;
  return t;
}
Mat4 apply_ops(const RtbXfOp* ops, int nops) {  // Transform fluent builder :60-92: op * acc
  Mat4 acc = identity();
  for (int i = 0; i < nops; i++) {
    const RtbXfOp& o = ops[i];
    Mat4 m;
    switch (o.op) {
      case RTB_XF_TRANSLATE: m = translation(o.a[0], o.a[1], o.a[2]); break;
      case RTB_XF_SCALE: m = scaling(o.a[0], o.a[1], o.a[2]); break;
      case RTB_XF_ROTATE_X: m = rotation_x(o.a[0]); break;
      case RTB_XF_ROTATE_Y: m = rotation_y(o.a[0]); break;
      case RTB_XF_ROTATE_Z: m = rotation_z(o.a[0]); break;
      case RTB_XF_SHEAR: m = shearing(o.a[0], o.a[1], o.a[2], o.a[3], o.a[4], o.a[5]); break;
      default: std::memcpy(m.m, o.m, sizeof(m.m)); break;
    }
    acc = matmul(m, acc);
  }
  return acc;
}
Mat4 view_transform(Tuple from, Tuple to, Tuple up) {  // :122-134
  Tuple forward = normalize(sub(to, from));
  Tuple upn = normalize(up);
  Tuple left = cross(forward, upn);
  Tuple true_up = cross(left, forward);
  Mat4 o{};
  o.m[0][0] = left.v[0];
  o.m[0][1] = left.v[1];
  o.m[0][2] = left.v[2];
  o.m[1][0] = true_up.v[0];
  o.m[1][1] = true_up.v[1];
  o.m[1][2] = true_up.v[2];
  o.m[2][0] = -forward.v[0];
  o.m[2][1] = -forward.v[1];
  o.m[2][2] = -forward.v[2];
  o.m[3][3] = 1.f;
  return matmul(o, translation(-from.v[0], -from.v[1], -from.v[2]));
}

// ---------------------------------------------------------------------------------------
// rays.rs
// ---------------------------------------------------------------------------------------
struct Ray {
  Tuple origin, direction;
};
inline Tuple position(const Ray& r, float t) { return add(r.origin, scale(t, r.direction)); }  // :15-17
inline Ray ray_transform(const Ray& r, const Mat4& m) {                                         // :19-24
  return Ray{mat_point(m, r.origin), mat_vector(m, r.direction)};
}

// ---------------------------------------------------------------------------------------
// bounds.rs
// ---------------------------------------------------------------------------------------
struct BBox {
  Tuple min, max;
};
inline BBox bbox_default() { return BBox{point(INF, INF, INF), point(-INF, -INF, -INF)}; }  // :16-23
inline void bbox_add_point(BBox& b, Tuple p) {                                             // :55-74
  if (p.v[0] < b.min.v[0]) b.min.v[0] = p.v[0];
  if (p.v[1] < b.min.v[1]) b.min.v[1] = p.v[1];
  if (p.v[2] < b.min.v[2]) b.min.v[2] = p.v[2];
  if (p.v[0] > b.max.v[0]) b.max.v[0] = p.v[0];
  if (p.v[1] > b.max.v[1]) b.max.v[1] = p.v[1];
  if (p.v[2] > b.max.v[2]) b.max.v[2] = p.v[2];
}
inline void bbox_add_box(BBox& b, const BBox& o) {  // :76-79
  bbox_add_point(b, o.min);
  bbox_add_point(b, o.max);
}
BBox bbox_transform(const BBox& b, const Mat4& t) {  // :36-53
  Tuple ps[8] = {b.min,
                 point(b.min.v[0], b.min.v[1], b.max.v[2]),
                 point(b.min.v[0], b.max.v[1], b.min.v[2]),
                 point(b.min.v[0], b.max.v[1], b.max.v[2]),
                 point(b.max.v[0], b.min.v[1], b.min.v[2]),
                 point(b.max.v[0], b.min.v[1], b.max.v[2]),
                 point(b.max.v[0], b.max.v[1], b.min.v[2]),
                 b.max};
  BBox r = bbox_default();
  for (auto& p : ps) bbox_add_point(r, mat_point(t, p));
  return r;
}
inline bool bbox_contains_point(const BBox& b, Tuple p) {  // :81-85
  return (b.min.v[0] <= p.v[0] && p.v[0] <= b.max.v[0]) && (b.min.v[1] <= p.v[1] && p.v[1] <= b.max.v[1]) &&
         (b.min.v[2] <= p.v[2] && p.v[2] <= b.max.v[2]);
}
inline bool bbox_contains_box(const BBox& b, const BBox& o) {  // :87-89
  return bbox_contains_point(b, o.min) && bbox_contains_point(b, o.max);
}
// Cube::check_axis, shapes/cube.rs:67-85
inline void check_axis(float origin, float direction, float mn, float mx, float* tmin, float* tmax) {
  float tmin_num = mn - origin;
  float tmax_num = mx - origin;
  float a, b;
  if (std::fabs(direction) >= 1e-4f) {
    a = tmin_num / direction;
    b = tmax_num / direction;
  } else {
    a = tmin_num * INF;
    b = tmax_num * INF;
  }
  if (a > b) {  // raw f32 compare (false for NaN)
    *tmin = b;
    *tmax = a;
  } else {
    *tmin = a;
    *tmax = b;
  }
}
bool bbox_intersects(const BBox& b, const Ray& r) {  // bounds.rs:91-123
  float x0, x1, y0, y1, z0, z1;
  check_axis(r.origin.v[0], r.direction.v[0], b.min.v[0], b.max.v[0], &x0, &x1);
  check_axis(r.origin.v[1], r.direction.v[1], b.min.v[1], b.max.v[1], &y0, &y1);
  check_axis(r.origin.v[2], r.direction.v[2], b.min.v[2], b.max.v[2], &z0, &z1);
  float tmin = of_max(of_max(x0, y0), z0);  // Iterator::max over OrderedFloat
  float tmax = of_min(of_min(x1, y1), z1);  // Iterator::min
  return of_cmp(tmin, tmax) <= 0;
}
void bbox_split(const BBox& b, BBox* left, BBox* right) {  // bounds.rs:125-153
  float dx = std::fabs(b.max.v[0] - b.min.v[0]);
  float dy = std::fabs(b.max.v[1] - b.min.v[1]);
  float dz = std::fabs(b.max.v[2] - b.min.v[2]);
  float greatest = of_max(of_max(dx, dy), dz);
  float x0 = b.min.v[0], y0 = b.min.v[1], z0 = b.min.v[2];
  float x1 = b.max.v[0], y1 = b.max.v[1], z1 = b.max.v[2];
  if (of_eq(greatest, dx)) {
    x1 = x0 + dx / 2.f;
    x0 += dx / 2.f;
  } else if (of_eq(greatest, dy)) {
    y1 = y0 + dy / 2.f;
    y0 += dy / 2.f;
  } else {
    z1 = z0 + dz / 2.f;
    z0 += dz / 2.f;
  }
  *left = BBox{b.min, point(x1, y1, z1)};
  *right = BBox{point(x0, y0, z0), b.max};
}

// ---------------------------------------------------------------------------------------
// patterns (patterns.rs, patterns/*.rs) and materials.rs data
// ---------------------------------------------------------------------------------------
struct UvPattern {  // patterns/uv_pattern.rs:8-12
  int kind = 0;     // 0 checker, 1 align-check, 2 image
  float width = 0, height = 0;
  Tuple c[5] = {};  // checker: a,b ; align: main,ul,ur,bl,br
  int image = -1;
};
struct Image {
  uint32_t w, h;
  std::vector<float> rgb;
};
struct Pattern {
  int kind;
  Tuple a, b;
  Mat4 transform, inv;
  int uv_mapping = 0;
  int uv[6] = {-1, -1, -1, -1, -1, -1};  // texture map: uv[0]; cube map: left,front,right,back,up,down
};
struct Material {  // materials.rs:6-33
  int pattern = -1;
  Tuple color = {{1.f, 1.f, 1.f, 0.f}};
  float ambient = 0.1f, diffuse = 0.9f, specular = 0.9f, shininess = 200.f, reflective = 0.f,
        transparency = 0.f, refractive_index = 1.f;
};

// ---------------------------------------------------------------------------------------
// shapes (shapes.rs, shapes/*.rs)
// ---------------------------------------------------------------------------------------
struct Shape {
  int kind = RTB_SPHERE;
  Mat4 transform = identity(), inv = identity();
  Material material;
  bool shadow = true;
  float mn = -INF, mx = INF;  // cylinder/cone
  bool closed = false;
  Tuple p1{}, p2{}, p3{}, e1{}, e2{}, normal{}, n1{}, n2{}, n3{};
  int parent = -1;  // Group handle; ignored by equality (derivative PartialEq="ignore")
  int rank = -1;    // insertion rank, assigned by World::freeze (object-ID for the parity masks)
};

struct Intersection {  // intersections.rs:11-17
  float t;
  const Shape* shape;
  float u, v;
};

struct Light {  // lights.rs:7-11, :36-40, :65-74
  int kind;     // 0 point, 1 area
  Tuple intensity, position;
  Tuple corner{}, uvec{}, vvec{};
  int usteps = 1, vsteps = 1;
};

struct GroupData {  // groups.rs:20-26
  Mat4 transform = identity(), inv = identity();
  std::vector<Shape> shapes;
  std::vector<int> children;
  int parent = -1;
  bool has_bounds = false;
  BBox bounds{};
};

struct Csg;
struct CsgChild {  // csg.rs:127-130
  int kind = RTB_CHILD_SHAPE;
  Shape shape;
  std::shared_ptr<Csg> csg;
};
struct Csg {  // csg.rs:7-11
  int op;
  CsgChild left, right;
};

// Per-thread render context: jitter source + ray counters + path id for the counter RNG.
struct Ctx {
  int jitter_mode = RTB_JITTER_CONST;
  float jitter_value = 0.5f;
  uint64_t seed = 0;
  uint32_t pixel = 0;
  uint32_t path = 1;
  const std::vector<float>* seq = nullptr;
  size_t seq_pos = 0;
  uint64_t n_primary = 0, n_shadow = 0, n_reflect = 0, n_refract = 0;
};
thread_local Ctx g_ctx;

// Counter-based jitter shared bit-for-bit with the HIP kernels (DESIGN.md "Area lights").
inline uint32_t mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
inline float counter_jitter(uint64_t seed, uint32_t pixel, uint32_t path, uint32_t light, uint32_t cell,
                            uint32_t site, uint32_t which) {
  uint32_t h = mix32(static_cast<uint32_t>(seed) ^ 0x9e3779b9u);
  h = mix32(h ^ static_cast<uint32_t>(seed >> 32));
  h = mix32(h ^ pixel);
  h = mix32(h ^ path);
  h = mix32(h ^ light);
  h = mix32(h ^ cell);
  h = mix32(h ^ (site * 2u + which));
  return static_cast<float>(h >> 8) * (1.0f / 16777216.0f);  // rand 0.8 Standard f32: 24 bits
}

}  // namespace

struct RtbWorld {
  std::vector<Light> lights;
  std::vector<Shape> shapes;  // world.rs:14
  std::vector<int> groups;    // world.rs:15
  std::vector<std::shared_ptr<Csg>> csgs;
  int recursion_limit = 4;  // world.rs:38

  // tables standing in for the reference's global slotmaps (groups.rs:16-18, uv_pattern.rs:109-111)
  std::vector<GroupData> group_table;
  std::vector<Image> images;
  std::vector<UvPattern> uvpatterns;
  std::vector<Pattern> patterns;
  std::vector<Shape> pending_shapes;  // detached shapes returned by shape_new
  std::vector<std::shared_ptr<Csg>> pending_csgs;
  std::vector<float> jitter_seq;

  bool frozen = false;
  std::vector<BBox> frozen_bounds;
  std::string error;

  // ---- equality by value (derived PartialEq) ------------------------------------------
  bool uv_eq(int a, int b) const {
    if (a == b && a < 0) return true;
    if (a < 0 || b < 0) return false;
    const UvPattern &x = uvpatterns[a], &y = uvpatterns[b];
    if (x.kind != y.kind) return false;
    if (x.kind == 0) return x.width == y.width && x.height == y.height && teq(x.c[0], y.c[0]) && teq(x.c[1], y.c[1]);
    if (x.kind == 1) {
      for (int i = 0; i < 5; i++)
        if (!teq(x.c[i], y.c[i])) return false;
      return true;
    }
    return x.image == y.image;  // slotmap key equality
  }
  bool pattern_eq(int a, int b) const {
    const Pattern &x = patterns[a], &y = patterns[b];
    if (x.kind != y.kind) return false;
    if (!mat_eq(x.transform, y.transform)) return false;
    switch (x.kind) {
      case RTB_PAT_XYZ_RGB: return true;
      case RTB_PAT_TEXTURE_MAP: return x.uv_mapping == y.uv_mapping && uv_eq(x.uv[0], y.uv[0]);
      case RTB_PAT_CUBE_MAP:
        for (int i = 0; i < 6; i++)
          if (!uv_eq(x.uv[i], y.uv[i])) return false;
        return true;
      default: return teq(x.a, y.a) && teq(x.b, y.b);
    }
  }
  bool material_eq(const Material& a, const Material& b) const {
    if ((a.pattern < 0) != (b.pattern < 0)) return false;
    if (a.pattern >= 0 && !pattern_eq(a.pattern, b.pattern)) return false;
    return teq(a.color, b.color) && a.ambient == b.ambient && a.diffuse == b.diffuse && a.specular == b.specular &&
           a.shininess == b.shininess && a.reflective == b.reflective && a.transparency == b.transparency &&
           a.refractive_index == b.refractive_index;
  }
  // Shape == Shape: all fields except `parent` (shapes/sphere.rs:14-22 and siblings).
  bool shape_eq(const Shape& a, const Shape& b) const {
    if (a.kind != b.kind) return false;
    if (!mat_eq(a.transform, b.transform)) return false;
    if (!material_eq(a.material, b.material)) return false;
    if (a.shadow != b.shadow) return false;
    switch (a.kind) {
      case RTB_CYLINDER:
      case RTB_CONE: return a.mn == b.mn && a.mx == b.mx && a.closed == b.closed;
      case RTB_TRIANGLE:
        return teq(a.p1, b.p1) && teq(a.p2, b.p2) && teq(a.p3, b.p3) && teq(a.e1, b.e1) && teq(a.e2, b.e2) &&
               teq(a.normal, b.normal);
      case RTB_SMOOTH_TRIANGLE:
        return teq(a.p1, b.p1) && teq(a.p2, b.p2) && teq(a.p3, b.p3) && teq(a.e1, b.e1) && teq(a.e2, b.e2) &&
               teq(a.n1, b.n1) && teq(a.n2, b.n2) && teq(a.n3, b.n3);
      default: return true;
    }
  }
  bool isect_eq(const Intersection& a, const Intersection& b) const {  // intersections.rs:11 derive(PartialEq)
    return of_eq(a.t, b.t) && shape_eq(*a.shape, *b.shape) && a.u == b.u && a.v == b.v;
  }

  // ---- shape bounds (shapes/*.rs bounds(), shapes.rs:306-320) ---------------------------
  static BBox shape_bounds(const Shape& s) {
    switch (s.kind) {
      case RTB_SPHERE:
      case RTB_CUBE: return BBox{point(-1.f, -1.f, -1.f), point(1.f, 1.f, 1.f)};  // sphere.rs:86-90, cube.rs:135-139
      case RTB_PLANE: return BBox{point(-INF, 0.f, -INF), point(INF, 0.f, INF)};   // plane.rs:79-83
      case RTB_CYLINDER: return BBox{point(-1.f, s.mn, -1.f), point(1.f, s.mx, 1.f)};  // cylinder.rs:169-173
      case RTB_CONE: {                                                                 // cone.rs:182-193
        float a = std::fabs(s.mn), b = std::fabs(s.mx);
        float limit = (std::isinf(a) || std::isinf(b)) ? INF : of_max(a, b);
        return BBox{point(-limit, s.mn, -limit), point(limit, s.mx, limit)};
      }
      default: {  // triangle.rs:117-123, smooth_triangle.rs:120-126
        BBox bb = bbox_default();
        bbox_add_point(bb, s.p1);
        bbox_add_point(bb, s.p2);
        bbox_add_point(bb, s.p3);
        return bb;
      }
    }
  }
  static BBox parent_space_bounds(const Shape& s) { return bbox_transform(shape_bounds(s), s.transform); }

  // ---- groups.rs -----------------------------------------------------------------------
  BBox make_bounds(int g) const {  // groups.rs:176-188 (incl. the null-parent re-transform quirk)
    const GroupData& gd = group_table[g];
    BBox bb = bbox_default();
    for (const Shape& s : gd.shapes) bbox_add_box(bb, parent_space_bounds(s));
    for (int child : gd.children) {
      if (gd.parent < 0) bb = bbox_transform(bb, gd.transform);
      bbox_add_box(bb, group_bounds(child));
    }
    return bb;
  }
  BBox group_bounds(int g) const {  // groups.rs:166-174
    if (frozen) return frozen_bounds[g];
    const GroupData& gd = group_table[g];
    if (gd.has_bounds) return gd.bounds;
    return make_bounds(g);
  }
  int group_len(int g) const { return (int)(group_table[g].shapes.size() + group_table[g].children.size()); }
  void group_add_child(int g, int child) {  // groups.rs:64-72
    group_table[g].children.push_back(child);
    group_table[child].parent = g;
  }
  void group_add_shape(int g, Shape s) {  // groups.rs:80-85
    s.parent = g;
    group_table[g].shapes.push_back(s);
  }
  int group_insert(GroupData&& gd) {
    group_table.push_back(std::move(gd));
    return (int)group_table.size() - 1;
  }
  void partition(int g, int* left_out, int* right_out) {  // groups.rs:198-282
    BBox left_bb, right_bb;
    bbox_split(group_bounds(g), &left_bb, &right_bb);
    std::vector<Shape> remaining;
    remaining.swap(group_table[g].shapes);
    std::vector<Shape> left_shapes, rest, right_shapes, rest2;
    for (auto& s : remaining) (bbox_contains_box(left_bb, parent_space_bounds(s)) ? left_shapes : rest).push_back(s);
    for (auto& s : rest) (bbox_contains_box(right_bb, parent_space_bounds(s)) ? right_shapes : rest2).push_back(s);
    group_table[g].shapes = std::move(rest2);

    std::vector<int> rem_children;
    rem_children.swap(group_table[g].children);
    std::vector<int> left_children, crest, right_children, crest2;
    for (int c : rem_children) (bbox_contains_box(left_bb, group_bounds(c)) ? left_children : crest).push_back(c);
    for (int c : crest) (bbox_contains_box(right_bb, group_bounds(c)) ? right_children : crest2).push_back(c);
    group_table[g].children = std::move(crest2);

    GroupData l;
    l.shapes = std::move(left_shapes);
    l.children = left_children;
    int lg = group_insert(std::move(l));
    for (auto& s : group_table[lg].shapes) s.parent = lg;
    for (int c : left_children) group_table[c].parent = lg;

    GroupData r;
    r.shapes = std::move(right_shapes);
    r.children = right_children;
    int rg = group_insert(std::move(r));
    for (auto& s : group_table[rg].shapes) s.parent = rg;
    for (int c : right_children) group_table[c].parent = rg;
    *left_out = lg;
    *right_out = rg;
  }
  void divide(int g, int threshold) {  // groups.rs:284-299
    if (threshold <= group_len(g)) {
      int l, r;
      partition(g, &l, &r);
      if (group_len(l) != 0) group_add_child(g, l);
      if (group_len(r) != 0) group_add_child(g, r);
    }
    std::vector<int> children = group_table[g].children;
    for (int c : children) divide(c, threshold);
  }

  // ---- freeze: memoise bounds() and assign insertion ranks ------------------------------
  void rank_csg(Csg& c, int* next) {
    // tie order inside a CSG node (csg.rs:32-46): shape child before csg child, else left then right
    auto visit = [&](CsgChild& ch) {
      if (ch.kind == RTB_CHILD_SHAPE)
        ch.shape.rank = (*next)++;
      else
        rank_csg(*ch.csg, next);
    };
    if (c.left.kind == RTB_CHILD_CSG && c.right.kind == RTB_CHILD_SHAPE) {
      visit(c.right);
      visit(c.left);
    } else {
      visit(c.left);
      visit(c.right);
    }
  }
  void rank_group(int g, int* next) {
    for (auto& s : group_table[g].shapes) s.rank = (*next)++;
    for (int c : group_table[g].children) rank_group(c, next);
  }
  void freeze_bounds_rec(int g) {
    for (int c : group_table[g].children) freeze_bounds_rec(c);
    const GroupData& gd = group_table[g];
    // children are already memoised; group_bounds(child) reads them through `frozen_bounds`
    BBox bb;
    if (gd.has_bounds) {
      bb = gd.bounds;
    } else {
      bb = bbox_default();
      for (const Shape& s : gd.shapes) bbox_add_box(bb, parent_space_bounds(s));
      for (int child : gd.children) {
        if (gd.parent < 0) bb = bbox_transform(bb, gd.transform);
        bbox_add_box(bb, frozen_bounds[child]);
      }
    }
    frozen_bounds[g] = bb;
  }
  void freeze() {
    if (frozen) return;
    int next = 0;
    for (auto& c : csgs) rank_csg(*c, &next);
    for (int g : groups) rank_group(g, &next);
    for (auto& s : shapes) s.rank = next++;
    frozen_bounds.assign(group_table.size(), bbox_default());
    for (int g : groups) freeze_bounds_rec(g);
    frozen = true;
  }
  void thaw() { frozen = false; }

  // ---- local intersections --------------------------------------------------------------
  static void sphere_local(const Shape& s, const Ray& r, std::vector<Intersection>& xs) {  // sphere.rs:64-80
    Tuple sphere_to_ray = sub(r.origin, point(0.f, 0.f, 0.f));
    float a = dot(r.direction, r.direction);
    float b = 2.f * dot(r.direction, sphere_to_ray);
    float c = dot(sphere_to_ray, sphere_to_ray) - 1.f;
    float disc = b * b - 4.f * a * c;
    if (disc < 0.f) return;
    float t1 = (-b - std::sqrt(disc)) / (2.f * a);
    float t2 = (-b + std::sqrt(disc)) / (2.f * a);
    xs.push_back({t1, &s, F32_MAX, F32_MAX});
    xs.push_back({t2, &s, F32_MAX, F32_MAX});
  }
  static void plane_local(const Shape& s, const Ray& r, std::vector<Intersection>& xs) {  // plane.rs:66-73
    if (std::fabs(r.direction.v[1]) < 1e-4f) return;
    float t = -r.origin.v[1] / r.direction.v[1];
    xs.push_back({t, &s, F32_MAX, F32_MAX});
  }
  static void cube_local(const Shape& s, const Ray& r, std::vector<Intersection>& xs) {  // cube.rs:87-114
    float x0, x1, y0, y1, z0, z1;
    check_axis(r.origin.v[0], r.direction.v[0], -1.f, 1.f, &x0, &x1);
    check_axis(r.origin.v[1], r.direction.v[1], -1.f, 1.f, &y0, &y1);
    check_axis(r.origin.v[2], r.direction.v[2], -1.f, 1.f, &z0, &z1);
    float tmin = of_max(of_max(x0, y0), z0);
    float tmax = of_min(of_min(x1, y1), z1);
    if (of_cmp(tmin, tmax) > 0) return;
    xs.push_back({tmin, &s, F32_MAX, F32_MAX});
    xs.push_back({tmax, &s, F32_MAX, F32_MAX});
  }
  // cylinder.rs:89-93 / cone.rs:90-94
  static bool check_cap_cyl(const Ray& r, float t) {
    float x = r.origin.v[0] + t * r.direction.v[0];
    float z = r.origin.v[2] + t * r.direction.v[2];
    return (x * x + z * z) <= 1.f;
  }
  static bool check_cap_cone(const Ray& r, float t, float y) {
    float x = r.origin.v[0] + t * r.direction.v[0];
    float z = r.origin.v[2] + t * r.direction.v[2];
    return (x * x + z * z) <= std::fabs(y);
  }
  template <int N>
  static void fill_first_free(float (&ts)[N], bool (&has)[N], float t) {  // xs.iter_mut().find(|x| x.is_none())
    for (int i = 0; i < N; i++)
      if (!has[i]) {
        has[i] = true;
        ts[i] = t;
        return;
      }
  }
  static void cylinder_local(const Shape& s, const Ray& r, std::vector<Intersection>& xs) {  // cylinder.rs:95-156
    float ts[2];
    bool has[2] = {false, false};
    auto caps = [&]() {  // intersect_caps :95-119
      if (!s.closed || std::fabs(r.direction.v[1]) < 1e-4f) return;
      float t = (s.mn - r.origin.v[1]) / r.direction.v[1];
      if (check_cap_cyl(r, t)) fill_first_free(ts, has, t);
      t = (s.mx - r.origin.v[1]) / r.direction.v[1];
      if (check_cap_cyl(r, t)) fill_first_free(ts, has, t);
    };
    float a = r.direction.v[0] * r.direction.v[0] + r.direction.v[2] * r.direction.v[2];
    if (std::fabs(a) < 1e-4f) {
      caps();
    } else {
      float b = 2.f * r.origin.v[0] * r.direction.v[0] + 2.f * r.origin.v[2] * r.direction.v[2];
      float c = r.origin.v[0] * r.origin.v[0] + r.origin.v[2] * r.origin.v[2] - 1.f;
      float disc = b * b - 4.f * a * c;
      if (disc < 0.f) return;
      float t0 = (-b - std::sqrt(disc)) / (2.f * a);
      float t1 = (-b + std::sqrt(disc)) / (2.f * a);
      if (t0 > t1) std::swap(t0, t1);
      float y0 = r.origin.v[1] + t0 * r.direction.v[1];
      if (s.mn < y0 && y0 < s.mx) {
        has[0] = true;
        ts[0] = t0;
      }
      float y1 = r.origin.v[1] + t1 * r.direction.v[1];
      if (s.mn < y1 && y1 < s.mx) {
        has[1] = true;
        ts[1] = t1;
      }
      caps();
    }
    for (int i = 0; i < 2; i++)
      if (has[i]) xs.push_back({ts[i], &s, F32_MAX, F32_MAX});
  }
  static void cone_local(const Shape& s, const Ray& r, std::vector<Intersection>& xs) {  // cone.rs:96-165
    float ts[4];
    bool has[4] = {false, false, false, false};
    auto caps = [&]() {  // :96-120
      if (!s.closed || std::fabs(r.direction.v[1]) < 1e-4f) return;
      float t = (s.mn - r.origin.v[1]) / r.direction.v[1];
      if (check_cap_cone(r, t, s.mn)) fill_first_free(ts, has, t);
      t = (s.mx - r.origin.v[1]) / r.direction.v[1];
      if (check_cap_cone(r, t, s.mx)) fill_first_free(ts, has, t);
    };
    const Tuple &o = r.origin, &d = r.direction;
    float a = d.v[0] * d.v[0] - d.v[1] * d.v[1] + d.v[2] * d.v[2];
    float b = 2.f * o.v[0] * d.v[0] - 2.f * o.v[1] * d.v[1] + 2.f * o.v[2] * d.v[2];
    float c = o.v[0] * o.v[0] - o.v[1] * o.v[1] + o.v[2] * o.v[2];
    if (std::fabs(a) < 1e-4f) {
      if (std::fabs(b) < 1e-4f) return;
      float t = -c / (2.f * b);
      has[0] = true;
      ts[0] = t;
      caps();
    } else {
      float disc = b * b - 4.f * a * c;
      if (disc < 0.f) return;
      float t0 = (-b - std::sqrt(disc)) / (2.f * a);
      float t1 = (-b + std::sqrt(disc)) / (2.f * a);
      if (t0 > t1) std::swap(t0, t1);
      float y0 = o.v[1] + t0 * d.v[1];
      if (s.mn < y0 && y0 < s.mx) {
        has[0] = true;
        ts[0] = t0;
      }
      float y1 = o.v[1] + t1 * d.v[1];
      if (s.mn < y1 && y1 < s.mx) {
        has[1] = true;
        ts[1] = t1;
      }
      caps();
    }
    for (int i = 0; i < 4; i++)
      if (has[i]) xs.push_back({ts[i], &s, F32_MAX, F32_MAX});
  }
  static void triangle_local(const Shape& s, const Ray& r, std::vector<Intersection>& xs, bool smooth) {
    // triangle.rs:93-115, smooth_triangle.rs:96-118
    Tuple dir_cross_e2 = cross(r.direction, s.e2);
    float det = dot(s.e1, dir_cross_e2);
    if (std::fabs(det) < 1e-4f) return;
    float f = 1.f / det;
    Tuple p1_to_origin = sub(r.origin, s.p1);
    float u = f * dot(p1_to_origin, dir_cross_e2);
    if (!(0.f <= u && u <= 1.f)) return;
    Tuple origin_cross_e1 = cross(p1_to_origin, s.e1);
    float v = f * dot(r.direction, origin_cross_e1);
    if (v < 0.f || (u + v) > 1.f) return;
    float t = f * dot(s.e2, origin_cross_e1);
    if (smooth)
      xs.push_back({t, &s, u, v});
    else
      xs.push_back({t, &s, F32_MAX, F32_MAX});
  }
  static void shape_local_intersect(const Shape& s, const Ray& local, std::vector<Intersection>& xs) {
    switch (s.kind) {
      case RTB_SPHERE: sphere_local(s, local, xs); break;
      case RTB_PLANE: plane_local(s, local, xs); break;
      case RTB_CUBE: cube_local(s, local, xs); break;
      case RTB_CYLINDER: cylinder_local(s, local, xs); break;
      case RTB_CONE: cone_local(s, local, xs); break;
      case RTB_TRIANGLE: triangle_local(s, local, xs, false); break;
      default: triangle_local(s, local, xs, true); break;
    }
  }
  static void shape_intersect(const Shape& s, std::vector<Intersection>& xs, const Ray& world_ray) {  // shapes.rs:204-246
    Ray local = ray_transform(world_ray, s.inv);
    shape_local_intersect(s, local, xs);
  }

  // ---- Intersections (intersections.rs:66-101) -----------------------------------------
  // Vec -> BTreeMap::from_iter: stable sort by key order (t only), then the std bulk builder drops
  // an entry when it is == (full PartialEq) to its successor.
  void make_intersections(std::vector<Intersection>& xs) const {
    std::stable_sort(xs.begin(), xs.end(),
                     [](const Intersection& a, const Intersection& b) { return of_cmp(a.t, b.t) < 0; });
    if (xs.size() < 2) return;
    std::vector<Intersection> out;
    out.reserve(xs.size());
    for (size_t i = 0; i < xs.size(); i++) {
      if (i + 1 < xs.size() && isect_eq(xs[i], xs[i + 1])) continue;  // DedupSortedIter keeps the later one
      out.push_back(xs[i]);
    }
    xs.swap(out);
  }
  static const Intersection* hit(const std::vector<Intersection>& xs) {  // intersections.rs:94-96
    for (auto& x : xs)
      if (x.t > 0.f) return &x;
    return nullptr;
  }

  // ---- groups.rs:110-133 ----------------------------------------------------------------
  void group_local_intersect(int g, const Ray& local, std::vector<Intersection>& xs) const {
    if (!bbox_intersects(group_bounds(g), local)) return;
    const GroupData& gd = group_table[g];
    for (const Shape& s : gd.shapes) shape_intersect(s, xs, local);
    for (int c : gd.children) group_intersect(c, xs, local);
  }
  void group_intersect(int g, std::vector<Intersection>& xs, const Ray& world_ray) const {
    Ray local = ray_transform(world_ray, group_table[g].inv);
    group_local_intersect(g, local, xs);
  }

  // ---- csg.rs ---------------------------------------------------------------------------
  BBox csg_bounds(const Csg& c) const {  // csg.rs:87-105
    BBox bb = bbox_default();
    auto add_child = [&](const CsgChild& ch) {
      if (ch.kind == RTB_CHILD_SHAPE)
        bbox_add_box(bb, parent_space_bounds(ch.shape));
      else
        bbox_add_box(bb, csg_bounds(*ch.csg));
    };
    if (c.left.kind == RTB_CHILD_CSG && c.right.kind == RTB_CHILD_SHAPE) {
      add_child(c.right);
      add_child(c.left);
    } else {
      add_child(c.left);
      add_child(c.right);
    }
    return bb;
  }
  bool csg_includes(const Csg& c, const Shape& s) const {  // csg.rs:74-85
    auto inc = [&](const CsgChild& ch) {
      return ch.kind == RTB_CHILD_SHAPE ? shape_eq(ch.shape, s) : csg_includes(*ch.csg, s);
    };
    if (c.left.kind == RTB_CHILD_CSG && c.right.kind == RTB_CHILD_SHAPE) return inc(c.right) || inc(c.left);
    return inc(c.left) || inc(c.right);
  }
  static bool intersection_allowed(int op, bool l_hit, bool in_l, bool in_r) {  // csg.rs:116-124
    switch (op) {
      case RTB_CSG_UNION: return (l_hit && !in_r) || (!l_hit && !in_l);
      case RTB_CSG_INTERSECT: return (l_hit && in_r) || (!l_hit && in_l);
      default: return (l_hit && !in_r) || (!l_hit && in_l);
    }
  }
  std::vector<Intersection> csg_filter(const Csg& c, const std::vector<Intersection>& xs) const {  // csg.rs:51-72
    bool in_l = false, in_r = false;
    std::vector<Intersection> result;
    for (const Intersection& x : xs) {
      bool l_hit = c.left.kind == RTB_CHILD_SHAPE ? shape_eq(c.left.shape, *x.shape) : csg_includes(*c.left.csg, *x.shape);
      if (intersection_allowed(c.op, l_hit, in_l, in_r)) result.push_back(x);
      if (l_hit)
        in_l = !in_l;
      else
        in_r = !in_r;
    }
    make_intersections(result);  // `result.into()` (csg.rs:71)
    return result;
  }
  std::vector<Intersection> csg_local_intersect(const Csg& c, const Ray& ray) const {  // csg.rs:26-49
    if (!bbox_intersects(csg_bounds(c), ray)) return {};
    std::vector<Intersection> xs;
    auto child = [&](const CsgChild& ch) {
      if (ch.kind == RTB_CHILD_SHAPE) {
        shape_intersect(ch.shape, xs, ray);
      } else {
        auto sub_xs = csg_local_intersect(*ch.csg, ray);
        xs.insert(xs.end(), sub_xs.begin(), sub_xs.end());
      }
    };
    if (c.left.kind == RTB_CHILD_CSG && c.right.kind == RTB_CHILD_SHAPE) {
      child(c.right);
      child(c.left);
    } else {
      child(c.left);
      child(c.right);
    }
    make_intersections(xs);
    return csg_filter(c, xs);
  }

  // ---- world.rs:44-62 -------------------------------------------------------------------
  std::vector<Intersection> intersect(const Ray& ray) const {
    std::vector<Intersection> xs;
    for (auto& c : csgs) {
      auto sub_xs = csg_local_intersect(*c, ray);
      xs.insert(xs.end(), sub_xs.begin(), sub_xs.end());
    }
    for (int g : groups) group_intersect(g, xs, ray);
    for (auto& s : shapes) shape_intersect(s, xs, ray);
    make_intersections(xs);
    return xs;
  }

  // ---- shapes.rs:187-202, :272-292 ; groups.rs:139-164 ---------------------------------
  Tuple group_world_to_object(int g, Tuple p) const {
    const GroupData& gd = group_table[g];
    if (gd.parent >= 0) p = group_world_to_object(gd.parent, p);
    return mat_point(gd.inv, p);
  }
  Tuple group_normal_to_world(int g, Tuple n) const {
    const GroupData& gd = group_table[g];
    Tuple t = mat_vector(transpose(gd.inv), n);
    Tuple nn = normalize(vector(t.v[0], t.v[1], t.v[2]));
    if (gd.parent >= 0) return group_normal_to_world(gd.parent, nn);
    return nn;
  }
  Tuple world_to_object(const Shape& s, Tuple p) const {
    if (s.parent >= 0) p = group_world_to_object(s.parent, p);
    return mat_point(s.inv, p);
  }
  Tuple normal_to_world(const Shape& s, Tuple n) const {
    Tuple t = mat_vector(transpose(s.inv), n);
    Tuple nn = normalize(vector(t.v[0], t.v[1], t.v[2]));
    if (s.parent >= 0) return group_normal_to_world(s.parent, nn);
    return nn;
  }
  static Tuple local_normal_at(const Shape& s, Tuple lp, float hu, float hv) {
    switch (s.kind) {
      case RTB_SPHERE: return sub(lp, point(0.f, 0.f, 0.f));  // sphere.rs:82-84
      case RTB_PLANE: return vector(0.f, 1.f, 0.f);           // plane.rs:75-77
      case RTB_CUBE: {                                        // cube.rs:116-133
        float ax = std::fabs(lp.v[0]), ay = std::fabs(lp.v[1]), az = std::fabs(lp.v[2]);
        float maxc = of_max(of_max(ax, ay), az);
        if (ax == maxc) return vector(lp.v[0], 0.f, 0.f);
        if (ay == maxc) return vector(0.f, lp.v[1], 0.f);
        return vector(0.f, 0.f, lp.v[2]);  // (a NaN point would hit unreachable!() in the reference)
      }
      case RTB_CYLINDER: {  // cylinder.rs:158-167
        float dist = lp.v[0] * lp.v[0] + lp.v[2] * lp.v[2];
        if (dist < 1.f && lp.v[1] >= s.mx - 1e-4f) return vector(0.f, 1.f, 0.f);
        if (dist < 1.f && lp.v[1] <= s.mn + 1e-4f) return vector(0.f, -1.f, 0.f);
        return vector(lp.v[0], 0.f, lp.v[2]);
      }
      case RTB_CONE: {  // cone.rs:167-180
        float dist = lp.v[0] * lp.v[0] + lp.v[2] * lp.v[2];
        if (dist < 1.f && lp.v[1] >= s.mx - 1e-4f) return vector(0.f, 1.f, 0.f);
        if (dist < 1.f && lp.v[1] <= s.mn + 1e-4f) return vector(0.f, -1.f, 0.f);
        float y = std::sqrt(lp.v[0] * lp.v[0] + lp.v[2] * lp.v[2]);
        if (lp.v[1] > 0.f) y = -y;
        return vector(lp.v[0], y, lp.v[2]);
      }
      case RTB_TRIANGLE: return s.normal;  // triangle.rs:89-91
      default:                             // smooth_triangle.rs:92-94
        return add(add(scale(hu, s.n2), scale(hv, s.n3)), scale(1.f - hu - hv, s.n1));
    }
  }
  Tuple normal_at(const Shape& s, Tuple world_point, float hu, float hv) const {  // shapes.rs:187-202
    Tuple lp = world_to_object(s, world_point);
    Tuple ln = local_normal_at(s, lp, hu, hv);
    return normal_to_world(s, ln);
  }

  // ---- intersections.rs:103-193 ----------------------------------------------------------
  struct Comps {
    float t;
    const Shape* shape;
    Tuple point, over_point, under_point, eyev, normalv, reflectv;
    bool inside;
    float n1, n2;
  };
  Comps prepare(const Intersection& ix, const Ray& ray, const std::vector<Intersection>& xs) const {
    Comps c;
    Tuple world_point = position(ray, ix.t);
    c.t = ix.t;
    c.shape = ix.shape;
    c.point = world_point;
    c.eyev = neg(ray.direction);
    c.normalv = normal_at(*ix.shape, world_point, ix.u, ix.v);
    if (dot(c.normalv, c.eyev) < 0.f) {
      c.normalv = neg(c.normalv);
      c.inside = true;
    } else {
      c.inside = false;
    }
    c.reflectv = reflect(ray.direction, c.normalv);
    c.over_point = add(c.point, scale(1e-4f, c.normalv));
    c.under_point = sub(c.point, scale(1e-4f, c.normalv));
    c.n1 = 1.f;
    c.n2 = 1.f;
    std::vector<const Shape*> containers;
    for (const Intersection& x : xs) {
      bool is_hit = isect_eq(x, ix);
      if (is_hit && !containers.empty()) c.n1 = containers.back()->material.refractive_index;
      size_t pos = containers.size();
      for (size_t i = 0; i < containers.size(); i++)
        if (shape_eq(*containers[i], *x.shape)) {
          pos = i;
          break;
        }
      if (pos < containers.size())
        containers.erase(containers.begin() + pos);
      else
        containers.push_back(x.shape);
      if (is_hit && !containers.empty()) c.n2 = containers.back()->material.refractive_index;
    }
    return c;
  }
  static float powi5(float x) {  // llvm powi expansion for a constant exponent: x * ((x*x)*(x*x))
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
  }
  static float schlick(const Comps& c) {  // intersections.rs:177-192
    float cosv = dot(c.eyev, c.normalv);
    if (c.n1 > c.n2) {
      float n = c.n1 / c.n2;
      float sin2_t = (n * n) * (1.f - cosv * cosv);
      if (sin2_t > 1.f) return 1.f;
      cosv = std::sqrt(1.f - sin2_t);
    }
    float r = (c.n1 - c.n2) / (c.n1 + c.n2);
    float r0 = r * r;
    return r0 + (1.f - r0) * powi5(1.f - cosv);
  }

  // ---- patterns --------------------------------------------------------------------------
  static float rem_euclid1(float x) {  // f32::rem_euclid(1.)
    float r = std::fmod(x, 1.f);
    return r < 0.f ? r + 1.f : r;
  }
  static void uv_map(int mapping, Tuple p, float* u, float* v) {  // texture_map.rs:59-114
    if (mapping == RTB_UVMAP_SPHERICAL) {
      float theta = atan2f(p.v[0], p.v[2]);
      float radius = magnitude(vector(p.v[0], p.v[1], p.v[2]));
      float phi = acosf(p.v[1] / radius);
      float raw_u = theta / (2.f * PI_F);
      *u = 1.f - (raw_u + 0.5f);
      *v = 1.f - phi / PI_F;
    } else if (mapping == RTB_UVMAP_PLANAR) {
      *u = rem_euclid1(p.v[0]);
      *v = rem_euclid1(p.v[2]);
    } else {
      float theta = atan2f(p.v[0], p.v[2]);
      float raw_u = theta / (2.f * PI_F);
      *u = 1.f - (raw_u + 0.5f);
      *v = rem_euclid1(p.v[1]);
    }
  }
  static uint32_t round_to_u32(float x) {  // `x.round() as u32` (saturating cast)
    float r = std::round(x);
    if (!(r > 0.f)) return 0;
    if (r >= 4294967296.f) return 0xFFFFFFFFu;
    return (uint32_t)r;
  }
  Tuple uv_pattern_at(int id, float u, float v) const {  // uv_pattern.rs:14-22
    const UvPattern& p = uvpatterns[id];
    if (p.kind == 0) {  // :44-52
      float u2 = std::floor(u * p.width);
      float v2 = std::floor(v * p.height);
      return std::fmod(u2 + v2, 2.f) == 0.f ? p.c[0] : p.c[1];
    }
    if (p.kind == 1) {  // :89-104
      if (v > 0.8f) {
        if (u < 0.2f) return p.c[1];
        if (u > 0.8f) return p.c[2];
      } else if (v < 0.2f) {
        if (u < 0.2f) return p.c[3];
        if (u > 0.8f) return p.c[4];
      }
      return p.c[0];
    }
    // image :122-132
    const Image& im = images[p.image];
    float vv = 1.f - v;
    float x = u * ((float)im.w - 1.f);
    float y = vv * ((float)im.h - 1.f);
    uint32_t xi = round_to_u32(x), yi = round_to_u32(y);
    if (xi >= im.w) xi = im.w - 1;  // the reference would panic out of bounds; clamp instead
    if (yi >= im.h) yi = im.h - 1;
    const float* px = &im.rgb[(size_t(yi) * im.w + xi) * 3];
    return color(px[0], px[1], px[2]);
  }
  static int cube_face(Tuple p) {  // cube_map.rs:81-101 ; 0 left,1 front,2 right,3 back,4 up,5 down
    float coord = of_max(of_max(std::fabs(p.v[0]), std::fabs(p.v[1])), std::fabs(p.v[2]));
    if (coord == p.v[0]) return 2;
    if (coord == -p.v[0]) return 0;
    if (coord == p.v[1]) return 4;
    if (coord == -p.v[1]) return 5;
    if (coord == p.v[2]) return 1;
    return 3;
  }
  static void cube_uv(int face, Tuple p, float* u, float* v) {  // cube_map.rs:103-138
    switch (face) {
      case 1:
        *u = std::fmod(p.v[0] + 1.f, 2.f) / 2.f;
        *v = std::fmod(p.v[1] + 1.f, 2.f) / 2.f;
        break;
      case 3:
        *u = std::fmod(1.f - p.v[0], 2.f) / 2.f;
        *v = std::fmod(p.v[1] + 1.f, 2.f) / 2.f;
        break;
      case 4:
        *u = std::fmod(p.v[0] + 1.f, 2.f) / 2.f;
        *v = std::fmod(1.f - p.v[2], 2.f) / 2.f;
        break;
      case 5:
        *u = std::fmod(p.v[0] + 1.f, 2.f) / 2.f;
        *v = std::fmod(p.v[2] + 1.f, 2.f) / 2.f;
        break;
      case 0:
        *u = std::fmod(p.v[2] + 1.f, 2.f) / 2.f;
        *v = std::fmod(p.v[1] + 1.f, 2.f) / 2.f;
        break;
      default:
        *u = std::fmod(1.f - p.v[2], 2.f) / 2.f;
        *v = std::fmod(p.v[1] + 1.f, 2.f) / 2.f;
        break;
    }
  }
  Tuple pattern_at(const Pattern& p, Tuple pp) const {
    switch (p.kind) {
      case RTB_PAT_STRIPED:  // striped.rs:28-34
        return std::fmod(std::floor(pp.v[0]), 2.f) == 0.f ? p.a : p.b;
      case RTB_PAT_GRADIENT: {  // gradient.rs:28-32
        Tuple distance = sub(p.b, p.a);
        float fraction = pp.v[0] - std::floor(pp.v[0]);
        return add(p.a, cscale(distance, fraction));
      }
      case RTB_PAT_RING:  // ring.rs:28-39
        return std::fmod(std::floor(std::sqrt(pp.v[0] * pp.v[0] + pp.v[2] * pp.v[2])), 2.f) == 0.f ? p.a : p.b;
      case RTB_PAT_CHECKER:  // checker.rs:28-36
        return std::fmod(std::floor(pp.v[0]) + std::floor(pp.v[1]) + std::floor(pp.v[2]), 2.f) == 0.f ? p.a : p.b;
      case RTB_PAT_XYZ_RGB: return color(pp.v[0], pp.v[1], pp.v[2]);  // xyz_rgb.rs:24-26
      case RTB_PAT_TEXTURE_MAP: {                                    // texture_map.rs:31-34
        float u, v;
        uv_map(p.uv_mapping, pp, &u, &v);
        return uv_pattern_at(p.uv[0], u, v);
      }
      default: {  // cube_map.rs:46-57
        int face = cube_face(pp);
        float u, v;
        cube_uv(face, pp, &u, &v);
        return uv_pattern_at(p.uv[face], u, v);
      }
    }
  }
  Tuple pattern_at_shape(const Pattern& p, const Shape& s, Tuple world_point) const {  // patterns.rs:79-91
    Tuple op = world_to_object(s, world_point);
    Tuple pp = mat_point(p.inv, op);
    return pattern_at(p, pp);
  }

  // ---- lights.rs -------------------------------------------------------------------------
  Tuple point_on_light(const Light& l, int light_index, int u, int v, int site) const {  // lights.rs:105-121
    float ju, jv;
    Ctx& cx = g_ctx;
    if (cx.jitter_mode == RTB_JITTER_SEQUENCE && cx.seq && !cx.seq->empty()) {
      ju = (*cx.seq)[cx.seq_pos++ % cx.seq->size()];
      jv = (*cx.seq)[cx.seq_pos++ % cx.seq->size()];
    } else if (cx.jitter_mode == RTB_JITTER_COUNTER) {
      uint32_t cell = (uint32_t)(u * l.vsteps + v);
      ju = counter_jitter(cx.seed, cx.pixel, cx.path, (uint32_t)light_index, cell, (uint32_t)site, 0);
      jv = counter_jitter(cx.seed, cx.pixel, cx.path, (uint32_t)light_index, cell, (uint32_t)site, 1);
    } else {
      ju = jv = cx.jitter_value;
    }
    return add(add(l.corner, scale((float)u + ju, l.uvec)), scale((float)v + jv, l.vvec));
  }
  bool is_shadowed(Tuple light_position, Tuple p) const {  // world.rs:101-111
    Tuple v = sub(light_position, p);
    float distance = magnitude(v);
    Tuple direction = normalize(v);
    Ray r{p, direction};
    g_ctx.n_shadow++;
    auto xs = intersect(r);
    if (const Intersection* h = hit(xs)) return h->t < distance && h->shape->shadow;
    return false;
  }
  float intensity_at(const Light& l, int light_index, Tuple p) const {  // lights.rs:28-33, :50-56, :123-134
    if (l.kind == 0) return is_shadowed(l.position, p) ? 0.f : 1.f;
    float total = 0.f;
    for (int u = 0; u < l.usteps; u++)
      for (int v = 0; v < l.vsteps; v++) {
        Tuple lp = point_on_light(l, light_index, u, v, 0);
        if (!is_shadowed(lp, p)) total += 1.f;
      }
    return total / (float)(l.usteps * l.vsteps);
  }

  // ---- materials.rs:81-177 ---------------------------------------------------------------
  Tuple lighting(const Material& m, const Shape& shape, const Light& light, int light_index, Tuple pt, Tuple eyev,
                 Tuple normalv, float light_intensity) const {
    Tuple col = m.pattern >= 0 ? pattern_at_shape(patterns[m.pattern], shape, pt) : m.color;
    Tuple effective = mul(col, light.intensity);
    Tuple ambient = cscale(effective, m.ambient);
    auto sample = [&](Tuple light_pos, Tuple* diffuse, Tuple* specular) -> bool {
      Tuple lightv = normalize(sub(light_pos, pt));
      float light_dot_normal = dot(lightv, normalv);
      if (!(light_dot_normal >= 0.f)) return false;
      *diffuse = cscale(cscale(effective, m.diffuse), light_dot_normal);
      Tuple reflectv = reflect(neg(lightv), normalv);
      float reflect_dot_eye = dot(reflectv, eyev);
      if (reflect_dot_eye > 0.f) {
        float factor = powf(reflect_dot_eye, m.shininess);
        *specular = cscale(cscale(light.intensity, m.specular), factor);
      } else {
        *specular = color(0.f, 0.f, 0.f);
      }
      return true;
    };
    if (light.kind == 0) {
      Tuple diffuse = color(0.f, 0.f, 0.f), specular = color(0.f, 0.f, 0.f);
      sample(light.position, &diffuse, &specular);
      return add(ambient, cscale(add(diffuse, specular), light_intensity));
    }
    Tuple sum = color(0.f, 0.f, 0.f);
    for (int u = 0; u < light.usteps; u++)
      for (int v = 0; v < light.vsteps; v++) {
        Tuple lp = point_on_light(light, light_index, u, v, 1);
        Tuple diffuse, specular;
        if (sample(lp, &diffuse, &specular)) sum = add(add(sum, diffuse), specular);
      }
    sum = cscale(sum, 1.f / (float)(light.usteps * light.vsteps));
    return add(ambient, cscale(sum, light_intensity));
  }

  // ---- world.rs:64-148 -------------------------------------------------------------------
  Tuple reflected_color(const Comps& c, int remaining) const {  // :113-125
    if (remaining == 0) return color(0.f, 0.f, 0.f);
    if (c.shape->material.reflective == 0.f) return color(0.f, 0.f, 0.f);
    Ray r{c.over_point, c.reflectv};
    uint32_t saved = g_ctx.path;
    g_ctx.path = saved * 2u;
    g_ctx.n_reflect++;
    Tuple col = color_at(r, remaining - 1);
    g_ctx.path = saved;
    return cscale(col, c.shape->material.reflective);
  }
  Tuple refracted_color(const Comps& c, int remaining) const {  // :127-148
    if (remaining == 0) return color(0.f, 0.f, 0.f);
    if (c.shape->material.transparency == 0.f) return color(0.f, 0.f, 0.f);
    float n_ratio = c.n1 / c.n2;
    float cos_i = dot(c.eyev, c.normalv);
    float sin2_t = (n_ratio * n_ratio) * (1.f - cos_i * cos_i);
    if (sin2_t > 1.f) return color(0.f, 0.f, 0.f);
    float cos_t = std::sqrt(1.f - sin2_t);
    Tuple direction = sub(scale(n_ratio * cos_i - cos_t, c.normalv), scale(n_ratio, c.eyev));
    Ray r{c.under_point, direction};
    uint32_t saved = g_ctx.path;
    g_ctx.path = saved * 2u + 1u;
    g_ctx.n_refract++;
    Tuple col = color_at(r, remaining - 1);
    g_ctx.path = saved;
    return cscale(col, c.shape->material.transparency);
  }
  Tuple shade_hit(const Comps& c, int remaining) const {  // :64-89
    Tuple col = color(0.f, 0.f, 0.f);
    for (size_t li = 0; li < lights.size(); li++) {
      const Light& light = lights[li];
      float li_intensity = intensity_at(light, (int)li, c.over_point);
      const Material& m = c.shape->material;
      Tuple surface = lighting(m, *c.shape, light, (int)li, c.over_point, c.eyev, c.normalv, li_intensity);
      Tuple reflected = reflected_color(c, remaining);
      Tuple refracted = refracted_color(c, remaining);
      if (m.reflective > 0.f && m.transparency > 0.f) {
        float reflectance = schlick(c);
        col = add(col, add(add(surface, cscale(reflected, reflectance)), cscale(refracted, 1.f - reflectance)));
      } else {
        col = add(col, add(add(surface, reflected), refracted));
      }
    }
    return col;
  }
  Tuple color_at(const Ray& ray, int remaining, Intersection* primary_hit = nullptr) const {  // :91-99
    auto xs = intersect(ray);
    const Intersection* h = hit(xs);
    if (primary_hit) {
      if (h)
        *primary_hit = *h;
      else
        *primary_hit = Intersection{std::numeric_limits<float>::quiet_NaN(), nullptr, 0.f, 0.f};
    }
    if (!h) return color(0.f, 0.f, 0.f);
    Comps c = prepare(*h, ray, xs);
    return shade_hit(c, remaining);
  }
};

namespace {

// ---------------------------------------------------------------------------------------
// camera.rs
// ---------------------------------------------------------------------------------------
struct Camera {
  uint32_t hsize, vsize;
  float fov;
  Mat4 transform, inv;
  float pixel_size, half_width, half_height;
};
bool make_camera(const RtbCamera* d, Camera* c) {  // camera.rs:19-43
  c->hsize = d->hsize;
  c->vsize = d->vsize;
  c->fov = d->field_of_view;
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
  if (d->use_view)
    c->transform = view_transform(point(d->from[0], d->from[1], d->from[2]), point(d->to[0], d->to[1], d->to[2]),
                                  vector(d->up[0], d->up[1], d->up[2]));
  else
    c->transform = apply_ops(d->ops, d->nops);
  return inverse(c->transform, &c->inv);
}
Ray ray_for_pixel(const Camera& c, uint32_t px, uint32_t py) {  // camera.rs:45-64
  float xoffset = ((float)px + 0.5f) * c.pixel_size;
  float yoffset = ((float)py + 0.5f) * c.pixel_size;
  float world_x = c.half_width - xoffset;
  float world_y = c.half_height - yoffset;
  Tuple pixel = mat_point(c.inv, point(world_x, world_y, -1.f));
  Tuple origin = mat_point(c.inv, point(0.f, 0.f, 0.f));
  Tuple direction = normalize(sub(pixel, origin));
  return Ray{origin, direction};
}

void setup_ctx(const RtbWorld* w, const RtbRenderOptions* o) {
  Ctx& cx = g_ctx;
  cx = Ctx{};
  if (o) {
    cx.jitter_mode = o->jitter_mode;
    cx.jitter_value = o->jitter_value;
    cx.seed = o->seed;
  }
  cx.seq = &w->jitter_seq;
}

Material material_from(const RtbMaterial* m) {
  Material r;
  if (!m) return r;
  r.pattern = m->pattern;
  r.color = color(m->color[0], m->color[1], m->color[2]);
  r.ambient = m->ambient;
  r.diffuse = m->diffuse;
  r.specular = m->specular;
  r.shininess = m->shininess;
  r.reflective = m->reflective;
  r.transparency = m->transparency;
  r.refractive_index = m->refractive_index;
  return r;
}

Shape make_triangle(Tuple p1, Tuple p2, Tuple p3) {  // triangle.rs:32-48
  Shape s;
  s.kind = RTB_TRIANGLE;
  s.p1 = p1;
  s.p2 = p2;
  s.p3 = p3;
  s.e1 = sub(p2, p1);
  s.e2 = sub(p3, p1);
  s.normal = normalize(cross(s.e2, s.e1));
  return s;
}
Shape make_smooth_triangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3) {  // smooth_triangle.rs:34-51
  Shape s;
  s.kind = RTB_SMOOTH_TRIANGLE;
  s.p1 = p1;
  s.p2 = p2;
  s.p3 = p3;
  s.e1 = sub(p2, p1);
  s.e2 = sub(p3, p1);
  s.n1 = n1;
  s.n2 = n2;
  s.n3 = n3;
  return s;
}

int fail(RtbWorld* w, const char* msg) {
  w->error = msg;
  return -1;
}

}  // namespace

// =========================================================================================
// builder API (include/rt_builder.h) with the orc_ prefix
// =========================================================================================
extern "C" {

RtbWorld* orc_world_new(void) { return new RtbWorld(); }
void orc_world_free(RtbWorld* w) { delete w; }
const char* orc_last_error(const RtbWorld* w) { return w->error.c_str(); }

int orc_uv_checker(RtbWorld* w, float width, float height, const float a[3], const float b[3]) {
  UvPattern p;
  p.kind = 0;
  p.width = width;
  p.height = height;
  p.c[0] = color(a[0], a[1], a[2]);
  p.c[1] = color(b[0], b[1], b[2]);
  w->uvpatterns.push_back(p);
  return (int)w->uvpatterns.size() - 1;
}
int orc_uv_align_check(RtbWorld* w, const float c[15]) {
  UvPattern p;
  p.kind = 1;
  for (int i = 0; i < 5; i++) p.c[i] = color(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
  w->uvpatterns.push_back(p);
  return (int)w->uvpatterns.size() - 1;
}
int orc_uv_image(RtbWorld* w, uint32_t width, uint32_t height, const float* rgb) {
  Image im;
  im.w = width;
  im.h = height;
  im.rgb.assign(rgb, rgb + size_t(width) * height * 3);
  w->images.push_back(std::move(im));
  UvPattern p;
  p.kind = 2;
  p.image = (int)w->images.size() - 1;
  w->uvpatterns.push_back(p);
  return (int)w->uvpatterns.size() - 1;
}
int orc_pattern_new(RtbWorld* w, int kind, const float colors[6], const RtbXfOp* ops, int nops, int uv_mapping,
                    const int32_t* uvpats) {
  Pattern p{};
  p.kind = kind;
  if (colors) {
    p.a = color(colors[0], colors[1], colors[2]);
    p.b = color(colors[3], colors[4], colors[5]);
  }
  p.transform = apply_ops(ops, nops);
  if (!inverse(p.transform, &p.inv)) return fail(w, "pattern transform is not invertible");
  p.uv_mapping = uv_mapping;
  for (int i = 0; i < 6; i++) p.uv[i] = -1;
  if (kind == RTB_PAT_TEXTURE_MAP) p.uv[0] = uvpats[0];
  if (kind == RTB_PAT_CUBE_MAP)
    for (int i = 0; i < 6; i++) p.uv[i] = uvpats[i];
  w->patterns.push_back(p);
  return (int)w->patterns.size() - 1;
}

int orc_shape_new(RtbWorld* w, int kind, const RtbXfOp* ops, int nops, const RtbMaterial* material, int shadow,
                  const float* params, int nparams) {
  Shape s;
  if (kind == RTB_TRIANGLE) {
    if (nparams < 9) return fail(w, "triangle needs 9 params");
    s = make_triangle(point(params[0], params[1], params[2]), point(params[3], params[4], params[5]),
                      point(params[6], params[7], params[8]));
  } else if (kind == RTB_SMOOTH_TRIANGLE) {
    if (nparams < 18) return fail(w, "smooth triangle needs 18 params");
    s = make_smooth_triangle(point(params[0], params[1], params[2]), point(params[3], params[4], params[5]),
                             point(params[6], params[7], params[8]), vector(params[9], params[10], params[11]),
                             vector(params[12], params[13], params[14]), vector(params[15], params[16], params[17]));
  } else {
    s.kind = kind;
    if ((kind == RTB_CYLINDER || kind == RTB_CONE) && params && nparams >= 3) {
      s.mn = params[0];
      s.mx = params[1];
      s.closed = params[2] != 0.f;
    }
  }
  s.transform = apply_ops(ops, nops);
  if (!inverse(s.transform, &s.inv)) return fail(w, "shape transform is not invertible");
  s.material = material_from(material);
  s.shadow = shadow != 0;
  w->pending_shapes.push_back(s);
  return (int)w->pending_shapes.size() - 1;
}

int orc_group_new(RtbWorld* w, const RtbXfOp* ops, int nops) {
  GroupData g;
  g.transform = apply_ops(ops, nops);
  if (!inverse(g.transform, &g.inv)) return fail(w, "group transform is not invertible");
  w->thaw();
  return w->group_insert(std::move(g));
}
int orc_group_set_transform(RtbWorld* w, int group, const RtbXfOp* ops, int nops) {
  GroupData& g = w->group_table[group];
  g.transform = apply_ops(ops, nops);
  if (!inverse(g.transform, &g.inv)) return fail(w, "group transform is not invertible");
  w->thaw();
  return 0;
}
int orc_group_add_shape(RtbWorld* w, int group, int shape) {
  w->thaw();
  w->group_add_shape(group, w->pending_shapes[shape]);
  return 0;
}
int orc_group_add_child(RtbWorld* w, int group, int child) {
  w->thaw();
  w->group_add_child(group, child);
  return 0;
}
int orc_group_add_mesh(RtbWorld* w, int group, const RtbMaterial* material, const float* positions, int npositions,
                       const float* normals, int nnormals, const int32_t* vidx, const int32_t* nidx, int ntriangles) {
  // obj.rs:12-38: one child Group per OBJ group, triangles added in file order.
  w->thaw();
  GroupData gd;
  int child = w->group_insert(std::move(gd));
  Material m = material_from(material);
  w->group_table[child].shapes.reserve(ntriangles);
  for (int t = 0; t < ntriangles; t++) {
    Tuple p[3];
    for (int c = 0; c < 3; c++) {
      int vi = vidx[3 * t + c];
      if (vi < 0 || vi >= npositions) return fail(w, "mesh position index out of range");
      p[c] = point(positions[3 * vi], positions[3 * vi + 1], positions[3 * vi + 2]);
    }
    bool smooth = nidx != nullptr && normals != nullptr;
    Tuple n[3];
    if (smooth)
      for (int c = 0; c < 3; c++) {
        int ni = nidx[3 * t + c];
        if (ni < 0) {
          smooth = false;
          break;
        }
        if (ni >= nnormals) return fail(w, "mesh normal index out of range");
        n[c] = vector(normals[3 * ni], normals[3 * ni + 1], normals[3 * ni + 2]);
      }
    Shape s = smooth ? make_smooth_triangle(p[0], p[1], p[2], n[0], n[1], n[2]) : make_triangle(p[0], p[1], p[2]);
    s.material = m;
    w->group_add_shape(child, s);
  }
  w->group_add_child(group, child);
  return child;
}
int orc_group_divide(RtbWorld* w, int group, int threshold) {
  w->thaw();
  w->divide(group, threshold);
  return 0;
}

int orc_csg_new(RtbWorld* w, int op, int left_kind, int left, int right_kind, int right) {
  auto c = std::make_shared<Csg>();
  c->op = op;
  auto set = [&](CsgChild& ch, int kind, int h) {
    ch.kind = kind;
    if (kind == RTB_CHILD_SHAPE)
      ch.shape = w->pending_shapes[h];
    else
      ch.csg = w->pending_csgs[h];
  };
  set(c->left, left_kind, left);
  set(c->right, right_kind, right);
  w->pending_csgs.push_back(c);
  return (int)w->pending_csgs.size() - 1;
}

int orc_world_add_shape(RtbWorld* w, int shape) {
  w->thaw();
  w->shapes.push_back(w->pending_shapes[shape]);
  return (int)w->shapes.size() - 1;
}
int orc_world_add_group(RtbWorld* w, int group, int cache_bounds) {
  w->thaw();
  if (cache_bounds) {  // groups.rs:190-196
    BBox bb = w->make_bounds(group);
    w->group_table[group].bounds = bb;
    w->group_table[group].has_bounds = true;
  }
  w->groups.push_back(group);
  return 0;
}
int orc_world_add_csg(RtbWorld* w, int csg) {
  w->thaw();
  w->csgs.push_back(w->pending_csgs[csg]);
  return 0;
}
int orc_world_add_point_light(RtbWorld* w, const float position[3], const float intensity[3]) {
  Light l{};
  l.kind = 0;
  l.position = point(position[0], position[1], position[2]);
  l.intensity = color(intensity[0], intensity[1], intensity[2]);
  w->lights.push_back(l);
  return (int)w->lights.size() - 1;
}
int orc_world_add_area_light(RtbWorld* w, const float corner[3], const float fu[3], int usteps, const float fv[3],
                             int vsteps, const float intensity[3]) {
  // lights.rs:83-103
  Light l{};
  l.kind = 1;
  Tuple full_u = vector(fu[0], fu[1], fu[2]), full_v = vector(fv[0], fv[1], fv[2]);
  l.corner = point(corner[0], corner[1], corner[2]);
  l.position = add(l.corner, divs(add(full_u, full_v), 2.f));
  l.uvec = divs(full_u, (float)usteps);
  l.vvec = divs(full_v, (float)vsteps);
  l.usteps = usteps;
  l.vsteps = vsteps;
  l.intensity = color(intensity[0], intensity[1], intensity[2]);
  w->lights.push_back(l);
  return (int)w->lights.size() - 1;
}
int orc_world_set_recursion_limit(RtbWorld* w, int limit) {
  w->recursion_limit = limit;
  return 0;
}

int orc_render(RtbWorld* w, const RtbCamera* cam, const RtbRenderOptions* opts, float* out_rgb, RtbAux* aux,
               RtbStats* stats) {
  Camera c;
  if (!make_camera(cam, &c)) return fail(w, "camera transform is not invertible");
  w->freeze();
  // default: the box's CPU share, not the machine (hardware_concurrency() reports every core of the host there)
  int threads = opts && opts->threads > 0 ? opts->threads : (int)std::min(16u, std::thread::hardware_concurrency());
  if (threads < 1) threads = 1;
  int row_begin = opts ? opts->row_begin : 0;
  int row_end = (opts && opts->row_end > 0) ? opts->row_end : (int)c.vsize;
  int row_step = (opts && opts->row_step > 0) ? opts->row_step : 1;
  int col_step = (opts && opts->col_step > 0) ? opts->col_step : 1;
  std::vector<int> rows;
  for (int r = row_begin; r < row_end; r += row_step) rows.push_back(r);
  std::atomic<size_t> next{0};
  std::atomic<uint64_t> n_primary{0}, n_shadow{0}, n_reflect{0}, n_refract{0};
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    setup_ctx(w, opts);
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= rows.size()) break;
      uint32_t py = (uint32_t)rows[i];
      for (uint32_t px = 0; px < c.hsize; px += col_step) {
        g_ctx.pixel = py * c.hsize + px;
        g_ctx.path = 1;
        g_ctx.n_primary++;
        Ray ray = ray_for_pixel(c, px, py);
        Intersection ph;
        Tuple col = w->color_at(ray, w->recursion_limit, &ph);
        size_t idx = size_t(py) * c.hsize + px;
        out_rgb[3 * idx + 0] = col.v[0];
        out_rgb[3 * idx + 1] = col.v[1];
        out_rgb[3 * idx + 2] = col.v[2];
        if (aux) {
          if (aux->hit_mask) aux->hit_mask[idx] = ph.shape ? 1 : 0;
          if (aux->object_id) aux->object_id[idx] = ph.shape ? ph.shape->rank : -1;
          if (aux->hit_t) aux->hit_t[idx] = ph.t;
        }
      }
    }
    n_primary += g_ctx.n_primary;
    n_shadow += g_ctx.n_shadow;
    n_reflect += g_ctx.n_reflect;
    n_refract += g_ctx.n_refract;
  };
  if (threads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; i++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }
  auto t1 = std::chrono::steady_clock::now();
  if (stats) {
    stats->primary = n_primary;
    stats->shadow = n_shadow;
    stats->reflect = n_reflect;
    stats->refract = n_refract;
    stats->overflow = 0;
    stats->seconds = std::chrono::duration<double>(t1 - t0).count();
  }
  return 0;
}

int orc_color_at(RtbWorld* w, const RtbRenderOptions* opts, const float* origins, const float* directions, int n,
                 int remaining, float* out_rgb, float* hit_t, int32_t* hit_id, float* hit_uv) {
  w->freeze();
  // rays are independent (the jitter key is the ray index): worker threads take chunks of 64
  int threads = opts && opts->threads > 0 ? opts->threads : (int)std::min(16u, std::thread::hardware_concurrency());
  if (threads < 1 || n < 256 || (opts && opts->jitter_mode == RTB_JITTER_SEQUENCE)) threads = 1;  // the cycling fake is one sequence
  std::atomic<int> next{0};
  auto worker = [&]() {
    setup_ctx(w, opts);
    for (;;) {
      const int b = next.fetch_add(64);
      if (b >= n) break;
      for (int i = b; i < std::min(n, b + 64); i++) {
        Ray r{point(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
              vector(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2])};
        g_ctx.pixel = (uint32_t)i;
        g_ctx.path = 1;
        Intersection ph;
        Tuple col = w->color_at(r, remaining, &ph);
        if (out_rgb) {
          out_rgb[3 * i] = col.v[0];
          out_rgb[3 * i + 1] = col.v[1];
          out_rgb[3 * i + 2] = col.v[2];
        }
        if (hit_t) hit_t[i] = ph.t;
        if (hit_id) hit_id[i] = ph.shape ? ph.shape->rank : -1;
        if (hit_uv) {
          hit_uv[2 * i] = ph.shape ? ph.u : 0.f;
          hit_uv[2 * i + 1] = ph.shape ? ph.v : 0.f;
        }
      }
    }
  };
  if (threads == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; i++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// What Scene::render does with the canvas after Camera::render (scene.rs:215-224): ordered dithering
// (canvas.rs:37-55 over the Bayer matrix of effects/dithering.rs:12-37), then the 8-bit quantisation of to_ppm
// (canvas.rs:73-75, the same rule as tuples.rs:367-377).  Restated for the device's u8 output path.
// ---------------------------------------------------------------------------------------
// effects/dithering.rs:12-37: out[i * n + j] = bayer[i][j]
int orc_bayer_matrix(int n, float* out) {
  if (n < 2 || (n & (n - 1)) != 0 || n > 64) return -1;  // is_pow2 (:39-41)
  std::vector<float> res((size_t)n * n, 0.f);
  auto at = [&](int i, int j) -> float& { return res[(size_t)i * n + j]; };
  for (int i = 0; i < n; i++) at(i, i) = 1.f;  // Matrix::identity()
  at(0, 0) = 0.f;
  at(0, 1) = 2.f;
  at(1, 0) = 3.f;
  at(1, 1) = 1.f;
  int size = 2;
  int log2n = 0;
  while ((1 << log2n) < n) log2n++;
  for (int step = 1; step < log2n; step++) {
    for (int j = 0; j < size; j++)
      for (int i = 0; i < size; i++) at(i, j) *= powf(2.f, 2.f);
    for (int j = 0; j < size; j++)
      for (int i = 0; i < size; i++) {
        at(size + i, j) = at(i, j) + 3.f;
        at(i, size + j) = at(i, j) + 2.f;
        at(size + i, size + j) = at(i, j) + 1.f;
      }
    size *= 2;
  }
  const float k = 1.f / (float)(n * n);  // (1. / res.size().0.pow(2) as f32) * res
  for (float& v : res) v = k * v;
  std::memcpy(out, res.data(), res.size() * sizeof(float));
  return 0;
}
static inline uint8_t quantise(float c) {  // (0_f32.max(c).min(1.) * 255.).round() as u8
  float lo = std::isnan(c) ? 0.f : (c > 0.f ? c : 0.f);  // f32::max returns the other operand for a NaN
  float cl = lo < 1.f ? lo : 1.f;
  float r = roundf(cl * 255.f);  // half away from zero
  return (uint8_t)r;
}
// dither: 0 none, 1..4 Bayer2/4/8/16 (black or white), 5 BayerColor (N = 4, added to the colour) -- scene.rs:216-221.
// rgb: w*h*3 f32 row-major (the canvas of Camera::render); out: w*h*3 u8.  Row r of the buffer is image row rows[r]
// (NULL: r itself), so that a part of an interleaved frame dithers like the whole frame does.
int orc_canvas_to_u8(const float* rgb, uint32_t w, uint32_t h, int dither, const uint32_t* rows, uint8_t* out) {
  const int ns[6] = {0, 2, 4, 8, 16, 4};
  if (dither < 0 || dither > 5) return -1;
  const int n = ns[dither];
  std::vector<float> bayer;
  if (n) {
    bayer.resize((size_t)n * n);
    orc_bayer_matrix(n, bayer.data());
  }
  for (uint32_t r = 0; r < h; r++) {
    const uint32_t y = rows ? rows[r] : r;
    for (uint32_t x = 0; x < w; x++) {
      const float* p = rgb + ((size_t)r * w + x) * 3;
      float c[3] = {p[0], p[1], p[2]};
      if (n) {
        const float d = bayer[(size_t)(x % n) * n + (y % n)];  // bayer[x % N][y % N] (canvas.rs:41)
        if (dither == 5) {
          for (float& v : c) v = v + d;  // pixel + Color::new(d, d, d)
        } else {
          const bool black = c[0] < d || c[1] < d || c[2] < d;
          c[0] = c[1] = c[2] = black ? 0.f : 1.f;
        }
      }
      uint8_t* o = out + ((size_t)r * w + x) * 3;
      for (int k = 0; k < 3; k++) o[k] = quantise(c[k]);
    }
  }
  return 0;
}

int orc_ray_for_pixel(RtbWorld* w, const RtbCamera* cam, uint32_t px, uint32_t py, float out[6], float* pixel_size) {
  Camera c;
  if (!make_camera(cam, &c)) return fail(w, "camera transform is not invertible");
  Ray r = ray_for_pixel(c, px, py);
  for (int i = 0; i < 3; i++) {
    out[i] = r.origin.v[i];
    out[3 + i] = r.direction.v[i];
  }
  if (pixel_size) *pixel_size = c.pixel_size;
  return 0;
}

// =========================================================================================
// probes: expose the individual reference functions so tests can restate the reference's own
// unit-test known answers.  Shape references: (refkind, a, b) with
//   refkind 0: detached shape handle a (as returned by shape_new)
//   refkind 1: world.shapes[a]
//   refkind 2: group a, shape index b
// =========================================================================================
static const Shape* probe_shape(RtbWorld* w, int refkind, int a, int b) {
  if (refkind == 0) return &w->pending_shapes[a];
  if (refkind == 1) return &w->shapes[a];
  return &w->group_table[a].shapes[b];
}

void orc_probe_set_jitter_sequence(RtbWorld* w, const float* seq, int n) { w->jitter_seq.assign(seq, seq + n); }

void orc_probe_matrix_inverse(const float m[16], float out[16], int* ok) {
  Mat4 a, r;
  std::memcpy(a.m, m, sizeof(a.m));
  *ok = inverse(a, &r) ? 1 : 0;
  if (*ok) std::memcpy(out, r.m, sizeof(r.m));
}
void orc_probe_matmul(const float a[16], const float b[16], float out[16]) {
  Mat4 x, y;
  std::memcpy(x.m, a, sizeof(x.m));
  std::memcpy(y.m, b, sizeof(y.m));
  Mat4 r = matmul(x, y);
  std::memcpy(out, r.m, sizeof(r.m));
}
void orc_probe_transform(const RtbXfOp* ops, int nops, float out[16]) {
  Mat4 r = apply_ops(ops, nops);
  std::memcpy(out, r.m, sizeof(r.m));
}
void orc_probe_view_transform(const float from[3], const float to[3], const float up[3], float out[16]) {
  Mat4 r = view_transform(point(from[0], from[1], from[2]), point(to[0], to[1], to[2]), vector(up[0], up[1], up[2]));
  std::memcpy(out, r.m, sizeof(r.m));
}
void orc_probe_mat_apply(const float m[16], const float t[4], float out[4]) {
  Mat4 a;
  std::memcpy(a.m, m, sizeof(a.m));
  Tuple r = mat_tuple(a, tup(t[0], t[1], t[2], t[3]));
  std::memcpy(out, r.v, sizeof(r.v));
}
void orc_probe_reflect(const float in[3], const float n[3], float out[3]) {
  Tuple r = reflect(vector(in[0], in[1], in[2]), vector(n[0], n[1], n[2]));
  std::memcpy(out, r.v, 3 * sizeof(float));
}

// World::intersect (sorted, deduplicated): returns count, fills up to max entries.
int orc_probe_intersect(RtbWorld* w, const float o[3], const float d[3], float* ts, int32_t* ids, int max) {
  w->freeze();
  setup_ctx(w, nullptr);
  auto xs = w->intersect(Ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])});
  for (int i = 0; i < (int)xs.size() && i < max; i++) {
    if (ts) ts[i] = xs[i].t;
    if (ids) ids[i] = xs[i].shape->rank;
  }
  return (int)xs.size();
}
// Shape::intersect on one shape (applies the shape's own transform), insertion order preserved.
int orc_probe_shape_intersect(RtbWorld* w, int refkind, int a, int b, const float o[3], const float d[3], float* ts,
                              float* uvs, int max) {
  std::vector<Intersection> xs;
  RtbWorld::shape_intersect(*probe_shape(w, refkind, a, b), xs, Ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])});
  for (int i = 0; i < (int)xs.size() && i < max; i++) {
    ts[i] = xs[i].t;
    if (uvs) {
      uvs[2 * i] = xs[i].u;
      uvs[2 * i + 1] = xs[i].v;
    }
  }
  return (int)xs.size();
}
// Group::intersect for a group handle (transforms by the group's own inverse first).
int orc_probe_group_intersect(RtbWorld* w, int group, const float o[3], const float d[3], float* ts, int max) {
  std::vector<Intersection> xs;
  bool was = w->frozen;
  w->frozen = false;
  w->group_intersect(group, xs, Ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])});
  w->frozen = was;
  w->make_intersections(xs);
  for (int i = 0; i < (int)xs.size() && i < max; i++) ts[i] = xs[i].t;
  return (int)xs.size();
}
void orc_probe_local_normal_at(RtbWorld* w, int refkind, int a, int b, const float p[3], float u, float v, float out[3]) {
  Tuple n = RtbWorld::local_normal_at(*probe_shape(w, refkind, a, b), point(p[0], p[1], p[2]), u, v);
  std::memcpy(out, n.v, 3 * sizeof(float));
}
void orc_probe_normal_at(RtbWorld* w, int refkind, int a, int b, const float p[3], float u, float v, float out[3]) {
  Tuple n = w->normal_at(*probe_shape(w, refkind, a, b), point(p[0], p[1], p[2]), u, v);
  std::memcpy(out, n.v, 3 * sizeof(float));
}
void orc_probe_world_to_object(RtbWorld* w, int refkind, int a, int b, const float p[3], float out[3]) {
  Tuple r = w->world_to_object(*probe_shape(w, refkind, a, b), point(p[0], p[1], p[2]));
  std::memcpy(out, r.v, 3 * sizeof(float));
}
void orc_probe_normal_to_world(RtbWorld* w, int refkind, int a, int b, const float n[3], float out[3]) {
  Tuple r = w->normal_to_world(*probe_shape(w, refkind, a, b), vector(n[0], n[1], n[2]));
  std::memcpy(out, r.v, 3 * sizeof(float));
}

// Computations / shading on an explicit intersection list (as the reference tests build them):
// entry i = (ts[i], shape ref (refkind[i], ra[i], rb[i]), uvs[2i..]) ; `index` selects the hit.
// what: 0 comps -> out[0..22] = point3, over3, under3, eyev3, normalv3, reflectv3, inside, n1, n2, t, schlick
//       1 shade_hit, 2 reflected_color, 3 refracted_color -> out[0..2]
int orc_probe_comps(RtbWorld* w, const float o[3], const float d[3], int n, const float* ts, const int32_t* refkind,
                    const int32_t* ra, const int32_t* rb, const float* uvs, int index, int remaining, int what,
                    float* out) {
  w->freeze();
  setup_ctx(w, nullptr);
  Ray ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])};
  std::vector<Intersection> xs;
  for (int i = 0; i < n; i++)
    xs.push_back({ts[i], probe_shape(w, refkind[i], ra[i], rb[i]), uvs ? uvs[2 * i] : F32_MAX, uvs ? uvs[2 * i + 1] : F32_MAX});
  Intersection hit = xs[index];
  // the reference tests pass either the same list or an empty one; n < 0 is not used
  std::vector<Intersection> sorted = xs;
  w->make_intersections(sorted);
  RtbWorld::Comps c = w->prepare(hit, ray, sorted);
  if (what == 0) {
    const Tuple* vs[6] = {&c.point, &c.over_point, &c.under_point, &c.eyev, &c.normalv, &c.reflectv};
    for (int k = 0; k < 6; k++)
      for (int j = 0; j < 3; j++) out[3 * k + j] = vs[k]->v[j];
    out[18] = c.inside ? 1.f : 0.f;
    out[19] = c.n1;
    out[20] = c.n2;
    out[21] = c.t;
    out[22] = RtbWorld::schlick(c);
    return 0;
  }
  Tuple col;
  if (what == 1)
    col = w->shade_hit(c, remaining);
  else if (what == 2)
    col = w->reflected_color(c, remaining);
  else
    col = w->refracted_color(c, remaining);
  std::memcpy(out, col.v, 3 * sizeof(float));
  return 0;
}
// Same but with an EMPTY intersection list passed to prepare (reference: `&vec![].into()`).
int orc_probe_comps_empty(RtbWorld* w, const float o[3], const float d[3], float t, int refkind, int ra, int rb,
                          int remaining, int what, float* out) {
  w->freeze();
  setup_ctx(w, nullptr);
  Ray ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])};
  Intersection hit{t, probe_shape(w, refkind, ra, rb), F32_MAX, F32_MAX};
  std::vector<Intersection> none;
  RtbWorld::Comps c = w->prepare(hit, ray, none);
  if (what == 0) {
    const Tuple* vs[6] = {&c.point, &c.over_point, &c.under_point, &c.eyev, &c.normalv, &c.reflectv};
    for (int k = 0; k < 6; k++)
      for (int j = 0; j < 3; j++) out[3 * k + j] = vs[k]->v[j];
    out[18] = c.inside ? 1.f : 0.f;
    out[19] = c.n1;
    out[20] = c.n2;
    out[21] = c.t;
    out[22] = RtbWorld::schlick(c);
    return 0;
  }
  Tuple col;
  if (what == 1)
    col = w->shade_hit(c, remaining);
  else if (what == 2)
    col = w->reflected_color(c, remaining);
  else
    col = w->refracted_color(c, remaining);
  std::memcpy(out, col.v, 3 * sizeof(float));
  return 0;
}
int orc_probe_is_shadowed(RtbWorld* w, const float light_pos[3], const float p[3]) {
  w->freeze();
  setup_ctx(w, nullptr);
  return w->is_shadowed(point(light_pos[0], light_pos[1], light_pos[2]), point(p[0], p[1], p[2])) ? 1 : 0;
}
float orc_probe_intensity_at(RtbWorld* w, int light, const float p[3], int jitter_mode, float jitter_value) {
  w->freeze();
  RtbRenderOptions o{};
  o.jitter_mode = jitter_mode;
  o.jitter_value = jitter_value;
  setup_ctx(w, &o);
  return w->intensity_at(w->lights[light], light, point(p[0], p[1], p[2]));
}
void orc_probe_point_on_light(RtbWorld* w, int light, int u, int v, int jitter_mode, float jitter_value, float out[3]) {
  RtbRenderOptions o{};
  o.jitter_mode = jitter_mode;
  o.jitter_value = jitter_value;
  setup_ctx(w, &o);
  Tuple p = w->point_on_light(w->lights[light], light, u, v, 0);
  std::memcpy(out, p.v, 3 * sizeof(float));
}
void orc_probe_light_position(RtbWorld* w, int light, float out[3]) {
  std::memcpy(out, w->lights[light].position.v, 3 * sizeof(float));
}
void orc_probe_lighting(RtbWorld* w, const RtbMaterial* m, int refkind, int a, int b, int light, const float pt[3],
                        const float eyev[3], const float normalv[3], float intensity, int jitter_mode,
                        float jitter_value, float out[3]) {
  RtbRenderOptions o{};
  o.jitter_mode = jitter_mode;
  o.jitter_value = jitter_value;
  setup_ctx(w, &o);
  Material mat = material_from(m);
  Tuple c = w->lighting(mat, *probe_shape(w, refkind, a, b), w->lights[light], light, point(pt[0], pt[1], pt[2]),
                        vector(eyev[0], eyev[1], eyev[2]), vector(normalv[0], normalv[1], normalv[2]), intensity);
  std::memcpy(out, c.v, 3 * sizeof(float));
}
void orc_probe_pattern_at(RtbWorld* w, int pattern, const float p[3], float out[3]) {
  Tuple c = w->pattern_at(w->patterns[pattern], point(p[0], p[1], p[2]));
  std::memcpy(out, c.v, 3 * sizeof(float));
}
void orc_probe_pattern_at_shape(RtbWorld* w, int pattern, int refkind, int a, int b, const float p[3], float out[3]) {
  Tuple c = w->pattern_at_shape(w->patterns[pattern], *probe_shape(w, refkind, a, b), point(p[0], p[1], p[2]));
  std::memcpy(out, c.v, 3 * sizeof(float));
}
void orc_probe_uv_map(int mapping, const float p[3], float out[2]) {
  RtbWorld::uv_map(mapping, point(p[0], p[1], p[2]), &out[0], &out[1]);
}
void orc_probe_uv_pattern_at(RtbWorld* w, int uvpat, float u, float v, float out[3]) {
  Tuple c = w->uv_pattern_at(uvpat, u, v);
  std::memcpy(out, c.v, 3 * sizeof(float));
}
int orc_probe_cube_face(const float p[3]) { return RtbWorld::cube_face(point(p[0], p[1], p[2])); }
void orc_probe_cube_uv(int face, const float p[3], float out[2]) {
  RtbWorld::cube_uv(face, point(p[0], p[1], p[2]), &out[0], &out[1]);
}
void orc_probe_shape_bounds(RtbWorld* w, int refkind, int a, int b, int parent_space, float out[6]) {
  const Shape* s = probe_shape(w, refkind, a, b);
  BBox bb = parent_space ? RtbWorld::parent_space_bounds(*s) : RtbWorld::shape_bounds(*s);
  std::memcpy(out, bb.min.v, 3 * sizeof(float));
  std::memcpy(out + 3, bb.max.v, 3 * sizeof(float));
}
void orc_probe_group_bounds(RtbWorld* w, int group, float out[6]) {
  bool was = w->frozen;
  w->frozen = false;
  BBox bb = w->group_bounds(group);
  w->frozen = was;
  std::memcpy(out, bb.min.v, 3 * sizeof(float));
  std::memcpy(out + 3, bb.max.v, 3 * sizeof(float));
}
void orc_probe_csg_bounds(RtbWorld* w, int csg, float out[6]) {
  BBox bb = w->csg_bounds(*w->pending_csgs[csg]);
  std::memcpy(out, bb.min.v, 3 * sizeof(float));
  std::memcpy(out + 3, bb.max.v, 3 * sizeof(float));
}
// Csg::local_intersect on a detached csg handle: returns count; ts and the detached-shape index
// of each surviving intersection's shape (matched by value against pending_shapes) are returned.
int orc_probe_csg_intersect(RtbWorld* w, int csg, const float o[3], const float d[3], float* ts, int32_t* shape_handles,
                            int max) {
  auto xs = w->csg_local_intersect(*w->pending_csgs[csg], Ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])});
  for (int i = 0; i < (int)xs.size() && i < max; i++) {
    ts[i] = xs[i].t;
    int h = -1;
    for (size_t k = 0; k < w->pending_shapes.size(); k++)
      if (w->shape_eq(w->pending_shapes[k], *xs[i].shape)) {
        h = (int)k;
        break;
      }
    if (shape_handles) shape_handles[i] = h;
  }
  return (int)xs.size();
}
// Csg::filter_intersections on an explicit list of (t, detached shape handle).
int orc_probe_csg_filter(RtbWorld* w, int csg, int n, const float* ts, const int32_t* handles, float* out_ts, int max) {
  std::vector<Intersection> xs;
  for (int i = 0; i < n; i++) xs.push_back({ts[i], &w->pending_shapes[handles[i]], F32_MAX, F32_MAX});
  w->make_intersections(xs);
  auto r = w->csg_filter(*w->pending_csgs[csg], xs);
  for (int i = 0; i < (int)r.size() && i < max; i++) out_ts[i] = r[i].t;
  return (int)r.size();
}
int orc_probe_intersection_allowed(int op, int l_hit, int in_l, int in_r) {
  return RtbWorld::intersection_allowed(op, l_hit != 0, in_l != 0, in_r != 0) ? 1 : 0;
}
int orc_probe_box_intersects(const float mn[3], const float mx[3], const float o[3], const float d[3]) {
  BBox b{point(mn[0], mn[1], mn[2]), point(mx[0], mx[1], mx[2])};
  return bbox_intersects(b, Ray{point(o[0], o[1], o[2]), vector(d[0], d[1], d[2])}) ? 1 : 0;
}
void orc_probe_box_split(const float mn[3], const float mx[3], float out[12]) {
  BBox b{point(mn[0], mn[1], mn[2]), point(mx[0], mx[1], mx[2])}, l, r;
  bbox_split(b, &l, &r);
  std::memcpy(out, l.min.v, 12);
  std::memcpy(out + 3, l.max.v, 12);
  std::memcpy(out + 6, r.min.v, 12);
  std::memcpy(out + 9, r.max.v, 12);
}
void orc_probe_box_transform(const float mn[3], const float mx[3], const RtbXfOp* ops, int nops, float out[6]) {
  BBox b{point(mn[0], mn[1], mn[2]), point(mx[0], mx[1], mx[2])};
  BBox r = bbox_transform(b, apply_ops(ops, nops));
  std::memcpy(out, r.min.v, 12);
  std::memcpy(out + 3, r.max.v, 12);
}
// Group structure introspection for the partition/divide tests.
int orc_probe_group_counts(RtbWorld* w, int group, int* nshapes, int* nchildren) {
  *nshapes = (int)w->group_table[group].shapes.size();
  *nchildren = (int)w->group_table[group].children.size();
  return 0;
}
int orc_probe_group_child(RtbWorld* w, int group, int i) { return w->group_table[group].children[i]; }
int orc_probe_group_parent(RtbWorld* w, int group) { return w->group_table[group].parent; }
int orc_probe_group_count(RtbWorld* w) { return (int)w->group_table.size(); }
// Returns the detached-shape handle equal (by value) to group shape i, or -1.
int orc_probe_group_shape_handle(RtbWorld* w, int group, int i) {
  const Shape& s = w->group_table[group].shapes[i];
  for (size_t k = 0; k < w->pending_shapes.size(); k++)
    if (w->shape_eq(w->pending_shapes[k], s)) return (int)k;
  return -1;
}
int orc_probe_shape_eq(RtbWorld* w, int rk1, int a1, int b1, int rk2, int a2, int b2) {
  return w->shape_eq(*probe_shape(w, rk1, a1, b1), *probe_shape(w, rk2, a2, b2)) ? 1 : 0;
}
int orc_probe_partition(RtbWorld* w, int group, int* left, int* right) {
  w->thaw();
  w->partition(group, left, right);
  return 0;
}
// Triangle fields for the constructor KATs: out = p1,p2,p3,e1,e2,normal (18 floats)
void orc_probe_triangle_fields(RtbWorld* w, int refkind, int a, int b, float out[18]) {
  const Shape* s = probe_shape(w, refkind, a, b);
  const Tuple* vs[6] = {&s->p1, &s->p2, &s->p3, &s->e1, &s->e2, &s->normal};
  for (int k = 0; k < 6; k++)
    for (int j = 0; j < 3; j++) out[3 * k + j] = vs[k]->v[j];
}
uint8_t orc_probe_color_u8(float c) {  // tuples.rs:367-377 / canvas.rs:73-75
  float x = std::fmax(0.f, c);
  x = std::fmin(x, 1.f);
  return (uint8_t)std::round(x * 255.f);
}
float orc_probe_counter_jitter(uint64_t seed, uint32_t pixel, uint32_t path, uint32_t light, uint32_t cell,
                               uint32_t site, uint32_t which) {
  return counter_jitter(seed, pixel, path, light, cell, site, which);
}

}  // extern "C"
