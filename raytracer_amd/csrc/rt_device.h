// rt_device.h — device-side ray tracing core for gfx950 (included only by rt_hip.hip).
//
// Everything here is f32 with NO fused multiply-add (built with -ffp-contract=off) and with
// correctly rounded divide/sqrt, evaluated in the reference's operation order, because the
// parity bar is bit-exact hit / object-id masks against lerouxrgd/raytracer.
// Reference citations are `file:line` relative to the reference root.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"

// Fixed capacities of the device code where the reference's structures are unbounded.  Two sets: the kernels every
// sample scene runs through, and the BIG set of the one slow-path instantiation (the SIDE kernels, which also keep the
// n1/n2 lists in device memory): a world that passes a limit of the first set renders through those instead of being
// refused (rt_hip.hip validate_scene / RtDeviceScene::needs_big); only past the BIG set is it RT_E_SCENE.
#define RT_MAX_XF_DEPTH 6    // nested non-identity group transforms along one path
#define RT_MAX_RAY_STACK 8   // pending secondary rays per pixel (recursion_limit + 1 needed)
#ifndef RT_CSG_CAP
#define RT_CSG_CAP 32        // per-ray intersection entries while evaluating one csg tree
#endif
#define RT_CSG_DEPTH 8       // csg tree depth
#define RT_BIG_XF_DEPTH 32
#define RT_BIG_RAY_STACK 17  // recursion_limit up to 16
#define RT_BIG_CSG_CAP 256
#define RT_BIG_CSG_DEPTH 32
template <bool BIG>
struct Lim {
  static constexpr int kXfDepth = BIG ? RT_BIG_XF_DEPTH : RT_MAX_XF_DEPTH;
  static constexpr int kRayStack = BIG ? RT_BIG_RAY_STACK : RT_MAX_RAY_STACK;
  static constexpr int kCsgCap = BIG ? RT_BIG_CSG_CAP : RT_CSG_CAP;
  static constexpr int kCsgDepth = BIG ? RT_BIG_CSG_DEPTH : RT_CSG_DEPTH;
};
#ifndef RT_XS_CAP
#define RT_XS_CAP 96         // per-ray entries for the n1/n2 container walk
#endif
#ifndef RT_AUX_STACK
#define RT_AUX_STACK 32      // per-lane auxiliary-BVH stack entries, kept in LDS (rt_accel.h: 3 per level of the 4-wide tree, kMaxDepth / 2 levels, + 2)
#endif
#ifndef RT_BLOCK
#define RT_BLOCK 256         // threads per block of every kernel that traces (LDS stack stride): 256 or 64
#endif
#define RT_AUX_NONE 0xFFFFFFFFu
#define RT_SHAPE_CULLABLE 0x40u  // device copy of a shape: `prim` holds the index of its cull record
#define RT_NODE_AUX 0x4u      // device copy only: this node is the root of a subtree covered by an auxiliary BVH
#define RT_NODE_HAS_NOSHADOW 0x8u  // device copy only: a shape with shadow() == false lives in this subtree
#define RT_NODE_CULLREC 0x10u     // device copy only (transformed nodes): the idrun slot holds the index of the subtree's cull record
#ifndef RT_CONE_MIN_PAD
#define RT_CONE_MIN_PAD 0.125f  // aux BVH: worst-case pad / longest triangle edge above which the cone bound is evaluated
#endif
#define RT_SCENE_HAS_NOSHADOW 0x1u // Scene::props: some shape has shadow() == false
#define RT_SCENE_SHADOW_TWO_PASS 0x4u // Scene::props: shadow-less shapes exist AND auxiliary BVHs exist: find F first, then any occluder
#define RT_SCENE_SHAPE_CULL 0x10u   // Scene::props: some shapes have cull records (k_render_jobs<., true>)
#define RT_SCENE_AREA_LIGHT 0x8u   // Scene::props: some area light takes more than one sample
#define RT_SCENE_HAS_PATTERN 0x80u  // Scene::props: some material has a pattern
#define RT_SCENE_OTHER_LIGHT 0x100u // Scene::props: some light is not a point light
#define RT_SCENE_HAS_COPIES 0x200u  // Scene::props: some triangle is listed more than once (end_of_copy_run)
// What a job-loop kernel is compiled with (integrate<.., FEAT>, world_intersect<.., FEAT>): the launch picks the
// instantiation with the fewest features that still covers the scene (rt_hip.hip jobs_kernel)
#define RT_F_CSG 1u      // csg trees (world.rs:45-49)
#define RT_F_COLLECT 2u  // the n1/n2 pass and its lists: some refractive index is not 1
#define RT_F_PATTERN 4u  // pattern evaluation
#define RT_F_AREA 8u     // lights other than point lights
#define RT_F_GROUPS 16u  // the reference walk over groups and the auxiliary trees below them
#define RT_F_ALL 31u
#define RT_F_WAVES4 32u  // not a feature of the scene: with RT_F_GROUPS alone, the instantiation at four waves per SIMD (rt_hip.hip)
#define RT_SCENE_RI_ALL_ONE 0x2u   // Scene::props: every material has refractive_index == 1 (n1 = n2 = 1 always)

namespace rtd {

struct V3 {
  float x, y, z;
};
struct Ray {
  V3 o, d;
};

__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }  // tuples.rs:83-93
__device__ __forceinline__ V3 mulc(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }      // tuples.rs:70-80
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }    // tuples.rs:96-106
// dot / magnitude of vectors: the w terms of the reference (tuples.rs:232-242) are 0*0 and drop out.
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float magnitude(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) { return a / magnitude(a); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {  // tuples.rs:244-250
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 reflect(V3 in, V3 n) { return in - (2.f * dot(in, n)) * n; }  // tuples.rs:252-254

// ordered-float total order (NaN greatest, NaN == NaN)
__device__ __forceinline__ bool of_gt(float a, float b) { return (a > b) || ((a != a) && (b == b)); }
__device__ __forceinline__ bool of_eq(float a, float b) { return (a == b) || ((a != a) && (b != b)); }
__device__ __forceinline__ float of_max(float a, float b) { return of_gt(a, b) ? a : b; }
__device__ __forceinline__ float of_min(float a, float b) { return of_gt(a, b) ? b : a; }

struct Counters {
  uint32_t primary, shadow, reflect, refract, overflow;
  // work counters (only maintained by the COUNT instantiations; bench roofline)
  uint32_t node_visits, tri_tests, shape_tests, shaded, xforms, aux_tests;
  uint32_t work_shadow, work_collect;  // COUNT only: tests spent in shadow rays / n1-n2 collect passes
  // COUNT only: the rays the REFERENCE would have cast for the same pixels.  It re-traces every secondary ray once
  // per light (world.rs:66-86) where the device traces it once and weights it, so a ray `k` levels below
  // the primary stands for n_lights^k reference rays.  Equal to primary+shadow+reflect+refract with one light.
  uint32_t ref_rays;
#ifdef RT_PHASE_TIMERS
  uint32_t t_iters, t_load;            // walk iterations, clocks waiting for node records
  uint32_t t_jobs, t_refill, t_draw;   // job-loop iterations, clocks in the refill and in its queue draw
  uint32_t u[8];  // lane / wave step counts: aux node, aux leaf, walk visit, walk shapes
  uint32_t t_world, t_aux, t_shapes;   // diagnostics build: wave clock spent in world_intersect / aux BVH / shape tests
#endif
};
#ifdef RT_PHASE_TIMERS
#define RT_T0(v) const uint64_t v = __builtin_readcyclecounter()
#define RT_TACC(field, v) cn.field += (uint32_t)(__builtin_readcyclecounter() - v)
// lane utilisation of a step: every active lane counts itself, the first active lane counts the wave
#define RT_UTIL(k)                                                                                        \
  do {                                                                                                    \
    cn.u[2 * (k)]++;                                                                                      \
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) cn.u[2 * (k) + 1]++; \
  } while (0)
#else
#define RT_T0(v)
#define RT_TACC(field, v)
#define RT_UTIL(k)
#endif

struct Scene {
  const RtXform* xforms;
  const RtShape* shapes;
  const float4* tri_geom;     // 3 float4 per triangle
  const float4* tri_normals;  // 3 float4 per triangle
  const RtNode* nodes;
  const RtCsgNode* csg_nodes;
  const RtCsgRoot* csg_roots;
  const RtMaterial* materials;
  const RtPattern* patterns;
  const RtUvPattern* uvpatterns;
  const RtImage* images;
  const float* texels;
  const RtLight* lights;
  const float4* aux_nodes;  // library-built auxiliary BVHs (rt_accel.h): 5 float4 per node
  const float4* shape_cull;  // cull spheres of cubes / spheres (rt_accel.h shape_cull_record): 3 float4 per record
  const float4* node_cull;   // cull records of transformed groups (rt_accel.h build_node_cull): 3 float4 per record
  const float4* aux_tris;   // their leaf triangles: 3 float4 per triangle, p1.w = offset in the group
  uint32_t n_shapes, n_nodes, n_csg_roots, n_lights, n_loose, recursion_limit;
  uint32_t flags;  // RT_RENDER_*
  uint32_t props;  // RT_SCENE_* (derived by rt_scene_create)
  unsigned long long* sticky;  // the context's overflow word: launches add to it, synchronising calls read and clear it
  // Side lists (rt_hip.hip grow_side): per-lane room in device memory for the rays whose n1/n2 intersection list passes
  // RT_XS_CAP.  side_cap entries per lane and list (0: none); only the persistent kernels use it, a lane's
  // region is found from its block and thread index.
  char* side;
  uint32_t side_cap;
};

// The scene tables are never written while a kernel runs.  Reading them through the constant address space tells the
// compiler so: a load whose address is wave-uniform then becomes a scalar load (SGPRs, off the vector-memory path)
// even though the persistent kernel stores pixels between traversals, which otherwise makes every earlier-proven
// "not clobbered" load a per-lane vector load (16 VGPRs for one group record).
#define RT_CONST_AS __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const T RT_CONST_AS* cst(const T* p) {
  return (const T RT_CONST_AS*)(uintptr_t)p;
}
typedef float f4n __attribute__((ext_vector_type(4)));     // native vectors: HIP's float4 / uint4 classes cannot be
typedef uint32_t u4n __attribute__((ext_vector_type(4)));  // copied out of another address space
struct M34 {  // rows 0..2 of an inverse transform, fetched as three 16-B loads
  f4n r0, r1, r2;
};
__device__ __forceinline__ M34 load_xform(const Scene& sc, uint32_t xform);
// Matrix<4,4> * Point / Vector (matrices.rs:68-80, :220-236): row sums left to right.
template <class P>
__device__ __forceinline__ V3 xf_point(P m, V3 p) {
  return V3{((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
            ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]};
}
template <class P>
__device__ __forceinline__ V3 xf_vector(P m, V3 v) {
  return V3{(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
            (m[8] * v.x + m[9] * v.y) + m[10] * v.z};
}
// inverse.transpose() * normal (shapes.rs:281-285): column sums of the 3x3 block.
template <class P>
__device__ __forceinline__ V3 xf_normal(P m, V3 n) {
  return V3{(m[0] * n.x + m[4] * n.y) + m[8] * n.z, (m[1] * n.x + m[5] * n.y) + m[9] * n.z,
            (m[2] * n.x + m[6] * n.y) + m[10] * n.z};
}
__device__ __forceinline__ M34 load_xform(const Scene& sc, uint32_t xform) {
  const f4n RT_CONST_AS* q = cst(reinterpret_cast<const f4n*>(sc.xforms + xform));
  return M34{q[0], q[1], q[2]};
}
__device__ __forceinline__ Ray xf_ray(const Scene& sc, uint32_t xform, Ray r) {  // rays.rs:19-24
  const M34 m = load_xform(sc, xform);
  return Ray{V3{((m.r0.x * r.o.x + m.r0.y * r.o.y) + m.r0.z * r.o.z) + m.r0.w,
                ((m.r1.x * r.o.x + m.r1.y * r.o.y) + m.r1.z * r.o.z) + m.r1.w,
                ((m.r2.x * r.o.x + m.r2.y * r.o.y) + m.r2.z * r.o.z) + m.r2.w},
             V3{(m.r0.x * r.d.x + m.r0.y * r.d.y) + m.r0.z * r.d.z, (m.r1.x * r.d.x + m.r1.y * r.d.y) + m.r1.z * r.d.z,
                (m.r2.x * r.d.x + m.r2.y * r.d.y) + m.r2.z * r.d.z}};
}

// Cube::check_axis (shapes/cube.rs:67-85)
__device__ __forceinline__ void check_axis(float origin, float direction, float mn, float mx, float& tmin,
                                           float& tmax) {
  float a = mn - origin, b = mx - origin;
  if (__builtin_fabsf(direction) >= 1e-4f) {
    a = a / direction;
    b = b / direction;
  } else {
    a = a * __builtin_inff();
    b = b * __builtin_inff();
  }
  if (a > b) {
    tmin = b;
    tmax = a;
  } else {
    tmin = a;
    tmax = b;
  }
}
// BoundingBox::intersects (bounds.rs:91-123) / Cube::local_intersect (cube.rs:87-114)
__device__ __forceinline__ bool slab(const Ray& r, V3 mn, V3 mx, float& tmin, float& tmax) {
  float x0, x1, y0, y1, z0, z1;
  check_axis(r.o.x, r.d.x, mn.x, mx.x, x0, x1);
  check_axis(r.o.y, r.d.y, mn.y, mx.y, y0, y1);
  check_axis(r.o.z, r.d.z, mn.z, mx.z, z0, z1);
  tmin = of_max(of_max(x0, y0), z0);
  tmax = of_min(of_min(x1, y1), z1);
  return !of_gt(tmin, tmax);
}

// The same decision (and t-range, to a few ulp) without the six IEEE divisions of check_axis: quotients through
// v_rcp_f32 are within 4 ulp of the correctly rounded ones, so when the approximate entry and exit values are
// further apart than that error allows, the reference's comparison tmin > tmax is already decided.  Anything
// closer, any axis on the "parallel" branch and anything non-finite takes the exact path above.
__device__ __forceinline__ bool slab_decided(const Ray& r, V3 mn, V3 mx, float& tmin, float& tmax) {
  const float ax = __builtin_fabsf(r.d.x), ay = __builtin_fabsf(r.d.y), az = __builtin_fabsf(r.d.z);
  if (__builtin_fminf(__builtin_fminf(ax, ay), az) >= 1e-4f) {  // false for NaN
    float ix = __builtin_amdgcn_rcpf(r.d.x), iy = __builtin_amdgcn_rcpf(r.d.y), iz = __builtin_amdgcn_rcpf(r.d.z);
    float x0 = (mn.x - r.o.x) * ix, x1 = (mx.x - r.o.x) * ix;
    float y0 = (mn.y - r.o.y) * iy, y1 = (mx.y - r.o.y) * iy;
    float z0 = (mn.z - r.o.z) * iz, z1 = (mx.z - r.o.z) * iz;
    float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    float hi = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)),
                                                __builtin_fmaxf(__builtin_fabsf(y0), __builtin_fabsf(y1))),
                                __builtin_fmaxf(__builtin_fabsf(z0), __builtin_fabsf(z1)));
    float e = 1e-6f * big;  // 4 ulp of the largest quotient, twice over
    if (lo - hi > e) return false;  // certain miss (tmin, tmax unused)
    if (hi - lo > e) {              // certain hit
      tmin = lo;
      tmax = hi;
      return true;
    }
  }
  return slab(r, mn, mx, tmin, tmax);
}

// Local intersections: fill ts[] in the reference's insertion order, return the count.
__device__ __forceinline__ int isect_sphere(const Ray& r, float* ts) {  // sphere.rs:64-80
  float a = dot(r.d, r.d);
  float b = 2.f * dot(r.d, r.o);
  float c = dot(r.o, r.o) - 1.f;
  float disc = b * b - 4.f * a * c;
  if (disc < 0.f) return 0;
  float sq = __builtin_sqrtf(disc);
  ts[0] = (-b - sq) / (2.f * a);
  ts[1] = (-b + sq) / (2.f * a);
  return 2;
}
__device__ __forceinline__ int isect_plane(const Ray& r, float* ts) {  // plane.rs:66-73
  if (__builtin_fabsf(r.d.y) < 1e-4f) return 0;
  ts[0] = -r.o.y / r.d.y;
  return 1;
}
__device__ __forceinline__ int isect_cube(const Ray& r, float* ts) {  // cube.rs:87-114
  float tmin, tmax;
  if (!slab(r, v3(-1.f, -1.f, -1.f), v3(1.f, 1.f, 1.f), tmin, tmax)) return 0;
  ts[0] = tmin;
  ts[1] = tmax;
  return 2;
}
__device__ __forceinline__ bool cap_cyl(const Ray& r, float t) {  // cylinder.rs:89-93
  float x = r.o.x + t * r.d.x, z = r.o.z + t * r.d.z;
  return (x * x + z * z) <= 1.f;
}
__device__ __forceinline__ bool cap_cone(const Ray& r, float t, float y) {  // cone.rs:90-94
  float x = r.o.x + t * r.d.x, z = r.o.z + t * r.d.z;
  return (x * x + z * z) <= __builtin_fabsf(y);
}
// cylinder.rs:95-156 — two Option slots; caps fill the first free slot; output in slot order.
__device__ __forceinline__ int isect_cylinder(const Ray& r, float mn, float mx, bool closed, float* ts) {
  float s0 = 0.f, s1 = 0.f;
  bool h0 = false, h1 = false;
  float a = r.d.x * r.d.x + r.d.z * r.d.z;
  if (!(__builtin_fabsf(a) < 1e-4f)) {
    float b = 2.f * r.o.x * r.d.x + 2.f * r.o.z * r.d.z;
    float c = r.o.x * r.o.x + r.o.z * r.o.z - 1.f;
    float disc = b * b - 4.f * a * c;
    if (disc < 0.f) return 0;
    float sq = __builtin_sqrtf(disc);
    float t0 = (-b - sq) / (2.f * a), t1 = (-b + sq) / (2.f * a);
    if (t0 > t1) {
      float tmp = t0;
      t0 = t1;
      t1 = tmp;
    }
    float y0 = r.o.y + t0 * r.d.y;
    if (mn < y0 && y0 < mx) {
      h0 = true;
      s0 = t0;
    }
    float y1 = r.o.y + t1 * r.d.y;
    if (mn < y1 && y1 < mx) {
      h1 = true;
      s1 = t1;
    }
  }
  if (closed && !(__builtin_fabsf(r.d.y) < 1e-4f)) {
    float t = (mn - r.o.y) / r.d.y;
    if (cap_cyl(r, t)) {
      if (!h0) {
        h0 = true;
        s0 = t;
      } else if (!h1) {
        h1 = true;
        s1 = t;
      }
    }
    t = (mx - r.o.y) / r.d.y;
    if (cap_cyl(r, t)) {
      if (!h0) {
        h0 = true;
        s0 = t;
      } else if (!h1) {
        h1 = true;
        s1 = t;
      }
    }
  }
  int n = 0;
  if (h0) ts[n++] = s0;
  if (h1) ts[n++] = s1;
  return n;
}
// cone.rs:96-165 — four Option slots.
__device__ __forceinline__ int isect_cone(const Ray& r, float mn, float mx, bool closed, float* ts) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  bool h[4] = {false, false, false, false};
  float a = r.d.x * r.d.x - r.d.y * r.d.y + r.d.z * r.d.z;
  float b = 2.f * r.o.x * r.d.x - 2.f * r.o.y * r.d.y + 2.f * r.o.z * r.d.z;
  float c = r.o.x * r.o.x - r.o.y * r.o.y + r.o.z * r.o.z;
  if (__builtin_fabsf(a) < 1e-4f) {
    if (__builtin_fabsf(b) < 1e-4f) return 0;
    h[0] = true;
    s[0] = -c / (2.f * b);
  } else {
    float disc = b * b - 4.f * a * c;
    if (disc < 0.f) return 0;
    float sq = __builtin_sqrtf(disc);
    float t0 = (-b - sq) / (2.f * a), t1 = (-b + sq) / (2.f * a);
    if (t0 > t1) {
      float tmp = t0;
      t0 = t1;
      t1 = tmp;
    }
    float y0 = r.o.y + t0 * r.d.y;
    if (mn < y0 && y0 < mx) {
      h[0] = true;
      s[0] = t0;
    }
    float y1 = r.o.y + t1 * r.d.y;
    if (mn < y1 && y1 < mx) {
      h[1] = true;
      s[1] = t1;
    }
  }
  if (closed && !(__builtin_fabsf(r.d.y) < 1e-4f)) {
    float t = (mn - r.o.y) / r.d.y;
    if (cap_cone(r, t, mn)) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!h[i]) {
          h[i] = true;
          s[i] = t;
          break;
        }
    }
    t = (mx - r.o.y) / r.d.y;
    if (cap_cone(r, t, mx)) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (!h[i]) {
          h[i] = true;
          s[i] = t;
          break;
        }
    }
  }
  int n = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (h[i]) ts[n++] = s[i];
  return n;
}
// triangle.rs:93-115 / smooth_triangle.rs:96-118 (Möller–Trumbore)
__device__ __forceinline__ bool isect_triangle(const Ray& r, V3 p1, V3 e1, V3 e2, float& t, float& u, float& v) {
  V3 dir_cross_e2 = cross(r.d, e2);
  float det = dot(e1, dir_cross_e2);
  if (__builtin_fabsf(det) < 1e-4f) return false;
  float f = 1.f / det;
  V3 p1_to_origin = r.o - p1;
  u = f * dot(p1_to_origin, dir_cross_e2);
  if (!(0.f <= u && u <= 1.f)) return false;
  V3 origin_cross_e1 = cross(p1_to_origin, e1);
  v = f * dot(r.d, origin_cross_e1);
  if (v < 0.f || (u + v) > 1.f) return false;
  t = f * dot(e2, origin_cross_e1);
  return true;
}
__device__ __forceinline__ void load_tri(const Scene& sc, uint32_t prim, V3& p1, V3& e1, V3& e2) {
  const f4n RT_CONST_AS* g = cst(reinterpret_cast<const f4n*>(sc.tri_geom)) + 3u * (size_t)prim;
  f4n a = g[0], b = g[1], c = g[2];
  p1 = v3(a.x, a.y, a.z);
  e1 = v3(b.x, b.y, b.z);
  e2 = v3(c.x, c.y, c.z);
}

// ---------------------------------------------------------------------------------------
// Sink: what a traversal does with each intersection it finds.  One type with two modes, so that
// the traversal code exists once in the kernel (instruction-cache footprint, DESIGN.md "Kernel").
//  closest: Intersections::hit() (intersections.rs:94-96) over the stable t-sorted list (SURVEY.md
//           Q1): minimise (t, insertion rank) over t > 0.  Traversal order is therefore free.
//  collect: every intersection with t <= tlimit (total order), for the n1/n2 container walk.
// ---------------------------------------------------------------------------------------
enum { SINK_CLOSEST = 0, SINK_COLLECT = 1, SINK_CLOSEST_NOSHADOW = 2, SINK_ANY_SHADOW = 3 };
#define RT_GLOBAL_AS __attribute__((address_space(1)))
#define RT_XS_OVERFLOWED 0x7FFFFFFF
// Every intersection of a ray up to its hit: input of the n1/n2 walk.  Two homes for it, chosen per KERNEL (template
// parameter SIDE of integrate): RT_XS_CAP entries in per-lane scratch -- every normal launch -- or the lane's side list
// in device memory, as long as the context says (Scene::side_cap), for the launches of a context one of whose rays has
// passed RT_XS_CAP (rt_hip.hip grow_side; the reference's list is unbounded, intersections.rs:66-73).  Two kernels
// rather than one list that continues in device memory: that version's extra branch in Sink::add, never taken, cost the
// collect-heavy cover scene 19 % of its frame.
struct XsList {
  static constexpr bool kBig = false;
  float t[RT_XS_CAP];
  uint32_t s[RT_XS_CAP];
  __device__ __forceinline__ void attach(const Scene&) {}
  __device__ __forceinline__ int capacity() const { return RT_XS_CAP; }
  __device__ __forceinline__ void put(int c, float tt, uint32_t ss) {
    t[c] = tt;
    s[c] = ss;
  }
};
// No list at all: the kernels without RT_F_COLLECT (worlds in which every refractive index is 1 never walk n1/n2 lists)
struct XsNone {
  static constexpr bool kBig = false;
  __device__ __forceinline__ void attach(const Scene&) {}
  __device__ __forceinline__ int capacity() const { return 0; }
  __device__ __forceinline__ void put(int, float, uint32_t) {}
};
// A lane's side lists: [xs t | xs s | containers], side_cap entries each.
enum { SIDE_XS_T = 0, SIDE_XS_S = 1, SIDE_CONTAINERS = 2, SIDE_LISTS = 3 };
__device__ __forceinline__ char* side_list(const Scene& sc, int which) {
  const size_t lane = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  return sc.side + (lane * SIDE_LISTS + (size_t)which) * sc.side_cap * 4u;
}
struct XsSideList {
  static constexpr bool kBig = true;  // the kernels that keep the lists here are also the ones with the BIG capacities (Lim)
  // (pointers into the GLOBAL address space, declared as such: a store through a generic pointer may alias the
  // per-lane arrays and the LDS state as far as the compiler knows)
  RT_GLOBAL_AS float* t;
  RT_GLOBAL_AS uint32_t* s;
  RT_GLOBAL_AS uint32_t* containers;
  int cap;
  __device__ __forceinline__ void attach(const Scene& sc) {
    t = (RT_GLOBAL_AS float*)(uintptr_t)side_list(sc, SIDE_XS_T);
    s = (RT_GLOBAL_AS uint32_t*)(uintptr_t)side_list(sc, SIDE_XS_S);
    containers = (RT_GLOBAL_AS uint32_t*)(uintptr_t)side_list(sc, SIDE_CONTAINERS);
    cap = (int)sc.side_cap;
  }
  __device__ __forceinline__ int capacity() const { return cap; }
  __device__ __forceinline__ void put(int c, float tt, uint32_t ss) {
    t[c] = tt;
    s[c] = ss;
  }
};
// Eight registers: the modes never need their private state at the same time, so it shares fields, and the
// per-intersection decisions (takes / wants / add) are written as selects over those fields rather than as
// branches on the mode: lanes of a wave are in different modes, and every branch costs exec-mask bookkeeping in the
// innermost loops.
//   (t, shape)  closest: best hit so far (inf, -1 if none) | any-shadow: F, the nearest hit on a shadow-less shape
//               A candidate counts only if it is positive, in front of `bound`, and before (t, shape) in (t, rank) order.
//   bound       closest: nothing at or beyond it matters (NaN = no bound: t = +inf is a hit the reference shades)
//               any-shadow: distance to the light, -inf once an occluder is found | collect: keep t <= bound
template <class List>
struct SinkT {
  static constexpr bool kBig = List::kBig;
  float t;
  int shape;
  float a, b;    // closest: (u, v) of the best hit
  int c;         // collect: entries stored (RT_XS_OVERFLOWED: the list, side list included, overflowed)
  List* xs;      // collect: where the entries go
  float bound;
  int mode;
  float lim;     // nothing beyond this t can matter any more (kept up to date by add); -inf: any-shadow found its occluder
  __device__ __forceinline__ void init(int mode_) {
    t = __builtin_inff();
    shape = -1;
    a = b = 0.f;
    c = 0;
    xs = nullptr;
    bound = __builtin_nanf("");
    mode = mode_;
    lim = __builtin_inff();
  }
  __device__ __forceinline__ void init_closest() { init(SINK_CLOSEST); }
  __device__ __forceinline__ void set_cap(float cap) {  // closest: nothing at or beyond cap can matter (shadow rays)
    bound = cap;
    lim = __builtin_fminf(t, cap);
  }
  __device__ __forceinline__ void init_collect(List* xs_, float tlimit_) {
    init(SINK_COLLECT);
    xs = xs_;
    bound = tlimit_;
    lim = tlimit_;
  }
  // World::is_shadowed (world.rs:101-111) looks at the closest hit h* only: shadowed <=> t* < distance and
  // h*.shape.shadow().  With F = closest hit among shadow-less shapes, that is equivalent to: some hit x on a
  // shadow-casting shape has 0 < t_x < distance and (t_x, rank_x) < (t_F, rank_F).  F is found first
  // (SINK_CLOSEST_NOSHADOW, only where such shapes exist), then any x ends the search (SINK_ANY_SHADOW).
  __device__ __forceinline__ void init_any_shadow(float distance, float tF_, int sF_) {
    init(SINK_ANY_SHADOW);
    bound = distance;
    t = tF_;
    shape = sF_;
    lim = __builtin_fminf(distance, tF_);
  }
  __device__ __forceinline__ float u() const { return a; }
  __device__ __forceinline__ float v() const { return b; }
  __device__ __forceinline__ bool found() const { return lim == -__builtin_inff(); }
  __device__ __forceinline__ int count() const { return c == RT_XS_OVERFLOWED ? 0 : c; }
  __device__ __forceinline__ int overflow() const { return c == RT_XS_OVERFLOWED ? 1 : 0; }
  // does this traversal look at a shape with these flags at all?
  __device__ __forceinline__ bool takes(uint32_t kind_flags) const {
    return (kind_flags & RT_SHAPE_SHADOW) ? mode != SINK_CLOSEST_NOSHADOW : mode != SINK_ANY_SHADOW;
  }
  __device__ __forceinline__ bool wants(float tt, uint32_t s) const {
    const bool ordered = tt > 0.f && !(tt >= bound) && (shape < 0 || tt < t || (tt == t && (int)s < shape));
    const bool listed = !of_gt(tt, bound);
    return mode == SINK_COLLECT ? listed : ordered;
  }
  __device__ __forceinline__ void add(float tt, uint32_t s, float uu, float vv) {
    if (!wants(tt, s)) return;
    if (mode == SINK_COLLECT) {
      if (c < xs->capacity()) {
        xs->put(c, tt, s);
        c++;
      } else {
        c = RT_XS_OVERFLOWED;
      }
    } else {
      const bool any = mode == SINK_ANY_SHADOW;  // an occluder ends the search; a closest hit replaces the best
      lim = any ? -__builtin_inff() : __builtin_fminf(tt, bound);
      bound = any ? -__builtin_inff() : bound;
      t = any ? t : tt;
      shape = any ? shape : (int)s;
      a = any ? a : uu;
      b = any ? b : vv;
    }
  }
  __device__ __forceinline__ float limit() const { return lim; }
  __device__ __forceinline__ bool wants_negative() const { return mode == SINK_COLLECT; }
  __device__ __forceinline__ bool done() const { return mode == SINK_ANY_SHADOW && found(); }
};
typedef SinkT<XsList> Sink;

// Fused forms for the library's OWN tests (cull records): these are not the reference's arithmetic, nothing has to
// round as rustc's code does, and a fused multiply-add only rounds less.  (The kernels are compiled with
// -ffp-contract=off: nothing else in them is ever fused.)
__device__ __forceinline__ float dot_fused(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross_fused(V3 a, V3 b) {
  return v3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
// The local_intersect of a shape's kind (shapes.rs:208-245) on a ray in the shape's space.
template <class Sink>
__device__ __forceinline__ void shape_test_local(const Scene& sc, uint32_t si, uint32_t kf, uint32_t prim, float mn, float mx,
                                                 const Ray& local, Sink& sink) {
  float ts[4];
  int n = 0;
  switch (kf & RT_SHAPE_KIND_MASK) {
    case RT_SPHERE: n = isect_sphere(local, ts); break;
    case RT_PLANE: n = isect_plane(local, ts); break;
    case RT_CUBE: n = isect_cube(local, ts); break;
    case RT_CYLINDER: n = isect_cylinder(local, mn, mx, (kf & RT_SHAPE_CLOSED) != 0, ts); break;
    case RT_CONE: n = isect_cone(local, mn, mx, (kf & RT_SHAPE_CLOSED) != 0, ts); break;
    default: {
      V3 p1, e1, e2;
      load_tri(sc, prim, p1, e1, e2);
      float t, u, v;
      if (isect_triangle(local, p1, e1, e2, t, u, v)) sink.add(t, si, u, v);
      return;
    }
  }
  for (int i = 0; i < n; i++) sink.add(ts[i], si, 0.f, 0.f);
}
// Shape::intersect (shapes.rs:204-246): inverse-transform the ray, dispatch on kind.
// CULL: the kernel variant for scenes that have cull records (decided per scene: the test costs registers and ~25
// operations per shape, which only pays where a ray meets many shapes in a row)
template <bool CULL = false, class Sink>
__device__ __forceinline__ void shape_intersect(const Scene& sc, uint32_t si, const Ray& ray, Sink& sink) {
  const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.shapes + si));
  u4n w0 = raw[0], w1 = raw[1];
  uint32_t kf = w0.x, xform = w0.y, prim = w1.y;
  if (!sink.takes(kf)) return;
  if (CULL && (kf & RT_SHAPE_CULLABLE) && !(sc.flags & RT_RENDER_NO_PRUNE)) {
    // cull sphere (rt_accel.h): a line this far from the shape is one the reference's own test rejects; skips the
    // transform and the test.  NaN => not culled.
    const f4n RT_CONST_AS* q = cst(reinterpret_cast<const f4n*>(sc.shape_cull)) + 3u * (size_t)prim;
    const f4n c0 = q[0], c1 = q[1], c2 = q[2];
    const V3 oc = v3(c0.x, c0.y, c0.z) - ray.o;
    const V3 cr = cross_fused(oc, ray.d);
    const float ol2 = dot_fused(oc, oc);
    const float dl = __builtin_amdgcn_sqrtf(dot_fused(ray.d, ray.d)), ol = __builtin_amdgcn_sqrtf(ol2);
    const float thr = __builtin_fmaf(dl, __builtin_fmaf(c1.z, ol2, __builtin_fmaf(c1.x, ol, c0.w) + c1.y), __builtin_fmaf(c1.w, ol, c2.x)) * 1.0001f;
    if (dot_fused(cr, cr) > thr * thr) return;
  }
  float mn = __uint_as_float(w1.z), mx = __uint_as_float(w1.w);
  Ray local = xform ? xf_ray(sc, xform, ray) : ray;
  shape_test_local(sc, si, kf, prim, mn, mx, local, sink);
}
// Shape::intersect for a ray whose ORIGIN is already in the shape's space (`local_o` = the shape's inverse transform
// applied to the ray's origin, Ray::transform's first half, rays.rs:19-24): the many rays that leave one point -- the
// shadow rays of an area light's samples (lights.rs:123-134) -- transform it once.  Same expressions on the same values
// as shape_intersect => same bits.
template <class Sink>
__device__ __forceinline__ void shape_intersect_from(const Scene& sc, uint32_t si, const V3 local_o, const V3 dir, Sink& sink) {
  const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.shapes + si));
  u4n w0 = raw[0], w1 = raw[1];
  const uint32_t kf = w0.x, xform = w0.y, prim = w1.y;
  if (!sink.takes(kf)) return;
  const Ray local{local_o, xform ? xf_vector(cst(sc.xforms)[xform].m, dir) : dir};
  shape_test_local(sc, si, kf, prim, __uint_as_float(w1.z), __uint_as_float(w1.w), local, sink);
}

// ---------------------------------------------------------------------------------------
// CSG (csg.rs:26-124), evaluated bottom-up over the post-ordered node array with one
// per-ray entry list; entries are kept sorted by (t total order, insertion rank), which is
// what the reference's nested stable sorts produce (DESIGN.md "CSG").
// ---------------------------------------------------------------------------------------
template <bool BIG>
struct CsgListT {
  // RT_CSG_CAP entries are enough for every tree rt_scene_create accepts: it adds up the most intersections each leaf
  // can have (validate_scene) and rejects a tree that could pass the capacity, so this list cannot overflow at run time
  // (`overflow` stays as the check of that claim).  A spill path into the side lists was tried twice: as a second,
  // inlined evaluation it cost the render kernel 28 more spilled registers and 5 % of the dragons frame; with the
  // arrays behind pointers the direct kernel spilled thousands.
  float t[Lim<BIG>::kCsgCap];
  uint32_t s[Lim<BIG>::kCsgCap];
  int n;
  int overflow;
  __device__ __forceinline__ bool takes(uint32_t) const { return true; }
  __device__ __forceinline__ void add(float tt, uint32_t ss, float, float) {
    if (n < Lim<BIG>::kCsgCap) {
      t[n] = tt;
      s[n] = ss;
      n++;
    } else {
      overflow = 1;
    }
  }
};
__device__ __forceinline__ bool vclass_eq(uint32_t a, uint32_t b) { return a == b && a != RT_VCLASS_NEVER; }
__device__ __forceinline__ bool entry_less(float ta, uint32_t sa, float tb, uint32_t sb) {
  return of_gt(tb, ta) || (of_eq(ta, tb) && sa < sb);
}
// Drop an entry that is == its successor (BTreeMap bulk build).
// (FP / UP: pointers to float / uint32_t in whatever address space the list lives in)
template <class FP, class UP>
__device__ inline int dedup_sorted(const Scene& sc, FP t, UP s, int lo, int hi) {
  int w = lo;
  for (int i = lo; i < hi; i++) {
    if (i + 1 < hi && of_eq(t[i], t[i + 1]) && vclass_eq(cst(sc.shapes)[s[i]].vclass, cst(sc.shapes)[s[i + 1]].vclass)) continue;
    t[w] = t[i];
    s[w] = s[i];
    w++;
  }
  return w;
}
template <class FP, class UP>
__device__ inline void sort_entries(FP t, UP s, int lo, int hi) {
  for (int i = lo + 1; i < hi; i++) {
    float ti = t[i];
    uint32_t si = s[i];
    int j = i - 1;
    while (j >= lo && entry_less(ti, si, t[j], s[j])) {
      t[j + 1] = t[j];
      s[j + 1] = s[j];
      j--;
    }
    t[j + 1] = ti;
    s[j + 1] = si;
  }
}
__device__ __forceinline__ bool csg_allowed(uint32_t op, bool l_hit, bool in_l, bool in_r) {  // csg.rs:116-124
  if (op == RT_CSG_UNION) return (l_hit && !in_r) || (!l_hit && !in_l);
  if (op == RT_CSG_INTERSECT) return (l_hit && in_r) || (!l_hit && in_l);
  return (l_hit && !in_r) || (!l_hit && in_l);
}
// Evaluates one top-level csg; on return L.t/L.s[0..L.n) is Csg::local_intersect(ray).
template <bool BIG>
__device__ inline void csg_eval(const Scene& sc, RtCsgRoot root, const Ray& ray, CsgListT<BIG>& L) {
  int seg[Lim<BIG>::kCsgDepth];
  int sp = 0;
  L.n = 0;
  for (uint32_t k = root.first; k < root.first + root.count; k++) {
    const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.csg_nodes + k));
    u4n w0 = raw[0], w1 = raw[1], w2 = raw[2];
    uint32_t op = w0.w, flags = w1.w;
    int nseg = ((flags & RT_CSG_LEFT_IS_SHAPE) ? 0 : 1) + ((flags & RT_CSG_RIGHT_IS_SHAPE) ? 0 : 1);
    int start = nseg ? seg[sp - nseg] : L.n;
    sp -= nseg;
    float tmin, tmax;
    V3 bmin = v3(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z));
    V3 bmax = v3(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z));
    if (!slab(ray, bmin, bmax, tmin, tmax)) {  // csg.rs:27-29
      L.n = start;
      seg[sp++] = start;
      continue;
    }
    if (flags & RT_CSG_LEFT_IS_SHAPE) shape_intersect(sc, w2.x, ray, L);
    if (flags & RT_CSG_RIGHT_IS_SHAPE) shape_intersect(sc, w2.y, ray, L);
    sort_entries(&L.t[0], &L.s[0], start, L.n);
    L.n = dedup_sorted(sc, &L.t[0], &L.s[0], start, L.n);
    // filter_intersections (csg.rs:51-72)
    bool in_l = false, in_r = false;
    int w = start;
    uint32_t left_lo = w2.z, left_hi = w2.w;
    for (int i = start; i < L.n; i++) {
      uint32_t s = L.s[i];
      uint32_t vc = cst(sc.shapes)[s].vclass;
      bool l_hit;
      if (flags & RT_CSG_LEFT_IS_SHAPE) {
        l_hit = vclass_eq(cst(sc.shapes)[w2.x].vclass, vc);
      } else {  // Csg::includes (csg.rs:74-85): by value over the left subtree's leaves
        l_hit = false;
        for (uint32_t q = left_lo; q < left_hi && !l_hit; q++) l_hit = vclass_eq(cst(sc.shapes)[q].vclass, vc);
      }
      if (csg_allowed(op, l_hit, in_l, in_r)) {
        L.t[w] = L.t[i];
        L.s[w] = s;
        w++;
      }
      if (l_hit)
        in_l = !in_l;
      else
        in_r = !in_r;
    }
    L.n = dedup_sorted(sc, &L.t[0], &L.s[0], start, w);  // `result.into()` (csg.rs:71)
    seg[sp++] = start;
  }
}

// ---------------------------------------------------------------------------------------
// Auxiliary BVH over a whole subtree of identity-transform triangle groups (rt_accel.h).  Same
// Möller–Trumbore arithmetic on the same triangles as the reference's linear loops
// (groups.rs:118-120); the BVH only skips triangles the test would have rejected, and an accepted
// triangle is kept only if the reference's own box tests would have let the ray reach it.
// ---------------------------------------------------------------------------------------
// ROWS: RT_AUX_STACK, which no traversal can outgrow, or fewer (the kernel for worlds of groups only keeps
// RT_AUX_ROWS_LEAN so that four of its blocks fit a CU): a traversal that would need more gives up and its ray takes
// the reference walk through the subtree instead (aux_coop: key 1).
#ifndef RT_AUX_ROWS_LEAN
#define RT_AUX_ROWS_LEAN 20
#endif
#define RT_AUX_ROWS_TEST 5  // RT_RENDER_SHORT_STACK: so few that the giving up is exercised (tests)
template <int ROWS>
__device__ __forceinline__ uint32_t* aux_stack() {
  __shared__ uint32_t stk[ROWS * RT_BLOCK];
  return stk + threadIdx.x;
}
// Conservative slab test against a box widened by `pad`: never reports a miss for a line that comes
// within the padding of the box contents; NaNs (0 * inf) count as "inside".
//   o_lo = o + pad meets the box minima, o_hi = o - pad the maxima: (mn - pad) - o taken as mn - (o + pad), six
//   additions a node instead of six a box (either way two roundings of coordinate size before the product: what the
//   relative margin below and the static padding of rt_accel.h are there for).
//   Margin: a miss is reported only where tn - 2e-6|tn| > tf + 2e-6|tf|, tn - 2e-6|tn| > hi or tf + 2e-6|tf| < lo
//   would hold as well: tn against tf + 5e-6|tf|, and against the limits widened by 5e-6 of themselves (aux_node4, once
//   a node).  fmax drops the NaN of -inf + inf (a ray parallel to a slab it lies outside of has tn = +inf or
//   tf = -inf and must stay a miss, not turn into "NaN = inside").
__device__ __forceinline__ bool aux_box(V3 o_lo, V3 o_hi, V3 inv, f4n mn, f4n mx, float lo_w, float hi_w, float& tnear) {
  float ax = (mn.x - o_lo.x) * inv.x, bx = (mx.x - o_hi.x) * inv.x;
  float ay = (mn.y - o_lo.y) * inv.y, by = (mx.y - o_hi.y) * inv.y;
  float az = (mn.z - o_lo.z) * inv.z, bz = (mx.z - o_hi.z) * inv.z;
  float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
  float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
  const float tfw = __builtin_fmaxf(tf, __builtin_fmaf(5e-6f, __builtin_fabsf(tf), tf));
  tnear = tn;
  return !(tn > tfw) && !(tn > hi_w) && !(tf < lo_w);
}
// One visit of a 4-wide node (rt_accel.h Layout): the boxes of its 2-4 children, widened for this ray by the node's
// normal-cone bound (see aux_intersect), against the t range [lo0, lim].  Returns the nearest child that is hit and
// pushes the others onto the lane's stack, farthest first; RT_AUX_NONE when none is hit.
//   K = |e1||e2||d| / |det| <= kmax for every accepted hit; the cone gives the usually far smaller bound
//   K <= 1 / (sin(e1,e2) * |cos(d, normal)|): |cos(d, normal)| >= cos(phi + w) for every triangle below, phi = angle(d, axis)
// rows: how many entries the stack may hold (wave-uniform); `full` is set, and nothing pushed, when they do not suffice
template <bool COUNT>
__device__ __forceinline__ uint32_t aux_node4(const Scene& sc, const uint32_t idx, const V3 o, const V3 inv, const V3 dn,
                                              const float kmax, const float padk, const float inv_dm, const bool use_cone,
                                              const float lo0, const float lim, uint32_t* stk, int& sp, Counters& cn,
                                              const int rows, bool& full) {
  const f4n RT_CONST_AS* N = cst(reinterpret_cast<const f4n*>(sc.aux_nodes)) + 8u * (size_t)idx;
  const f4n q0 = N[0], q1 = N[1], q2 = N[2], q3 = N[3], q4 = N[4], q5 = N[5], q6 = N[6], q7 = N[7];
  RT_UTIL(0);
  uint32_t r0 = __float_as_uint(q0.w), r1 = __float_as_uint(q1.w), r2 = __float_as_uint(q2.w), r3 = __float_as_uint(q3.w);
  if (COUNT) cn.aux_tests += 2u + (r2 != RT_AUX_NONE) + (r3 != RT_AUX_NONE);
  float kb = kmax;
  if (use_cone && q7.w >= -1.f) {
    const float ik2 = q4.w * q4.w + q5.w * q5.w + q6.w * q6.w;               // (1 / max(1/sin(e1,e2)))^2
    const float cp = __builtin_fabsf(dn.x * q4.w + dn.y * q5.w + dn.z * q6.w);  // cos(phi) / kalpha
    const float sw = __builtin_amdgcn_sqrtf(__builtin_fmaxf(0.f, 1.f - q7.w * q7.w));
    const float q = cp * q7.w - __builtin_amdgcn_sqrtf(__builtin_fmaxf(0.f, ik2 - cp * cp)) * sw;  // cos(phi+w) / kalpha
    // rounding (worst where ik2 - cp^2 cancels) moves q by < 4e-4 / kalpha: under 4 % of a q that passes
    if (q > 0.f && q * q > 1e-4f * ik2) kb = __builtin_fminf(kmax, __builtin_amdgcn_rcpf(q) * 1.06f);
  }
  const float pad = padk * kb;
  const float tpad = pad * inv_dm;
  const float lo = lo0 - tpad;
  const float hi = (lim + 1e-5f * __builtin_fabsf(lim)) + tpad;
  float t0, t1, t2, t3;
  const V3 o_lo = v3(o.x + pad, o.y + pad, o.z + pad), o_hi = v3(o.x - pad, o.y - pad, o.z - pad);
  const float lo_w = __builtin_fmaf(-5e-6f, __builtin_fabsf(lo), lo), hi_w = __builtin_fmaf(5e-6f, __builtin_fabsf(hi), hi);
  const bool h0 = aux_box(o_lo, o_hi, inv, q0, q1, lo_w, hi_w, t0);
  const bool h1 = aux_box(o_lo, o_hi, inv, q2, q3, lo_w, hi_w, t1);
  const bool h2 = aux_box(o_lo, o_hi, inv, q4, q5, lo_w, hi_w, t2) && r2 != RT_AUX_NONE;
  const bool h3 = aux_box(o_lo, o_hi, inv, q6, q7, lo_w, hi_w, t3) && r3 != RT_AUX_NONE;
  const float inf = __builtin_inff();
  if (!h0) r0 = RT_AUX_NONE, t0 = inf;
  if (!h1) r1 = RT_AUX_NONE, t1 = inf;
  if (!h2) r2 = RT_AUX_NONE, t2 = inf;
  if (!h3) r3 = RT_AUX_NONE, t3 = inf;
  // nearest first (the order only decides how soon the search narrows): a five-exchange network
#define RT_CSWAP(ta, ra, tb, rb)        \
  {                                     \
    const bool sw_ = tb < ta;           \
    const float tt_ = sw_ ? tb : ta;    \
    const uint32_t rr_ = sw_ ? rb : ra; \
    tb = sw_ ? ta : tb;                 \
    rb = sw_ ? ra : rb;                 \
    ta = tt_;                           \
    ra = rr_;                           \
  }
  RT_CSWAP(t0, r0, t1, r1)
  RT_CSWAP(t2, r2, t3, r3)
  RT_CSWAP(t0, r0, t2, r2)
  RT_CSWAP(t1, r1, t3, r3)
  RT_CSWAP(t1, r1, t2, r2)
#undef RT_CSWAP
  const int n_push = (r1 != RT_AUX_NONE) + (r2 != RT_AUX_NONE) + (r3 != RT_AUX_NONE);
  if (sp + n_push <= rows) {
    if (r3 != RT_AUX_NONE) stk[(sp++) * RT_BLOCK] = r3;
    if (r2 != RT_AUX_NONE) stk[(sp++) * RT_BLOCK] = r2;
    if (r1 != RT_AUX_NONE) stk[(sp++) * RT_BLOCK] = r1;
  } else {
    full = true;
  }
  return r0;
}
// Would the reference have reached the shapes of group `g` from the subtree root `root`?  The box test
// of g, then of every group on the way up whose test is not implied by nesting (rt_accel.h gate_next).
template <bool COUNT>
__device__ __forceinline__ bool aux_gate(const Scene& sc, uint32_t g, uint32_t root, const Ray& cur, Counters& cn) {
  while (g != root && g != RT_AUX_NONE) {
    const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.nodes + g));
    u4n w0 = raw[0], w1 = raw[1], w3 = raw[3];
    if (COUNT) cn.node_visits++;
    float tmin, tmax;
    V3 bmin = v3(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z));
    V3 bmax = v3(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z));
    if (!slab(cur, bmin, bmax, tmin, tmax)) return false;  // groups.rs:111-113
    g = w3.w;
  }
  return true;
}
// false: this ray must take the reference walk through the subtree (nothing has been added to the sink)
template <bool COUNT, class Sink>
__device__ inline bool aux_intersect(const Scene& sc, uint32_t root_node, uint32_t aux_root, const Ray& cur, Sink& sink,
                                     Counters& cn) {
  // the nesting argument behind aux_gate needs one axis on the non-parallel branch of check_axis
  if (!(__builtin_fabsf(cur.d.x) >= 1e-4f || __builtin_fabsf(cur.d.y) >= 1e-4f || __builtin_fabsf(cur.d.z) >= 1e-4f))
    return false;
  const f4n RT_CONST_AS* A = cst(reinterpret_cast<const f4n*>(sc.aux_nodes)) + 8u * (size_t)aux_root;
  f4n h0 = A[0], h1 = A[1];
  // Ray-dependent widening (DESIGN.md "Auxiliary BVH"): in f32 the reference's triangle test accepts a
  // line that passes the triangle's plane up to 16u * K * (|s| + L) outside it (u = 2^-24,
  // K = |e1||e2||d| / |det| <= 1e4 * |e1||e2||d| because |det| >= 1e-4 for every accepted hit, |s| = distance
  // origin -> p1).  With S = farthest distance from the origin to the tree's triangles, a box widened by
  // 1.5 * that bound can only be missed by lines the triangle test rejects.
  V3 oc = cur.o - v3(h0.x, h0.y, h0.z);
  float S = magnitude(oc) + h0.w;
  float dm = magnitude(cur.d);
  // pad = 1.5 * 16u * K * (S + L): K <= kmax from the |det| >= 1e-4 rule of the triangle test; per node the normal
  // cone gives the usually far smaller bound K <= 1 / (sin(e1,e2) * |cos(d, normal)|)
  const float kmax = 1e4f * h1.y * dm;
  const float padk = 1.43e-6f * (S + h1.x);  // 1.5 * 16 * 2^-24 * (S + L)
  // NaN/inf rays take the reference walk.  A worst-case pad wider than the tree is fine: it only applies below
  // nodes without a cone bound, and the walk through them is still no longer than the reference's.
  if (!(padk * kmax < __builtin_inff())) return false;
  const V3 dn = cur.d / dm;
  const float inv_dm = __builtin_amdgcn_rcpf(dm) * 1.000001f;  // t slack = pad / |d|, rounded up
  // the cone bound costs a fifth load and ~25 operations per node: only worth it when the worst-case pad is not
  // already small against the tree's triangles
  const bool use_cone = padk * kmax > RT_CONE_MIN_PAD * h1.x;
  V3 inv = v3(__builtin_amdgcn_rcpf(cur.d.x), __builtin_amdgcn_rcpf(cur.d.y), __builtin_amdgcn_rcpf(cur.d.z));
  uint32_t* stk = aux_stack<RT_AUX_STACK>();
  int sp = 0;
  // entry record: {centre, radius}{lmax, amax, top node, first shape of this instance}; the tree may be shared
  // between instances, so its leaves hold shape and group indices relative to the instance
  const uint32_t shape_base = __float_as_uint(h1.w);
  uint32_t ref = __float_as_uint(h1.z) << 4;
  // the computed t of an accepted hit lies within 13u * K * |s| / |d| < 0.55 * pad / |d| of the line's true
  // crossing of the triangle's plane, which is inside the widened box: slack for the t-range pruning
  const float lo0 = sink.wants_negative() ? -__builtin_inff() : 0.f;
  while (true) {
    uint32_t cnt = ref & 15u, idx = ref >> 4;  // reference = index << 4 | leaf triangle count (0: inner node)
    if (cnt == 0u) {
      bool full = false;
      const uint32_t next = aux_node4<COUNT>(sc, idx, cur.o, inv, dn, kmax, padk, inv_dm, use_cone, lo0, sink.limit(), stk, sp, cn,
                                             RT_AUX_STACK, full);
      if (full) cn.overflow++;  // (RT_AUX_STACK rows hold every traversal of a tree of rt_accel.h's depth)
      if (next != RT_AUX_NONE) {
        ref = next;
        continue;
      }
    } else {
      if (COUNT) cn.tri_tests += cnt;
      RT_UTIL(1);
      const f4n RT_CONST_AS* T = cst(reinterpret_cast<const f4n*>(sc.aux_tris)) + 3u * (size_t)idx;
      for (uint32_t k = 0; k < cnt; k++) {
        f4n a = T[3 * k], b = T[3 * k + 1], c = T[3 * k + 2];
        float t, u, v;
        if (sink.takes(__float_as_uint(c.w)) &&
            isect_triangle(cur, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), t, u, v)) {
          uint32_t shape = shape_base + __float_as_uint(a.w);
          if (sink.wants(t, shape) && aux_gate<COUNT>(sc, root_node + __float_as_uint(b.w), root_node, cur, cn))
            sink.add(t, shape, u, v);
        }
      }
    }
    if (sp == 0 || sink.done()) break;
    ref = stk[(--sp) * RT_BLOCK];
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// The same traversal run by the WHOLE WAVE (closest / shadow searches; the collect pass keeps aux_intersect).
// A wave's rays need trees of very different lengths: most finish in a few steps or need no tree at all, a few
// (grazing rays) take a hundred, and with one ray per lane the wave issues every instruction of the long ones for
// 64 lanes (measured: 30 % of the lanes active in node steps).  Here every lane of the wave takes part in the loop,
// and lanes without work of their own STEAL: a busy lane hands the deferred subtree at the BOTTOM of its stack (the
// largest, farthest one) to an idle lane, which traverses it for the owner's ray.  Candidates from all helpers meet
// in one 64-bit word per owner in LDS, (t bits << 32 | shape): for t > 0 its unsigned order is the reference's
// (t, insertion rank) order, so ds_min_u64 is Intersections::hit(), and every helper prunes against
// the owner's current best.  Stolen subtrees are farther than what the victim keeps, so part of that work is
// speculative -- it runs on lanes that would otherwise idle through the same instructions.
// Results are independent of who traverses what: the set of accepted candidates is the same, min is commutative.
// The winner's (u, v) are not carried: the shading step recomputes them for smooth triangles (same arithmetic).
// ---------------------------------------------------------------------------------------
#ifndef RT_STEAL_MIN_IDLE
#define RT_STEAL_MIN_IDLE 8   // lanes without work before the wave looks for work to hand out (4-wide tree: 8 and 6-triangle
                             // leaves measured best, dragons 4K 2.44 -> 2.39 ms; 16 and 9 were best on the binary tree)
#endif
#ifndef RT_STEAL_PERIOD
#define RT_STEAL_PERIOD 1     // ... every this many steps
#endif
#ifndef RT_STEAL_MAX_PAIRS
#define RT_STEAL_MAX_PAIRS 16
#endif
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 40      // lanes waiting at a leaf that start a leaf step while others could still descend
#endif
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ unsigned long long* aux_keys() {
  __shared__ unsigned long long keys[RT_BLOCK];
  return keys;
}
__device__ __forceinline__ unsigned long long hit_key(float t, int shape) {
  return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)shape;
}
__device__ __forceinline__ float pull_f(int lane, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ uint32_t pull_u(int lane, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)v);
}
// What a lane traverses, and for whom: the ray in tree space, its pruning bounds, the search mode and the owner lane.
struct CoopTask {
  Ray T;
  float Tpadk, Tbound;
  int Tmode;
  uint32_t Towner;
  // derived from the ray (see aux_intersect for the bounds behind the numbers); recomputed when the task changes hands
  float Tkmax, inv_dm;
  V3 dn, inv;
  bool use_cone;
  __device__ __forceinline__ void derive(float h1x, float h1y) {
    const float dm = magnitude(T.d);
    Tkmax = 1e4f * h1y * dm;  // (the owner's expression on the owner's values: the same bits wherever it is computed)
    inv_dm = __builtin_amdgcn_rcpf(dm) * 1.000001f;
    dn = inv_dm * T.d;  // feeds the cone bound only (6 % slack there): no need for three IEEE divisions
    use_cone = Tpadk * Tkmax > RT_CONE_MIN_PAD * h1x;
    inv = v3(__builtin_amdgcn_rcpf(T.d.x), __builtin_amdgcn_rcpf(T.d.y), __builtin_amdgcn_rcpf(T.d.z));
  }
};
// One step of a lane's traversal: a node (both child boxes) or a leaf (its triangles), then the next reference from
// the lane's stack rows [bot, sp), or `work` = false.
template <bool COUNT>
__device__ __forceinline__ void coop_step(const Scene& sc, Counters& cn, const CoopTask& K, const uint32_t shape_base,
                                          const uint32_t root_node, unsigned long long* keys, uint32_t* stk, uint32_t& ref,
                                          int& sp, const int bot, bool& work, const int rows) {
  const unsigned long long kcur = __hip_atomic_load(&keys[K.Towner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // nothing at or beyond lim can matter: the owner's best so far, capped by its bound (shadow: the light);
  // key 0 = a shadow search that has its occluder, key 1 = the search was given up (a stack ran full, see aux_coop)
  const float lim = kcur <= 1ull ? -__builtin_inff() : __builtin_fminf(__uint_as_float((uint32_t)(kcur >> 32)), K.Tbound);
  uint32_t next = RT_AUX_NONE;
  const uint32_t cnt = ref & 15u, idx = ref >> 4;
  if (kcur <= 1ull) {
    sp = bot;  // drop what is left
  } else if (cnt == 0u) {
    // closest / shadow searches never want t <= 0
    bool full = false;
    next = aux_node4<COUNT>(sc, idx, K.T.o, K.inv, K.dn, K.Tkmax, K.Tpadk, K.inv_dm, K.use_cone, 0.f, lim, stk, sp, cn, rows, full);
    if (full) {
      // children that found no room are children nobody visits: the owner's search is void (smaller than every key of a
      // hit, so it stays; only an occluder already found, key 0, stands: that answer does not need the rest)
      __hip_atomic_fetch_min(&keys[K.Towner], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      next = RT_AUX_NONE;
      sp = bot;
    }
  } else {
    if (COUNT) cn.tri_tests += cnt;
    RT_UTIL(1);
    const f4n RT_CONST_AS* Tr = cst(reinterpret_cast<const f4n*>(sc.aux_tris)) + 3u * (size_t)idx;
    for (uint32_t k = 0; k < cnt; k++) {
      f4n a = Tr[3 * k], b = Tr[3 * k + 1], c = Tr[3 * k + 2];
      const uint32_t kf = __float_as_uint(c.w);
      const bool takes = (kf & RT_SHAPE_SHADOW) ? K.Tmode != SINK_CLOSEST_NOSHADOW : K.Tmode != SINK_ANY_SHADOW;
      float t, u, v;
      if (takes && isect_triangle(K.T, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), t, u, v)) {
        const uint32_t shape = shape_base + __float_as_uint(a.w);
        const unsigned long long kc = hit_key(t, (int)shape);
        const unsigned long long kn = __hip_atomic_load(&keys[K.Towner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // Sink::wants: positive, in front of the bound, before the best so far in (t, rank) order
        if (t > 0.f && !(t >= K.Tbound) && kc < kn && aux_gate<COUNT>(sc, root_node + __float_as_uint(b.w), root_node, K.T, cn))
          __hip_atomic_fetch_min(&keys[K.Towner], K.Tmode == SINK_ANY_SHADOW ? 0ull : kc, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (next == RT_AUX_NONE) {
    if (sp > bot)
      next = stk[(--sp) * RT_BLOCK];
    else
      work = false;
  }
  ref = next;
}
// Which lanes step now: lanes that have reached a leaf wait while others still descend, until enough of them are
// there.  A leaf step is a serial loop over up to RT_LEAF_TRIS triangles, and run for a few lanes at a time (every
// iteration, as lanes happen to arrive) it dominated the cost (16 % of the lanes active in it).
__device__ __forceinline__ bool coop_steps_now(bool work, uint32_t ref) {
  const bool at_leaf = work && (ref & 15u) != 0u;
  const unsigned long long leaf_lanes = __builtin_amdgcn_ballot_w64(at_leaf);
  const bool leaf_phase = (uint32_t)__builtin_popcountll(leaf_lanes) >= RT_LEAF_BATCH ||
                          leaf_lanes == __builtin_amdgcn_ballot_w64(work);
  return work && at_leaf == leaf_phase;
}
// Hand deferred subtrees to lanes of the wave without work.  Pairing: the r-th idle lane takes from the r-th donor.
// Ranks come from the ballot masks (mbcnt); the donors' lane numbers are compacted to the low lanes with one
// ds_permute (every lane pushes its number: donors to their rank, the others to the remaining slots from the top, a
// bijection), and a thief pulls its victim's number from the lane of its own rank.  (A serial loop over pairs cost ~4
// vector instructions per pair.)  Returns true for a lane that has just taken a task.
__device__ __forceinline__ bool coop_steal(const unsigned long long busy, const uint32_t n_idle, CoopTask& K, uint32_t* stk,
                                           uint32_t& ref, int& sp, int& bot, bool& work) {
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long donors = __builtin_amdgcn_ballot_w64(work && sp > bot);
  if (donors == 0ull) return false;
  const unsigned long long idle = ~busy;
  const uint32_t n_donors = (uint32_t)__builtin_popcountll(donors);
  const uint32_t rd = __builtin_amdgcn_mbcnt_hi((uint32_t)(donors >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)donors, 0u));
  const uint32_t ri = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
  const bool is_donor = ((donors >> lane) & 1ull) != 0ull;
  const uint32_t dest = is_donor ? rd : 63u - (lane - rd);
  const uint32_t compact = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)lane);
  const uint32_t donor_of_rank = pull_u((int)(ri & 63u), compact);
  const bool steals = !work && ri < n_donors;
  const int victim = steals ? (int)donor_of_rank : -1;  // thief: the lane it takes from
  const bool gives = is_donor && rd < n_idle;           // donor: its bottom entry goes to a thief
  uint32_t entry = 0u;
  if (gives) {
    entry = stk[bot * RT_BLOCK];
    bot++;
  }
  // thieves pull the task from their victim's registers (every lane executes the pulls; a lane without a victim
  // reads its own)
  const int src = victim >= 0 ? victim : (int)lane;
  const uint32_t e = pull_u(src, entry);
  const float ox = pull_f(src, K.T.o.x), oy = pull_f(src, K.T.o.y), oz = pull_f(src, K.T.o.z);
  const float dx = pull_f(src, K.T.d.x), dy = pull_f(src, K.T.d.y), dz = pull_f(src, K.T.d.z);
  const float pk = pull_f(src, K.Tpadk), bd = pull_f(src, K.Tbound);
  const uint32_t om = pull_u(src, K.Towner | ((uint32_t)K.Tmode << 16));  // owner (< RT_BLOCK) and mode in one word
  if (victim >= 0) {
    K.T = Ray{v3(ox, oy, oz), v3(dx, dy, dz)};
    K.Tpadk = pk;
    K.Tbound = bd;
    K.Towner = om & 0xFFFFu;
    K.Tmode = (int)(om >> 16);
    ref = e;
    sp = bot = 0;
    work = true;
    return true;
  }
  return false;
}

// want: this lane's ray reached the subtree root (wave-uniform root_node / aux_root).  handled (for lanes that want):
// false = this ray must take the reference walk through the subtree.  Every lane of the wave must call.
template <bool COUNT, int ROWS, class Sink>
__device__ inline void aux_coop(const Scene& sc, const uint32_t root_node, const uint32_t aux_root, const Ray& cur, Sink& sink,
                                Counters& cn, const bool want, bool& handled) {
  const f4n RT_CONST_AS* A = cst(reinterpret_cast<const f4n*>(sc.aux_nodes)) + 8u * (size_t)aux_root;
  const f4n h0 = A[0], h1 = A[1];
  // ---- the task of this lane's own ray ----
  CoopTask K;
  K.T = cur;
  bool work;
  {
    const bool axis_ok = __builtin_fabsf(cur.d.x) >= 1e-4f || __builtin_fabsf(cur.d.y) >= 1e-4f || __builtin_fabsf(cur.d.z) >= 1e-4f;
    const V3 oc = cur.o - v3(h0.x, h0.y, h0.z);
    const float S = magnitude(oc) + h0.w;
    K.Tpadk = 1.43e-6f * (S + h1.x);
    K.derive(h1.x, h1.y);
    work = want && axis_ok && (K.Tpadk * K.Tkmax < __builtin_inff());
  }
  handled = work;
  K.Tbound = sink.bound;
  K.Tmode = sink.mode;
  K.Towner = threadIdx.x;
  const uint32_t shape_base = __float_as_uint(h1.w);
  unsigned long long* keys = aux_keys();
  const unsigned long long key0 = hit_key(sink.t, sink.shape);
  if (work) __hip_atomic_store(&keys[K.Towner], key0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t* stk = aux_stack<ROWS>();
  const int rows = (ROWS > RT_AUX_ROWS_TEST && (sc.flags & RT_RENDER_SHORT_STACK)) ? RT_AUX_ROWS_TEST : ROWS;
  int sp = 0, bot = 0;  // deferred subtrees of this lane: rows [bot, sp) of its column
  uint32_t ref = __float_as_uint(h1.z) << 4;
  uint32_t iter = 0;
  while (true) {
    // one kind of step per iteration (coop_steps_now)
    if (coop_steps_now(work, ref)) coop_step<COUNT>(sc, cn, K, shape_base, root_node, keys, stk, ref, sp, bot, work, rows);
    const unsigned long long busy = __builtin_amdgcn_ballot_w64(work);
    if (busy == 0ull) break;
    // ---- hand deferred subtrees to lanes without work ----
    iter++;
    const uint32_t n_idle = 64u - (uint32_t)__builtin_popcountll(busy);
    if (n_idle >= RT_STEAL_MIN_IDLE && (iter % RT_STEAL_PERIOD) == 0u) {
      if (coop_steal(busy, n_idle, K, stk, ref, sp, bot, work)) K.derive(h1.x, h1.y);
    }
  }
  // ---- the owner takes its result ----
  if (handled) {
    const unsigned long long k = __hip_atomic_load(&keys[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k == 1ull) {
      handled = false;  // some lane's stack ran full on this ray: nothing is taken from the tree, the reference walk decides
    } else if (sink.mode == SINK_ANY_SHADOW) {
      if (k == 0ull) {
        sink.lim = -__builtin_inff();
        sink.bound = -__builtin_inff();
      }
    } else if (k != key0) {
      sink.t = __uint_as_float((uint32_t)(k >> 32));
      sink.shape = (int)(uint32_t)k;
      sink.a = sink.b = __builtin_nanf("");  // (u, v): recomputed by the shading step
      sink.lim = __builtin_fminf(sink.t, sink.bound);
    }
  }
}

// ---------------------------------------------------------------------------------------
// World::intersect (world.rs:44-62) as one traversal feeding a sink.
// Groups: stackless depth-first walk over the DFS-ordered node array (skip pointers); the ray is
// re-expressed in a group's space when the group has a non-identity transform (groups.rs:127-133).
// No per-lane stack of saved rays: when the walk leaves a transformed subtree, the ray of the next
// node's space is rebuilt from the world ray through that node's chain of transformed ancestors
// (same matrices in the same order => same bits).
// ---------------------------------------------------------------------------------------
template <bool BIG = false>
__device__ inline Ray ray_in_space_of(const Scene& sc, int xf_node, const Ray& world) {
  // xf_node: innermost transformed group whose space is wanted (-1 = world)
  int chain[Lim<BIG>::kXfDepth];
  int n = 0;
  int q = xf_node;
  while (q >= 0 && n < Lim<BIG>::kXfDepth) {
    chain[n++] = q;
    int par = cst(sc.nodes)[q].parent;
    q = par >= 0 ? cst(sc.nodes)[par].next_xf : -1;
  }
  Ray r = world;
  for (int k = n - 1; k >= 0; k--) r = xf_ray(sc, cst(sc.nodes)[chain[k]].xform, r);
  return r;
}
// Which of several value-identical shapes a hit is booked on.  A mesh that lists a triangle twice gives two entries
// equal in (t, Shape, u, v); the reference's Intersections is a BTreeMap built in bulk from the list sorted by t
// (stable: insertion order within one t), and that build drops an entry that is == its SUCCESSOR.  So of
// the entries at the hit's t, in insertion order, the hit is the last of the first run of equal neighbours: the sink's
// winner `w` (lowest rank at that t) gives way to the next entry at that t if that is an equal shape with equal (u, v).
// rt_scene_create links every shape to the next of its value class (a slot its kind does not use).  A copy is
// first intersected by itself, with the ray of its own group: one that is not hit at that very t (nearly every copy
// under ANOTHER transform: the instances of a mesh) is no entry there and is passed over.  One that is -- every copy
// under the same transformed ancestor, and the odd one moved within its own plane -- is walked to, from where the walk
// stands, in the order the reference gathers entries (groups depth-first, shapes before children, groups.rs:111-123,
// with the reference's own box tests and rays; then the loose shapes, world.rs:56-59): every shape passed on the way
// is intersected, and one with that t ends the run, as does a copy whose group the ray does not reach (no entry) or
// whose (u, v) differ.  The csgs of the world come first in that order (world.rs:45-49): the entries of a csg are what
// csg_eval leaves of its leaves' intersections, so a leaf is an entry only if the evaluated list holds it.  After
// RT_COPY_GAP_MAX intersections the run ends where it stands.  Not part of the render kernels: a pass of its own over
// the ids (rt_hip.hip k_copies_pixels / k_copies_rays), only for scenes with copies and callers that ask for ids.
#ifndef RT_COPY_GAP_MAX
#define RT_COPY_GAP_MAX 65536
#endif
struct SameT {
  float t0;
  bool found;
  __device__ __forceinline__ bool takes(uint32_t) const { return true; }
  __device__ __forceinline__ void add(float tt, uint32_t, float, float) { found = found || of_eq(tt, t0); }
};
// (the link sits in a slot of the record that the shape's kind does not use: the min of everything but cylinders and
// cones, whose slot is the triangle index)
__device__ inline int next_copy(const Scene& sc, int s) {
  const u4n w0 = cst(reinterpret_cast<const u4n*>(sc.shapes + s))[0], w1 = cst(reinterpret_cast<const u4n*>(sc.shapes + s))[1];
  const uint32_t kind = w0.x & RT_SHAPE_KIND_MASK;
  return (int)(kind == RT_CYLINDER || kind == RT_CONE ? w1.y : w1.z);
}
// A shape's own entries: the nearest in front of the origin (the hit, when the shape is the one hit) ...
struct OwnHit {
  float t, u, v;
  bool found;
  __device__ __forceinline__ bool takes(uint32_t) const { return true; }
  __device__ __forceinline__ void add(float tt, uint32_t, float uu, float vv) {
    if (tt > 0.f && (!found || tt < t)) found = true, t = tt, u = uu, v = vv;  // (t = +inf is a hit the reference shades)
  }
};
// ... and whether one of them is at t0 (with its (u, v))
struct EntryAt {
  float t0, u, v;
  bool found;
  __device__ __forceinline__ bool takes(uint32_t) const { return true; }
  __device__ __forceinline__ void add(float tt, uint32_t, float uu, float vv) {
    if (!found && of_eq(tt, t0)) found = true, u = uu, v = vv;
  }
};
enum { RUN_BROKEN = 0, RUN_REACHED = 1, RUN_UNREACHED = 2 };
// Where a walk stands in the reference's order of entries.
struct OrderWalk {
  bool csg;                        // the walk is among the csgs
  uint32_t root;                   // ... at this one
  uint32_t i;                      // the node the walk is at
  uint32_t k;                      // inside a node: its next own shape; among the loose shapes: the next shape
  uint32_t first_shape, n_shapes;  // of node i, when inside
  bool inside;                     // node i was reached: its shapes from k on come next
  bool loose;                      // the groups are done
  int at;                          // every shape up to this index is behind the walk
  Ray cur;                         // the ray in the space of the node the walk is at
  uint32_t leave_at;
  // Walks on to shape `target` (> at); every shape passed on the way is intersected: RUN_BROKEN = one of them has t0.
  __device__ inline int advance(const Scene& sc, const Ray& world, int target, float t0, int& budget) {
    if (csg) {
      CsgListT<true> L;
      L.overflow = 0;
      for (; root < sc.n_csg_roots; root++) {
        if ((budget -= 8) <= 0) return RUN_BROKEN;
        csg_eval<true>(sc, sc.csg_roots[root], world, L);
        int first = 0x7FFFFFFF;  // the next entry at t0 in this csg's list
        for (int e = 0; e < L.n; e++)
          if (of_eq(L.t[e], t0) && (int)L.s[e] > at && (int)L.s[e] < first) first = (int)L.s[e];
        if (first == 0x7FFFFFFF) continue;
        if (first > target) return RUN_UNREACHED;  // (the leaves of a csg are a range of their own: the target is not in its list)
        at = first;
        return first == target ? RUN_REACHED : RUN_BROKEN;
      }
      csg = false;
    }
    for (;;) {
      if (!loose && !inside) {
        if (i >= sc.n_nodes) {
          loose = true;
          const int first_loose = (int)(sc.n_shapes - sc.n_loose);
          k = (uint32_t)(first_loose > at + 1 ? first_loose : at + 1);
          continue;
        }
        const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.nodes + i));
        const u4n n0 = raw[0], n1 = raw[1], n2 = raw[2];
        if ((int)n2.x > target) return RUN_UNREACHED;  // the target lies behind: in a subtree the ray does not enter
        const uint32_t xform = n0.w, skip = n2.z;
        if (i >= leave_at) {
          const int par = (int)n2.w;
          const int q = par >= 0 ? cst(sc.nodes)[par].next_xf : -1;
          cur = q >= 0 ? ray_in_space_of<true>(sc, q, world) : world;
          leave_at = q >= 0 ? cst(sc.nodes)[q].skip : 0xFFFFFFFFu;
        }
        if (xform) {
          cur = xf_ray(sc, xform, cur);
          leave_at = skip;
        }
        float tmin, tmax;
        if (!slab_decided(cur, v3(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z)),
                          v3(__uint_as_float(n1.x), __uint_as_float(n1.y), __uint_as_float(n1.z)), tmin, tmax)) {  // groups.rs:111-113
          i = skip;
          continue;
        }
        inside = true;
        first_shape = n2.x;
        n_shapes = n2.y;
        k = 0;
      }
      const uint32_t end = loose ? sc.n_shapes : n_shapes;
      while (k < end) {
        const int idx = (int)(loose ? k : first_shape + k);
        if (idx > target) return RUN_UNREACHED;
        k++;
        if (idx <= at) continue;
        at = idx;
        if (idx == target) return RUN_REACHED;
        if (budget-- <= 0) return RUN_BROKEN;
        SameT between{t0, false};
        shape_intersect<false>(sc, (uint32_t)idx, loose ? world : cur, between);
        if (between.found) return RUN_BROKEN;
      }
      if (loose) return RUN_UNREACHED;
      inside = false;
      i = i + 1u;
    }
  }
};
__device__ inline int end_of_copy_run(const Scene& sc, const Ray& world, int w) {
  int next = next_copy(sc, w);
  if (next == w) return w;
  const int first_loose = (int)(sc.n_shapes - sc.n_loose);
  const u4n w0 = cst(reinterpret_cast<const u4n*>(sc.shapes + w))[0];
  const int parent = (int)w0.w;
  const bool smooth = (w0.x & RT_SHAPE_KIND_MASK) == RT_SMOOTH_TRIANGLE;
  // the ray in the space of w's group, through the same chain of transforms as the walk (same bits), and the hit's t
  const int anchor = parent >= 0 ? cst(sc.nodes)[parent].next_xf : -1;
  const Ray local = anchor >= 0 ? ray_in_space_of<true>(sc, anchor, world) : world;
  OwnHit own{0.f, 0.f, 0.f, false};
  shape_intersect<false>(sc, (uint32_t)w, local, own);
  if (!own.found) return w;
  const float t0 = own.t, u0 = own.u, v0 = own.v;
  OrderWalk walk;
  walk.at = w;
  walk.cur = local;
  walk.leave_at = anchor >= 0 ? cst(sc.nodes)[anchor].skip : 0xFFFFFFFFu;
  walk.csg = parent < 0 && w < first_loose;  // the leaf of a csg
  walk.root = 0u;
  walk.loose = parent < 0 && w >= first_loose;
  walk.inside = parent >= 0;  // w's own group: the ray got there, and `local` is the ray it was given
  walk.i = parent >= 0 ? (uint32_t)parent : 0u;
  walk.first_shape = parent >= 0 ? cst(sc.nodes)[parent].first_shape : 0u;
  walk.n_shapes = parent >= 0 ? cst(sc.nodes)[parent].n_shapes : 0u;
  walk.k = walk.loose ? (uint32_t)(w + 1) : parent >= 0 ? (uint32_t)w + 1u - walk.first_shape : 0u;
  int budget = RT_COPY_GAP_MAX;
  while (next != w) {
    const int c = next;
    next = next_copy(sc, c);
    const bool last = next == c;
    // is the copy hit at t0 at all, by the ray of its own group?
    const int c_parent = cst(sc.shapes)[c].parent;
    const int c_anchor = c_parent >= 0 ? cst(sc.nodes)[c_parent].next_xf : -1;
    const Ray rc = c_anchor == anchor ? local : c_anchor >= 0 ? ray_in_space_of<true>(sc, c_anchor, world) : world;
    if (budget-- <= 0) break;
    EntryAt ec{t0, 0.f, 0.f, false};
    shape_intersect<false>(sc, (uint32_t)c, rc, ec);
    bool entry = ec.found;
    if (entry) {
      const int r = walk.advance(sc, world, c, t0, budget);
      if (r == RUN_BROKEN) break;
      entry = r == RUN_REACHED;
    }
    if (entry) {
      if (smooth && !(ec.u == u0 && ec.v == v0)) break;  // an entry at t0 that is not equal: the run ends before it
      w = c;
    }
    if (last) break;
  }
  return w;
}

// The walk is WAVE-SYNCHRONOUS: every lane walks the same DFS order and lanes differ only in the
// subtrees they skip, so the node index is kept uniform across the wave (all node, transform, shape and
// triangle records of the reference tree are then fetched with scalar loads, off the vector-memory
// path that the divergent auxiliary-BVH fetches need), and a lane takes part in a visit only when the
// node is the one it is waiting for.  After node i, a visiting lane waits for i+1 or for skip(i); lanes
// that sat the visit out wait for something >= skip(i); so the next uniform index is i+1 if any lane
// descends and skip(i) otherwise.
#ifndef RT_WI_INLINE
#define RT_WI_INLINE inline
#endif
template <bool COUNT, bool CULL, uint32_t FEAT = RT_F_ALL, class Sink>
__device__ RT_WI_INLINE void world_intersect(const Scene& sc, const Ray& ray, Sink& sink, Counters& cn,
                                             const bool active = true) {
  // `active` false: this lane has no ray to trace but still takes part in the wave-wide votes of the walk
  // csgs first (world.rs:45-49)
  if ((FEAT & RT_F_CSG) && sc.n_csg_roots && active) {
    CsgListT<Sink::kBig> L;
    L.overflow = 0;
    for (uint32_t c = 0; c < sc.n_csg_roots; c++) {
      csg_eval(sc, sc.csg_roots[c], ray, L);
      for (int i = 0; i < L.n; i++)
        if (sink.takes(cst(sc.shapes)[L.s[i]].kind_flags)) sink.add(L.t[i], L.s[i], __builtin_nanf(""), __builtin_nanf(""));
    }
    cn.overflow += L.overflow;
  }
  // groups (world.rs:51-54)
  if constexpr ((FEAT & RT_F_GROUPS) != 0u) {
    Ray cur = ray;
    uint32_t leave_at = 0xFFFFFFFFu;  // first node outside the innermost transformed subtree this lane is in
    const bool prune = !(sc.flags & RT_RENDER_NO_PRUNE);
    uint32_t my_i = active ? 0u : 0xFFFFFFFFu;  // the node this lane visits next
    uint32_t i = 0;                   // wave-uniform
    while (i < sc.n_nodes) {
      i = __builtin_amdgcn_readfirstlane(i);
      const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.nodes + i));
      RT_T0(tl);
      u4n w0 = raw[0], w1 = raw[1], w2 = raw[2], w3 = raw[3];
#ifdef RT_PHASE_TIMERS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RT_TACC(t_load, tl);
      cn.t_iters++;
#endif
      const uint32_t xform = w0.w, flags = w1.w, skip = w2.z;
      bool descend = false;
      if (sink.done()) my_i = 0xFFFFFFFFu;  // an occluder is enough (any-shadow): this lane is finished
      if (my_i == i && sink.mode == SINK_CLOSEST_NOSHADOW && !(flags & RT_NODE_HAS_NOSHADOW)) {
        my_i = skip;  // nothing shadow-less below this group
      }
      if (my_i == i) {
        if (i >= leave_at) {
          int par = (int)w2.w;
          int q = par >= 0 ? cst(sc.nodes)[par].next_xf : -1;
          cur = q >= 0 ? ray_in_space_of<Sink::kBig>(sc, q, ray) : ray;
          leave_at = q >= 0 ? cst(sc.nodes)[q].skip : 0xFFFFFFFFu;
        }
        RT_UTIL(2);
        bool culled = false;
        if ((flags & RT_NODE_CULLREC) && prune) {
          // cull record of the whole subtree, in the space `cur` is in (rt_accel.h build_node_cull): a line this far
          // from everything below the group is one that none of their f32 tests accepts.  Skips the transform and the
          // reference's box test; NaN => not culled.
          const f4n RT_CONST_AS* q = cst(reinterpret_cast<const f4n*>(sc.node_cull)) + 3u * (size_t)w3.z;
          const f4n c0 = q[0], c1 = q[1], c2 = q[2];
          const V3 oc = v3(c0.x, c0.y, c0.z) - cur.o;
          const V3 cr = cross_fused(oc, cur.d);
          const float ol2 = dot_fused(oc, oc);
          const float dl = __builtin_amdgcn_sqrtf(dot_fused(cur.d, cur.d)), ol = __builtin_amdgcn_sqrtf(ol2);
          const float inner = __builtin_fmaf(dl, __builtin_fmaf(c2.y, ol, c2.z), __builtin_fmaf(c1.z, ol2, __builtin_fmaf(c1.x, ol, c0.w)));
          const float thr = __builtin_fmaf(dl, inner, __builtin_fmaf(c1.w, ol, c2.x)) * 1.0001f;
          culled = dot_fused(cr, cr) > thr * thr;
        }
        if (!culled) {
          if (COUNT) cn.node_visits++;
          if (COUNT && xform) cn.xforms++;
          if (xform) {
            cur = xf_ray(sc, xform, cur);
            leave_at = skip;
          }
          float tmin, tmax;
          V3 bmin = v3(__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z));
          V3 bmax = v3(__uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z));
          descend = slab_decided(cur, bmin, bmax, tmin, tmax);  // groups.rs:111-113
          if (descend && prune && (flags & RT_NODE_PRUNABLE)) {
            // Distance pruning (not in the reference): a box that truly encloses its subtree cannot
            // hold a hit beyond its exit or before its entry; margins cover f32 rounding of t.
            float lim = sink.limit();
            if (tmin > lim + (1e-4f * __builtin_fabsf(lim) + 1e-4f)) descend = false;
            if (!sink.wants_negative() && tmax < -(1e-4f * __builtin_fabsf(tmax) + 1e-4f)) descend = false;
          }
        }
      }
      // device copy of a node: the slot of n_children holds the auxiliary BVH of the subtree (RT_NODE_AUX)
      if ((flags & RT_NODE_AUX) && !(sc.flags & RT_RENDER_NO_ACCEL)) {  // wave-uniform
        RT_T0(ta);
        bool want = descend;
        if ((FEAT & RT_F_COLLECT) && want && sink.mode == SINK_COLLECT) {  // the n1/n2 pass keeps a per-lane list: its own traversal
          if (aux_intersect<COUNT>(sc, i, w3.w, cur, sink, cn)) descend = false;
          want = false;
        }
        if (wave_any(want)) {  // closest / shadow searches: the whole wave traverses, idle lanes help (aux_coop)
          bool handled = false;
          aux_coop<COUNT, FEAT == (RT_F_GROUPS | RT_F_WAVES4) ? RT_AUX_ROWS_LEAN : RT_AUX_STACK>(sc, i, w3.w, cur, sink, cn, want, handled);
          if (handled) descend = false;
        }
        RT_TACC(t_aux, ta);
      }
      if (wave_any(descend)) {
        RT_T0(ts);
        const uint32_t first_shape = w2.x, n_shapes = w2.y;
        if (flags & RT_NODE_ALL_TRIS) {
          if (COUNT && descend) cn.tri_tests += n_shapes;
          for (uint32_t k = 0; k < n_shapes; k++) {
            V3 p1, e1, e2;
            load_tri(sc, w3.x + k, p1, e1, e2);
            float t, u, v;
            if (descend && sink.takes(cst(sc.shapes)[first_shape + k].kind_flags) && isect_triangle(cur, p1, e1, e2, t, u, v))
              sink.add(t, first_shape + k, u, v);
          }
        } else {
          if (COUNT && descend) cn.shape_tests += n_shapes;
          for (uint32_t k = 0; k < n_shapes; k++)
            if (descend) {
              RT_UTIL(3);
              shape_intersect<CULL>(sc, first_shape + k, cur, sink);
            }
        }
        RT_TACC(t_shapes, ts);
        if (my_i == i) my_i = descend ? i + 1u : skip;
        i = i + 1u;
      } else {
        if (my_i == i) my_i = skip;
        i = skip;
      }
      if (!wave_any(my_i != 0xFFFFFFFFu)) break;  // every lane of the wave is finished
    }
  }
  // loose shapes (world.rs:56-59): the last n_loose entries of shapes[]
  if (COUNT && active) cn.shape_tests += sc.n_loose;
  for (uint32_t k = sc.n_shapes - sc.n_loose; k < sc.n_shapes; k++)
    if (active && !sink.done()) shape_intersect<CULL>(sc, k, ray, sink);
}

// ---------------------------------------------------------------------------------------
// Shape::normal_at (shapes.rs:187-202) with the parent-chain walks of world_to_object /
// normal_to_world (shapes.rs:272-292, groups.rs:139-164).  Identity levels are exact no-ops on
// points and idempotent renormalisations on normals (DESIGN.md "Parent chains").
// ---------------------------------------------------------------------------------------
template <bool BIG = false>
__device__ inline V3 world_to_object(const Scene& sc, uint32_t shape_xform, int parent, V3 p) {
  int chain[Lim<BIG>::kXfDepth];
  int n = 0;
  int g = parent;
  while (g >= 0) {
    int q = cst(sc.nodes)[g].next_xf;
    if (q < 0) break;
    if (n < Lim<BIG>::kXfDepth) chain[n++] = q;
    g = cst(sc.nodes)[q].parent;
  }
  for (int i = n - 1; i >= 0; i--) p = xf_point(cst(sc.xforms)[cst(sc.nodes)[chain[i]].xform].m, p);
  if (shape_xform) p = xf_point(cst(sc.xforms)[shape_xform].m, p);
  return p;
}
__device__ __forceinline__ V3 renorm_identity(V3 n, uint32_t levels) {
  // each identity level is `normalize(I^T n)`; once |n| rounds to exactly 1 further levels are no-ops
  for (uint32_t k = 0; k < levels; k++) {
    float m = magnitude(n);
    if (m == 1.f) break;
    n = n / m;
  }
  return n;
}
__device__ inline V3 normal_to_world(const Scene& sc, uint32_t shape_xform, int parent, V3 n) {
  if (shape_xform)
    n = normalize(xf_normal(cst(sc.xforms)[shape_xform].m, n));
  else
    n = normalize(n);
  int g = parent;
  while (g >= 0) {
    const RtNode RT_CONST_AS& nd = cst(sc.nodes)[g];
    n = renorm_identity(n, nd.xform ? 0u : nd.idrun);  // (a transformed node's idrun is 0; the device copy reuses the slot)
    int q = nd.next_xf;
    if (q < 0) break;
    n = normalize(xf_normal(cst(sc.xforms)[cst(sc.nodes)[q].xform].m, n));
    g = cst(sc.nodes)[q].parent;
  }
  return n;
}
__device__ inline V3 local_normal_at(const Scene& sc, uint32_t kf, float mn, float mx, uint32_t prim, V3 lp, float hu,
                                     float hv) {
  switch (kf & RT_SHAPE_KIND_MASK) {
    case RT_SPHERE: return lp;                    // sphere.rs:82-84
    case RT_PLANE: return v3(0.f, 1.f, 0.f);      // plane.rs:75-77
    case RT_CUBE: {                               // cube.rs:116-133
      float ax = __builtin_fabsf(lp.x), ay = __builtin_fabsf(lp.y), az = __builtin_fabsf(lp.z);
      float maxc = of_max(of_max(ax, ay), az);
      if (ax == maxc) return v3(lp.x, 0.f, 0.f);
      if (ay == maxc) return v3(0.f, lp.y, 0.f);
      return v3(0.f, 0.f, lp.z);
    }
    case RT_CYLINDER: {  // cylinder.rs:158-167
      float dist = lp.x * lp.x + lp.z * lp.z;
      if (dist < 1.f && lp.y >= mx - 1e-4f) return v3(0.f, 1.f, 0.f);
      if (dist < 1.f && lp.y <= mn + 1e-4f) return v3(0.f, -1.f, 0.f);
      return v3(lp.x, 0.f, lp.z);
    }
    case RT_CONE: {  // cone.rs:167-180
      float dist = lp.x * lp.x + lp.z * lp.z;
      if (dist < 1.f && lp.y >= mx - 1e-4f) return v3(0.f, 1.f, 0.f);
      if (dist < 1.f && lp.y <= mn + 1e-4f) return v3(0.f, -1.f, 0.f);
      float y = __builtin_sqrtf(lp.x * lp.x + lp.z * lp.z);
      if (lp.y > 0.f) y = -y;
      return v3(lp.x, y, lp.z);
    }
    case RT_TRIANGLE: {  // triangle.rs:89-91
      f4n a = cst(reinterpret_cast<const f4n*>(sc.tri_normals))[3u * (size_t)prim];
      return v3(a.x, a.y, a.z);
    }
    default: {  // smooth_triangle.rs:92-94
      const f4n RT_CONST_AS* g = cst(reinterpret_cast<const f4n*>(sc.tri_normals)) + 3u * (size_t)prim;
      f4n a = g[0], b = g[1], c = g[2];
      V3 n1 = v3(a.x, a.y, a.z), n2 = v3(b.x, b.y, b.z), n3 = v3(c.x, c.y, c.z);
      return (hu * n2 + hv * n3) + (1.f - hu - hv) * n1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Patterns (patterns.rs:79-91, patterns/*.rs)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float rem_euclid1(float x) {
  float r = fmodf(x, 1.f);
  return r < 0.f ? r + 1.f : r;
}
__device__ __forceinline__ uint32_t round_to_u32(float x) {  // `x.round() as u32`
  float r = roundf(x);
  if (!(r > 0.f)) return 0u;
  if (r >= 4294967296.f) return 0xFFFFFFFFu;
  return (uint32_t)r;
}
__device__ inline V3 uv_pattern_at(const Scene& sc, int id, float u, float v) {
  const RtUvPattern& p = sc.uvpatterns[id];
  if (p.kind == RT_UV_CHECKER) {  // uv_pattern.rs:44-52
    float u2 = floorf(u * p.width), v2 = floorf(v * p.height);
    int k = (fmodf(u2 + v2, 2.f) == 0.f) ? 0 : 1;
    return v3(p.c[k][0], p.c[k][1], p.c[k][2]);
  }
  if (p.kind == RT_UV_ALIGN_CHECK) {  // uv_pattern.rs:89-104
    int k = 0;
    if (v > 0.8f) {
      if (u < 0.2f)
        k = 1;
      else if (u > 0.8f)
        k = 2;
    } else if (v < 0.2f) {
      if (u < 0.2f)
        k = 3;
      else if (u > 0.8f)
        k = 4;
    }
    return v3(p.c[k][0], p.c[k][1], p.c[k][2]);
  }
  // image, uv_pattern.rs:122-132
  const RtImage& im = sc.images[p.image];
  float vv = 1.f - v;
  float x = u * ((float)im.width - 1.f), y = vv * ((float)im.height - 1.f);
  uint32_t xi = round_to_u32(x), yi = round_to_u32(y);
  if (xi >= im.width) xi = im.width - 1;
  if (yi >= im.height) yi = im.height - 1;
  const float* px = sc.texels + im.offset + ((size_t)yi * im.width + xi) * 3u;
  return v3(px[0], px[1], px[2]);
}
__device__ inline V3 pattern_at(const Scene& sc, const RtPattern& p, V3 pp) {
  const float PI_F = 3.14159265358979323846f;
  V3 a = v3(p.a[0], p.a[1], p.a[2]), b = v3(p.b[0], p.b[1], p.b[2]);
  switch (p.kind) {
    case RT_PAT_STRIPED: return fmodf(floorf(pp.x), 2.f) == 0.f ? a : b;  // striped.rs:28-34
    case RT_PAT_GRADIENT: {                                               // gradient.rs:28-32
      V3 distance = b - a;
      float fraction = pp.x - floorf(pp.x);
      return a + fraction * distance;
    }
    case RT_PAT_RING:  // ring.rs:28-39
      return fmodf(floorf(__builtin_sqrtf(pp.x * pp.x + pp.z * pp.z)), 2.f) == 0.f ? a : b;
    case RT_PAT_CHECKER:  // checker.rs:28-36
      return fmodf(floorf(pp.x) + floorf(pp.y) + floorf(pp.z), 2.f) == 0.f ? a : b;
    case RT_PAT_XYZ_RGB: return pp;  // xyz_rgb.rs:24-26
    case RT_PAT_TEXTURE_MAP: {       // texture_map.rs:31-34, :59-114
      float u, v;
      if (p.uv_mapping == RT_UVMAP_SPHERICAL) {
        float theta = atan2f(pp.x, pp.z);
        float radius = magnitude(pp);
        float phi = acosf(pp.y / radius);
        float raw_u = theta / (2.f * PI_F);
        u = 1.f - (raw_u + 0.5f);
        v = 1.f - phi / PI_F;
      } else if (p.uv_mapping == RT_UVMAP_PLANAR) {
        u = rem_euclid1(pp.x);
        v = rem_euclid1(pp.z);
      } else {
        float theta = atan2f(pp.x, pp.z);
        float raw_u = theta / (2.f * PI_F);
        u = 1.f - (raw_u + 0.5f);
        v = rem_euclid1(pp.y);
      }
      return uv_pattern_at(sc, p.uv[0], u, v);
    }
    default: {  // cube_map.rs:46-138; faces: 0 left,1 front,2 right,3 back,4 up,5 down
      float coord = of_max(of_max(__builtin_fabsf(pp.x), __builtin_fabsf(pp.y)), __builtin_fabsf(pp.z));
      int face;
      float u, v;
      if (coord == pp.x) {
        face = 2;
        u = fmodf(1.f - pp.z, 2.f) / 2.f;
        v = fmodf(pp.y + 1.f, 2.f) / 2.f;
      } else if (coord == -pp.x) {
        face = 0;
        u = fmodf(pp.z + 1.f, 2.f) / 2.f;
        v = fmodf(pp.y + 1.f, 2.f) / 2.f;
      } else if (coord == pp.y) {
        face = 4;
        u = fmodf(pp.x + 1.f, 2.f) / 2.f;
        v = fmodf(1.f - pp.z, 2.f) / 2.f;
      } else if (coord == -pp.y) {
        face = 5;
        u = fmodf(pp.x + 1.f, 2.f) / 2.f;
        v = fmodf(pp.z + 1.f, 2.f) / 2.f;
      } else if (coord == pp.z) {
        face = 1;
        u = fmodf(pp.x + 1.f, 2.f) / 2.f;
        v = fmodf(pp.y + 1.f, 2.f) / 2.f;
      } else {
        face = 3;
        u = fmodf(1.f - pp.x, 2.f) / 2.f;
        v = fmodf(pp.y + 1.f, 2.f) / 2.f;
      }
      return uv_pattern_at(sc, p.uv[face], u, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Area-light jitter (lights.rs:105-121): counter hash shared bit-for-bit with the oracle.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float counter_jitter(uint64_t seed, uint32_t pixel, uint32_t path, uint32_t light,
                                                uint32_t cell, uint32_t site, uint32_t which) {
  uint32_t h = mix32((uint32_t)seed ^ 0x9e3779b9u);
  h = mix32(h ^ (uint32_t)(seed >> 32));
  h = mix32(h ^ pixel);
  h = mix32(h ^ path);
  h = mix32(h ^ light);
  h = mix32(h ^ cell);
  h = mix32(h ^ (site * 2u + which));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// n_lights ^ depth: how many times the reference traces a ray `depth` levels below the primary (Q4)
__device__ __forceinline__ uint32_t ref_multiplicity(uint32_t n_lights, uint32_t depth) {
  uint32_t m = 1u;
  for (uint32_t k = 0; k < depth; k++) m *= n_lights;
  return m;
}
struct RenderParams {
  uint32_t jitter_mode;
  float jitter_value;
  uint64_t seed;
  uint32_t queue_chunk;  // tiles a wave draws from the queue per atomic (integrate), >= 1
};
struct Hit {
  float t, u, v;
  int shape;
};

__device__ __forceinline__ V3 point_on_light(const RtLight RT_CONST_AS& l, uint32_t li, uint32_t u, uint32_t v, uint32_t site,
                                             const RenderParams& rp, uint32_t pixel, uint32_t path) {
  float ju, jv;
  if (rp.jitter_mode == RT_JITTER_COUNTER) {
    uint32_t cell = u * l.vsteps + v;
    ju = counter_jitter(rp.seed, pixel, path, li, cell, site, 0u);
    jv = counter_jitter(rp.seed, pixel, path, li, cell, site, 1u);
  } else {
    ju = jv = rp.jitter_value;
  }
  V3 corner = v3(l.corner[0], l.corner[1], l.corner[2]);
  V3 uvec = v3(l.uvec[0], l.uvec[1], l.uvec[2]), vvec = v3(l.vvec[0], l.vvec[1], l.vvec[2]);
  return (corner + ((float)u + ju) * uvec) + ((float)v + jv) * vvec;
}

// One Phong sample (materials.rs:103-132 / :141-169). Returns false when the light is behind.
__device__ __forceinline__ bool phong_sample(V3 light_pos, V3 pt, V3 eyev, V3 normalv, V3 effective, V3 lcol,
                                             float mdiffuse, float mspecular, float shininess, V3& diffuse,
                                             V3& specular) {
  V3 lightv = normalize(light_pos - pt);
  float light_dot_normal = dot(lightv, normalv);
  if (!(light_dot_normal >= 0.f)) return false;
  diffuse = light_dot_normal * (mdiffuse * effective);
  V3 reflectv = reflect(-lightv, normalv);
  float reflect_dot_eye = dot(reflectv, eyev);
  if (reflect_dot_eye > 0.f) {
    float factor = powf(reflect_dot_eye, shininess);
    specular = factor * (mspecular * lcol);
  } else {
    specular = v3(0.f, 0.f, 0.f);
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// World::color_at (world.rs:91-99) as a per-lane job loop around ONE traversal call site.
// The reference recursion  color_at -> shade_hit -> {is_shadowed per light sample, reflected_color,
// refracted_color -> color_at}  (world.rs:64-148) becomes a state machine: every iteration traces
// one ray (a primary/secondary ray, a shadow ray of the current light sample, or the n1/n2 collect
// pass of a transparent hit) and then advances the shading state.  Lanes of a wave therefore trace
// together whatever kind of ray each is on, and the kernel holds a single copy of the traversal.
// Secondary rays wait on an explicit stack of weighted rays (DESIGN.md "Integrator").
// ---------------------------------------------------------------------------------------
struct PendingRay {
  Ray ray;
  float weight;
  uint32_t remaining;
  uint32_t path;
};
enum { JOB_SHADE = 0, JOB_SHADOW = 1, JOB_COLLECT = 2, JOB_SHADOW_F = 3 };
enum { ST_TRACE = 0, ST_NEXT_LIGHT, ST_LIGHTS_DONE, ST_SECONDARY, ST_NEXT_RAY, ST_DONE };

// The n1/n2 walk of Computations::prepare (intersections.rs:141-160) over the sorted,
// de-duplicated list of every intersection up to the hit.
template <class FP, class UP>
__device__ inline void refractive_indices(const Scene& sc, FP ts, UP ss, UP containers, int count,
                                          float hit_t, int hit_shape, float& n1, float& n2) {
  sort_entries(ts, ss, 0, count);
  int n = dedup_sorted(sc, ts, ss, 0, count);
  n1 = 1.f;
  n2 = 1.f;
  int nc = 0;
  uint32_t hit_vc = cst(sc.shapes)[hit_shape].vclass;
  for (int i = 0; i < n; i++) {
    uint32_t s = ss[i];
    uint32_t vc = cst(sc.shapes)[s].vclass;
    // `x == &intersection`: same t, equal shape value (u,v are functions of ray and shape value)
    bool is_hit = of_eq(ts[i], hit_t) && vclass_eq(vc, hit_vc);
    if (is_hit && nc > 0) n1 = cst(sc.materials)[cst(sc.shapes)[containers[nc - 1]].material].refractive_index;
    int pos = -1;
    for (int k = 0; k < nc; k++)
      if (vclass_eq(cst(sc.shapes)[containers[k]].vclass, vc)) {
        pos = k;
        break;
      }
    if (pos >= 0) {
      for (int k = pos; k + 1 < nc; k++) containers[k] = containers[k + 1];
      nc--;
    } else {
      containers[nc++] = s;
    }
    if (is_hit && nc > 0) n2 = cst(sc.materials)[cst(sc.shapes)[containers[nc - 1]].material].refractive_index;
  }
}
// The n1/n2 walk over what a COLLECT traversal left in `xl` (count entries), in the list's own memory.
__device__ inline void refractive_indices_of(const Scene& sc, XsList& xl, int count, float hit_t, int hit_shape, float& n1,
                                             float& n2) {
  uint32_t containers[RT_XS_CAP];
  refractive_indices(sc, &xl.t[0], &xl.s[0], &containers[0], count, hit_t, hit_shape, n1, n2);
}
__device__ inline void refractive_indices_of(const Scene& sc, XsSideList& xl, int count, float hit_t, int hit_shape, float& n1,
                                             float& n2) {
  refractive_indices(sc, xl.t, xl.s, xl.containers, count, hit_t, hit_shape, n1, n2);
}

// Register budget: the traversal alone needs ~100 VGPRs and the kernel is built for 128 (four waves per SIMD), so the
// shading state must not ride through the traversal in registers.  What prepare() derived from a hit is kept in its
// smallest form -- the shaded ray, t, the normal, the material index -- and eyev, point, over_point, under_point,
// reflectv and the material parameters are recomputed/reloaded where they are used (same expressions => same bits).
// The values only touched between traversals (accumulators, the shaded ray, the pattern colour) are parked in LDS.
enum { PK_RESULT = 0, PK_SRAY_O = 3, PK_SRAY_D = 6, PK_SURFACE = 9, PK_MCOLOR = 12, PK_WEIGHT = 15, PK_SLOTS = 16 };
// accessors take the lane's slot offset (threadIdx.x), so the LDS address space stays visible to the compiler
struct Park {
  uint32_t off;
  __device__ __forceinline__ float* base() const {
    __shared__ float park[PK_SLOTS * RT_BLOCK];
    return park;
  }
};
__device__ __forceinline__ float pk_ld(const Park& pk, int k) { return pk.base()[k * RT_BLOCK + pk.off]; }
__device__ __forceinline__ void pk_st(const Park& pk, int k, float x) { pk.base()[k * RT_BLOCK + pk.off] = x; }
__device__ __forceinline__ V3 pk_ld3(const Park& pk, int k) { return v3(pk_ld(pk, k), pk_ld(pk, k + 1), pk_ld(pk, k + 2)); }
__device__ __forceinline__ void pk_st3(const Park& pk, int k, V3 x) {
  pk_st(pk, k, x.x);
  pk_st(pk, k + 1, x.y);
  pk_st(pk, k + 2, x.z);
}

// Persistent waves with a work queue: a wave draws its next items itself (one atomic per refill) instead of being
// born for one tile, so heavy tiles do not leave CUs idle at the end of a frame and the grid is sized to the device,
// not to the image.  Lanes that finish go idle; a wave takes a whole 8x8 tile (64 consecutive items) when all its lanes
// are done (RT_REFILL_MIN = 64).  Refilling partly idle waves keeps more lanes busy but mixes pixels of different tiles
// in one wave, and the loss of coherence costs more than the idle lanes did -- measured twice: with round 1's
// per-lane traversal (dragons 4K 4.8 / 4.7 / 4.5 / 4.3 ms for a threshold of 1 / 8 / 32 / 64 idle lanes) and with the
// wave-cooperative one (2.87 / 2.80 / 2.80 / 2.70 / 2.51 ms for 8 / 16 / 32 / 48 / 64).
// `Source` supplies the work: total(), start(item, ray) -> false if the item is outside the image, id(item) (the key
// of the item's random numbers), first_hit(item, hit) and finish(item, colour, work).  `queue` is a zeroed counter in device memory.
// Every wave leaves the loop: each refill advances the queue, and a wave whose lanes are all idle either refills
// or, once the queue has run past total(), breaks.
#define RT_QUEUE_SLOT_OFFSET 64  // the counter block starts this many slots before `queue` (RT_QUEUE_SLOT)
#define RT_QUEUE_STRIDE 8  // u64 slots between the queues of two classes (64 B: one atomic unit each)
enum { ST_IDLE = 100 };
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 64u
#endif
template <bool v, class A, class B>
struct Pick {
  typedef A type;
};
template <class A, class B>
struct Pick<false, A, B> {
  typedef B type;
};
// SIDE: the n1/n2 lists live in the context's side lists (see XsList)
// FEAT: what the scene needs of RT_F_* (rt_hip.hip jobs_kernel picks the instantiation; dragons.yaml needs
// RT_F_GROUPS alone, cover.yaml RT_F_COLLECT alone): the code of everything else -- csg evaluation, the n1/n2 collect pass and
// its lists, pattern evaluation with its libm calls, the area-light sample loops -- is not in the kernel at all.  The
// job loop is ~14 000 instructions, more than the instruction cache of a pair of CUs holds.
template <bool COUNT, bool CULL, bool SIDE, uint32_t FEAT = RT_F_ALL, class Source>
__device__ inline void integrate(const Scene& sc, const RenderParams& rp, const Source& src, uint32_t remaining0,
                                 unsigned long long* queue, Counters& cn) {
  PendingRay stack[Lim<SIDE>::kRayStack];
  typedef typename Pick<!(FEAT & RT_F_COLLECT), XsNone, typename Pick<SIDE, XsSideList, XsList>::type>::type List;
  List xl;
  xl.attach(sc);
  Park pk{threadIdx.x};
  int sp = 0;
  bool first = true;
  const float nl = (float)sc.n_lights;
  bool exhausted = false;  // wave-uniform: every queue has run past its last item
  unsigned long long chunk_next = 0, chunk_end = 0;  // wave-uniform: items already drawn from the queue, not yet started
  // HW_REG_XCC_ID (id 20, 4 bits): the XCD this wave runs on
  // (with more than 8 classes: the XCD picks the class modulo 8, the block the rest)
  uint32_t q_class = Source::kClasses > 1u
                         ? (((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) + 8u * ((blockIdx.x >> 3) % ((Source::kClasses + 7u) / 8u))) % Source::kClasses
                         : 0u;
  uint32_t q_tried = 0;
  uint32_t item = 0, work_start = 0;
#ifdef RT_PHASE_TIMERS
  unsigned long long tile_t0 = 0ull, tile_first = 0ull;
#endif

  // the hit being shaded: with the parked ray, enough to rebuild what Computations::prepare holds
  uint32_t remaining = 0u, path = 1u;
  float hit_t = 0.f;
  int hit_shape = -1;
  V3 normalv = v3(0.f, 0.f, 0.f);
  uint32_t mat_id = 0;
  // light loop
  uint32_t li = 0, sample = 0;
  float total = 0.f, distance = 0.f;
  float n1 = 1.f, n2 = 1.f;
  float shadow_tF = __builtin_inff();
  int shadow_sF = -1;

  // the ray to trace next
  Ray tray{v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 1.f)};
  int job = JOB_SHADE;
  int st = ST_IDLE;
  while (true) {
    // ---------------- refill: a wave whose lanes are all idle takes the next tile(s) ----------------
    RT_T0(trf);
#ifdef RT_PHASE_TIMERS
    cn.t_jobs++;
#endif
    // The work is split into Source::kClasses classes, one queue each.  A wave draws from the class of the XCD it runs
    // on (every XCD has its own L2: a class is a fixed set of screen regions, so an XCD keeps meeting the same part of
    // the scene's trees) and moves on to the other classes' queues when its own is empty, so the split costs no balance.
    {
      // Lanes whose pixel is finished take new items once RT_REFILL_MIN of them are idle (64: a wave takes a whole 8x8
      // tile when all its lanes are done).  The wave draws whole tiles from the queue into [chunk_next, chunk_end) and
      // hands the items out to idle lanes from there, so a tile never spreads over two waves.
      const unsigned long long idle_mask = __builtin_amdgcn_ballot_w64(st == ST_IDLE);
      const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_mask);
#ifdef RT_PHASE_TIMERS
      // how long the wave's last draw of items took (100 MHz ticks): histogram over log2 in slots 32.., the longest in 60
      if (n_idle == 64u && tile_t0 != 0ull && (threadIdx.x & 63u) == 0u) {
        const unsigned long long dur = wall_clock64() - tile_t0;
        atomicAdd(queue - RT_QUEUE_SLOT_OFFSET + 32 + (dur ? 63 - __builtin_clzll(dur) : 0), 1ull);
        atomicMax(queue - RT_QUEUE_SLOT_OFFSET + 60, (dur << 32) | (tile_first & 0xFFFFFFFFull));
        tile_t0 = 0ull;
      }
#endif
      if (n_idle >= RT_REFILL_MIN && (!exhausted || chunk_next < chunk_end)) {
        const uint32_t lane = threadIdx.x & 63u;
        // A draw takes rp.queue_chunk tiles with one atomic and the wave works through them: at one atomic per tile the
        // queue alone took 1.7 ms of a 4K frame (130 k tiles), the whole frame time of the simple scenes. Scenes with
        // heavy tiles keep a chunk of 1 (launch_part): there a run of neighbouring tiles in one wave cost more than
        // the atomics (dragons 4K 3.5 -> 4.5 ms).
        RT_T0(tdr);
        if (chunk_next >= chunk_end) {
          unsigned long long base = 0;
          const unsigned long long take = 64ull * rp.queue_chunk;
          while (true) {  // wave-uniform
            const unsigned long long size = src.class_items(q_class);
            if (lane == 0u) base = atomicAdd(queue + RT_QUEUE_STRIDE * q_class, take);
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            if (base < size) {
              chunk_next = base;
              chunk_end = base + take < size ? base + take : size;
              break;
            }
            q_class = q_class + 1u == Source::kClasses ? 0u : q_class + 1u;  // this class is done: help with the next
            if (++q_tried == Source::kClasses) {
              exhausted = true;
              break;
            }
          }
        }
        RT_TACC(t_draw, tdr);
        if (chunk_next < chunk_end) {
          const unsigned long long avail = chunk_end - chunk_next;
          const uint32_t n_take = avail < n_idle ? (uint32_t)avail : n_idle;
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
          const unsigned long long want = chunk_next + rank;
#ifdef RT_PHASE_TIMERS
          tile_t0 = wall_clock64();
          tile_first = src.item_of(q_class, (uint32_t)chunk_next);
#endif
          chunk_next += n_take;
          Ray primary;
          uint32_t it = 0;
          if (st == ST_IDLE && rank < n_take && src.start(q_class, (uint32_t)want, it, primary)) {
            item = it;
            cn.primary++;
            if (COUNT) cn.ref_rays++;
            pk_st3(pk, PK_RESULT, v3(0.f, 0.f, 0.f));
            pk_st3(pk, PK_SRAY_O, primary.o);
            pk_st3(pk, PK_SRAY_D, primary.d);
            pk_st(pk, PK_WEIGHT, 1.f);
            remaining = remaining0;
            path = 1u;
            first = true;
            sp = 0;
            tray = primary;
            job = JOB_SHADE;
            st = ST_TRACE;
            if (COUNT) work_start = cn.node_visits + cn.aux_tests + cn.tri_tests + cn.shape_tests;
          }
        }
      }
      if (!wave_any(st != ST_IDLE)) {
        if (exhausted && chunk_next >= chunk_end) break;
        continue;  // a draw that only held items outside the image: draw again
      }
    }
    RT_TACC(t_refill, trf);
    const bool active = st != ST_IDLE;
    // ---------------- the one traversal ----------------
    SinkT<List> sink;
    if ((FEAT & RT_F_COLLECT) && job == JOB_COLLECT)
      sink.init_collect(&xl, hit_t);
    else if (job == JOB_SHADOW && (sc.props & RT_SCENE_HAS_NOSHADOW) && !(sc.props & RT_SCENE_SHADOW_TWO_PASS)) {
      sink.init_closest();  // small scene with shadow-less shapes: one closest-hit pass, capped at the light
      sink.set_cap(distance);
    } else if (job == JOB_SHADOW)
      sink.init_any_shadow(distance, shadow_tF, shadow_sF);
    else if (job == JOB_SHADOW_F)
      sink.init(SINK_CLOSEST_NOSHADOW);
    else
      sink.init_closest();
    const uint32_t work0 = COUNT ? cn.node_visits + cn.aux_tests + cn.tri_tests + cn.shape_tests : 0u;
    RT_T0(tw);
    world_intersect<COUNT, CULL, FEAT>(sc, tray, sink, cn, active);
    RT_TACC(t_world, tw);
    // parked values are re-read after the traversal, not carried through it in registers: the pointer is laundered
    // so that loads below cannot be forwarded from the stores above (a "memory" clobber would do the same but also
    // turns the walk's scalar loads into vector loads)
    asm volatile("" : "+v"(pk.off));
    if (COUNT) {
      uint32_t dw = cn.node_visits + cn.aux_tests + cn.tri_tests + cn.shape_tests - work0;
      if (job == JOB_SHADOW || job == JOB_SHADOW_F) cn.work_shadow += dw;
      if (job == JOB_COLLECT) cn.work_collect += dw;
    }

    // ---------------- what the traced ray was for ----------------
    if (!active) {
      // idle lane: nothing was traced for it
    } else if (job == JOB_SHADE) {
      Hit hit = Hit{sink.t, sink.u(), sink.v(), sink.shape};
      hit_t = hit.t;
      hit_shape = hit.shape;
      if (hit.shape < 0) {
        if (first) {
          src.first_hit(item, hit);
          first = false;
        }
        st = ST_NEXT_RAY;
      } else {
        if (COUNT) cn.shaded++;
        // ---- Computations::prepare (intersections.rs:121-139) ----
        const Ray sray{pk_ld3(pk, PK_SRAY_O), pk_ld3(pk, PK_SRAY_D)};
        const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.shapes + hit.shape));
        u4n w0 = raw[0], w1 = raw[1];
        uint32_t kf = w0.x, sxf = w0.y, prim = w1.y;
        mat_id = w0.z;
        int parent = (int)w0.w;
        float smn = __uint_as_float(w1.z), smx = __uint_as_float(w1.w);
        if ((FEAT & (RT_F_CSG | RT_F_GROUPS)) && (kf & RT_SHAPE_KIND_MASK) == RT_SMOOTH_TRIANGLE && hit.u != hit.u) {
          // the hit came through a csg list or through the wave-wide tree traversal (aux_coop), which do not carry
          // (u, v): recompute them with the ray in the triangle's space, through the same chain of transforms the
          // walk applied (same matrices in the same order => same bits)
          Ray local = parent >= 0 ? ray_in_space_of<SIDE>(sc, cst(sc.nodes)[parent].next_xf, sray) : sray;
          V3 p1, e1, e2;
          load_tri(sc, prim, p1, e1, e2);
          if (sxf) local = xf_ray(sc, sxf, local);
          float tt;
          isect_triangle(local, p1, e1, e2, tt, hit.u, hit.v);
        }
        if (first) {
          // outputs of the primary hit go out now instead of riding along in registers; (u, v) exist for smooth
          // triangles only (intersections.rs:34-41)
          const bool smooth = (kf & RT_SHAPE_KIND_MASK) == RT_SMOOTH_TRIANGLE;
          src.first_hit(item, Hit{hit.t, smooth ? hit.u : 0.f, smooth ? hit.v : 0.f, hit.shape});  // (copies of a triangle: rt_hip.hip k_copies_*)
          first = false;
        }
        V3 point = sray.o + hit.t * sray.d;
        V3 eyev = -sray.d;
        V3 lp = world_to_object<SIDE>(sc, sxf, parent, point);
        V3 ln = local_normal_at(sc, kf, smn, smx, prim, lp, hit.u, hit.v);
        normalv = normal_to_world(sc, sxf, parent, ln);
        if (dot(normalv, eyev) < 0.f) normalv = -normalv;
        V3 over_point = point + 1e-4f * normalv;
        const RtMaterial RT_CONST_AS& m = cst(sc.materials)[mat_id];
        V3 mcolor;
        if ((FEAT & RT_F_PATTERN) && m.pattern >= 0) {  // pattern_at_shape at over_point (patterns.rs:79-91, world.rs:72)
          const RtPattern& pat = sc.patterns[m.pattern];
          V3 op = world_to_object<SIDE>(sc, sxf, parent, over_point);
          V3 pp = xf_point(cst(sc.xforms)[pat.xform].m, op);
          mcolor = pattern_at(sc, pat, pp);
        } else {
          mcolor = v3(m.color[0], m.color[1], m.color[2]);
        }
        pk_st3(pk, PK_MCOLOR, mcolor);
        // ---- World::shade_hit (world.rs:64-89): direct lighting, summed over lights ----
        pk_st3(pk, PK_SURFACE, v3(0.f, 0.f, 0.f));
        li = 0;
        sample = 0;
        total = 0.f;
        st = ST_NEXT_LIGHT;
      }
    } else if (job == JOB_SHADOW_F) {
      shadow_tF = sink.t;  // closest shadow-less hit (inf, -1 if none); now look for an occluder in front of it
      shadow_sF = sink.shape;
      job = JOB_SHADOW;
      st = ST_TRACE;
    } else if (job == JOB_SHADOW) {
      // World::is_shadowed (world.rs:101-111): closest-hit semantics, see Sink::init_any_shadow
      bool shadowed = sink.mode == SINK_ANY_SHADOW
                          ? sink.found()
                          : (sink.shape >= 0 && sink.t < distance && (cst(sc.shapes)[sink.shape].kind_flags & RT_SHAPE_SHADOW));
      const RtLight RT_CONST_AS& L = cst(sc.lights)[li];
      const RtMaterial RT_CONST_AS& m = cst(sc.materials)[mat_id];
      const float m_ambient = m.ambient, m_diffuse = m.diffuse, m_specular = m.specular, m_shininess = m.shininess;
      const Ray sray{pk_ld3(pk, PK_SRAY_O), pk_ld3(pk, PK_SRAY_D)};
      const V3 eyev = -sray.d;
      const V3 over_point = (sray.o + hit_t * sray.d) + 1e-4f * normalv;
      V3 lcol = v3(L.intensity[0], L.intensity[1], L.intensity[2]);
      V3 effective = mulc(pk_ld3(pk, PK_MCOLOR), lcol);
      V3 ambient = m_ambient * effective;
      V3 surface_sum = pk_ld3(pk, PK_SURFACE);
      if (!(FEAT & RT_F_AREA) || L.kind == RT_LIGHT_POINT) {
        V3 lpos = v3(L.position[0], L.position[1], L.position[2]);
        float intensity = shadowed ? 0.f : 1.f;  // lights.rs:50-56
        V3 diffuse = v3(0.f, 0.f, 0.f), specular = v3(0.f, 0.f, 0.f);
        phong_sample(lpos, over_point, eyev, normalv, effective, lcol, m_diffuse, m_specular, m_shininess, diffuse,
                     specular);
        surface_sum = surface_sum + (ambient + intensity * (diffuse + specular));  // materials.rs:134
        pk_st3(pk, PK_SURFACE, surface_sum);
        li++;
        st = ST_NEXT_LIGHT;
      } else {
        // AreaLight::intensity_at (lights.rs:123-134), one sample per traced ray
        if (!shadowed) total += 1.f;
        sample++;
        uint32_t nbs = L.usteps * L.vsteps;
        if (sample < nbs) {
          st = ST_NEXT_LIGHT;  // next sample of the same light
        } else {
          float nb = (float)nbs;
          float intensity = total / nb;
          // materials.rs:136-175
          V3 sum = v3(0.f, 0.f, 0.f);
          for (uint32_t u = 0; u < L.usteps; u++)
            for (uint32_t v = 0; v < L.vsteps; v++) {
              V3 lpos = point_on_light(L, li, u, v, 1u, rp, src.id(item), path);
              V3 diffuse, specular;
              if (phong_sample(lpos, over_point, eyev, normalv, effective, lcol, m_diffuse, m_specular, m_shininess,
                               diffuse, specular))
                sum = (sum + diffuse) + specular;
            }
          sum = (1.f / nb) * sum;
          surface_sum = surface_sum + (ambient + intensity * sum);
          pk_st3(pk, PK_SURFACE, surface_sum);
          li++;
          sample = 0;
          total = 0.f;
          st = ST_NEXT_LIGHT;
        }
      }
    } else if constexpr ((FEAT & RT_F_COLLECT) != 0u) {  // JOB_COLLECT
      cn.overflow += sink.overflow();
      refractive_indices_of(sc, xl, sink.count(), hit_t, hit_shape, n1, n2);
      st = ST_SECONDARY;
    }

    // ---------------- advance the shading state until a ray has to be traced ----------------
    while (st != ST_TRACE && st != ST_DONE && st != ST_IDLE) {
      if (st == ST_NEXT_LIGHT) {
        if (li < sc.n_lights) {
          const RtLight RT_CONST_AS& L = cst(sc.lights)[li];
          V3 lpos;
          if (!(FEAT & RT_F_AREA) || L.kind == RT_LIGHT_POINT) {
            lpos = v3(L.position[0], L.position[1], L.position[2]);
          } else {
            uint32_t u = sample / L.vsteps, v = sample % L.vsteps;  // u outer, v inner (lights.rs:125-126)
            lpos = point_on_light(L, li, u, v, 0u, rp, src.id(item), path);
          }
          const V3 over_point = (pk_ld3(pk, PK_SRAY_O) + hit_t * pk_ld3(pk, PK_SRAY_D)) + 1e-4f * normalv;
          V3 v = lpos - over_point;
          distance = magnitude(v);
          tray = Ray{over_point, v / distance};
          cn.shadow++;
          if (COUNT) cn.ref_rays += ref_multiplicity(sc.n_lights, remaining0 - remaining);
          shadow_tF = __builtin_inff();
          shadow_sF = -1;
          job = (sc.props & RT_SCENE_SHADOW_TWO_PASS) ? JOB_SHADOW_F : JOB_SHADOW;
          st = ST_TRACE;
        } else {
          st = ST_LIGHTS_DONE;
        }
      } else if (st == ST_LIGHTS_DONE) {
        pk_st3(pk, PK_RESULT, pk_ld3(pk, PK_RESULT) + pk_ld(pk, PK_WEIGHT) * pk_ld3(pk, PK_SURFACE));
        // ---- secondary rays (world.rs:78-86, :113-148); once per light in the reference (Q4) ----
        const RtMaterial RT_CONST_AS& m = cst(sc.materials)[mat_id];
        bool want_reflect = m.reflective != 0.f, want_refract = m.transparency != 0.f;
        if (remaining == 0 || sc.n_lights == 0 || (!want_reflect && !want_refract)) {
          st = ST_NEXT_RAY;
        } else if ((FEAT & RT_F_COLLECT) && want_refract && !(sc.props & RT_SCENE_RI_ALL_ONE)) {
          tray = Ray{pk_ld3(pk, PK_SRAY_O), pk_ld3(pk, PK_SRAY_D)};
          job = JOB_COLLECT;
          st = ST_TRACE;
        } else {
          n1 = 1.f;
          n2 = 1.f;
          st = ST_SECONDARY;
        }
      } else if (st == ST_SECONDARY) {
        const RtMaterial RT_CONST_AS& m = cst(sc.materials)[mat_id];
        const float m_reflective = m.reflective, m_transparency = m.transparency;
        const Ray sray{pk_ld3(pk, PK_SRAY_O), pk_ld3(pk, PK_SRAY_D)};
        const float weight = pk_ld(pk, PK_WEIGHT);
        const V3 eyev = -sray.d;
        const V3 point = sray.o + hit_t * sray.d;
        bool want_reflect = m_reflective != 0.f, want_refract = m_transparency != 0.f;
        float reflectance = 1.f, transmittance = 1.f;
        if (m_reflective > 0.f && m_transparency > 0.f) {  // Computations::schlick (intersections.rs:177-192)
          float cosv = dot(eyev, normalv);
          bool tir = false;
          if (n1 > n2) {
            float n = n1 / n2;
            float sin2_t = (n * n) * (1.f - cosv * cosv);
            if (sin2_t > 1.f)
              tir = true;
            else
              cosv = __builtin_sqrtf(1.f - sin2_t);
          }
          if (tir) {
            reflectance = 1.f;
          } else {
            float r = (n1 - n2) / (n1 + n2);
            float r0 = r * r;
            float x = 1.f - cosv;
            float x2 = x * x;
            float x4 = x2 * x2;
            reflectance = r0 + (1.f - r0) * (x * x4);
          }
          transmittance = 1.f - reflectance;
        }
        if (want_refract) {  // refracted_color (world.rs:127-148); pushed first so reflection is traced first
          float n_ratio = n1 / n2;
          float cos_i = dot(eyev, normalv);
          float sin2_t = (n_ratio * n_ratio) * (1.f - cos_i * cos_i);
          if (!(sin2_t > 1.f)) {
            float cos_t = __builtin_sqrtf(1.f - sin2_t);
            V3 direction = (n_ratio * cos_i - cos_t) * normalv - n_ratio * eyev;
            float w = weight * nl * m_transparency * transmittance;
            if (sp < Lim<SIDE>::kRayStack) {
              const V3 under_point = point - 1e-4f * normalv;
              stack[sp++] = PendingRay{Ray{under_point, direction}, w, remaining - 1u, path * 2u + 1u};
              cn.refract++;
              if (COUNT) cn.ref_rays += ref_multiplicity(sc.n_lights, remaining0 - remaining + 1u);
            } else {
              cn.overflow++;
            }
          }
        }
        if (want_reflect) {  // reflected_color (world.rs:113-125)
          float w = weight * nl * m_reflective * reflectance;
          if (sp < Lim<SIDE>::kRayStack) {
            const V3 over_point = point + 1e-4f * normalv;
            stack[sp++] = PendingRay{Ray{over_point, reflect(sray.d, normalv)}, w, remaining - 1u, path * 2u};
            cn.reflect++;
            if (COUNT) cn.ref_rays += ref_multiplicity(sc.n_lights, remaining0 - remaining + 1u);
          } else {
            cn.overflow++;
          }
        }
        st = ST_NEXT_RAY;
      } else {  // ST_NEXT_RAY
        if (sp > 0) {
          PendingRay pr = stack[--sp];
          pk_st3(pk, PK_SRAY_O, pr.ray.o);
          pk_st3(pk, PK_SRAY_D, pr.ray.d);
          pk_st(pk, PK_WEIGHT, pr.weight);
          remaining = pr.remaining;
          path = pr.path;
          tray = pr.ray;
          job = JOB_SHADE;
          st = ST_TRACE;
        } else {
          st = ST_DONE;
        }
      }
    }
    if (st == ST_DONE) {
      const uint32_t work = COUNT ? cn.node_visits + cn.aux_tests + cn.tri_tests + cn.shape_tests - work_start : 0u;
      src.finish(item, pk_ld3(pk, PK_RESULT), work, cn);
      st = ST_IDLE;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Direct integrator for worlds without groups (every sample scene built from loose shapes and
// CSG): World::intersect is a short loop, so the shadow / secondary traversals are simply inlined
// where shade_hit needs them (world.rs:64-148) instead of going through the job loop above.
// Same arithmetic, same accumulation order: the two integrators produce the same bits.
// ---------------------------------------------------------------------------------------
template <bool COUNT>
__device__ __forceinline__ void simple_intersect(const Scene& sc, const Ray& ray, Sink& sink, Counters& cn) {
  if (sc.n_csg_roots) {  // world.rs:45-49
    CsgListT<false> L;
    L.overflow = 0;
    for (uint32_t c = 0; c < sc.n_csg_roots; c++) {
      csg_eval(sc, sc.csg_roots[c], ray, L);
      for (int i = 0; i < L.n; i++)
        if (sink.takes(cst(sc.shapes)[L.s[i]].kind_flags)) sink.add(L.t[i], L.s[i], __builtin_nanf(""), __builtin_nanf(""));
    }
    cn.overflow += L.overflow;
  }
  if (COUNT) cn.shape_tests += sc.n_loose;
  for (uint32_t k = sc.n_shapes - sc.n_loose; k < sc.n_shapes; k++)  // world.rs:56-59
    if (!sink.done()) shape_intersect(sc, k, ray, sink);
}
template <bool COUNT>
__device__ __forceinline__ Hit direct_closest(const Scene& sc, const Ray& ray, Counters& cn) {
  Sink sink;
  sink.init_closest();
  simple_intersect<COUNT>(sc, ray, sink, cn);
  return Hit{sink.t, sink.u(), sink.v(), sink.shape};
}
// The shadow rays of one hit all leave over_point (world.rs:101-105): a world of a few loose shapes and no csg keeps that
// point in every shape's space (RT_HOIST_SHAPES of them: 12 registers), so a ray costs each shape the transform of its
// direction only, and tests the shadow-casting shapes first: World::is_shadowed asks whether the CLOSEST hit casts a
// shadow, i.e. whether some hit X on a shadow-casting shape lies before the light and no hit on a shadow-less shape lies
// before X in (t, rank) order -- without an X the shadow-less shapes (soft_shadows.yaml: the cube around the light,
// the dearest test of the four) need not be looked at.
#define RT_HOIST_SHAPES 4u
// (only in the HOIST instantiation of the direct kernel, which the host launches for worlds of at most RT_HOIST_SHAPES
// loose shapes and nothing else: rt_hip.hip launch_part.  The points live in LDS, one column per lane, so that the
// loop over the shapes stays a loop: unrolled for registers it was eight inlined copies of Shape::intersect and 250
// spilled registers.)
struct ShadowOrigin {
  uint32_t off;
  __device__ __forceinline__ float* base() const {
    __shared__ float pts[3u * RT_HOIST_SHAPES * RT_BLOCK];
    return pts;
  }
  __device__ __forceinline__ V3 get(uint32_t k) const {
    const float* b = base();
    return v3(b[(3u * k) * RT_BLOCK + off], b[(3u * k + 1u) * RT_BLOCK + off], b[(3u * k + 2u) * RT_BLOCK + off]);
  }
  __device__ __forceinline__ void put(uint32_t k, V3 p) const {
    float* b = base();
    b[(3u * k) * RT_BLOCK + off] = p.x;
    b[(3u * k + 1u) * RT_BLOCK + off] = p.y;
    b[(3u * k + 2u) * RT_BLOCK + off] = p.z;
  }
};
__device__ __forceinline__ ShadowOrigin shadow_origin(const Scene& sc, V3 p) {
  ShadowOrigin so{threadIdx.x};
  for (uint32_t k = 0; k < sc.n_loose && k < RT_HOIST_SHAPES; k++) {
    const uint32_t xform = cst(sc.shapes)[sc.n_shapes - sc.n_loose + k].xform;
    so.put(k, xform ? xf_point(cst(sc.xforms)[xform].m, p) : p);
  }
  return so;
}
template <bool COUNT>
__device__ __forceinline__ bool hoisted_is_shadowed(const Scene& sc, const ShadowOrigin& so, V3 light_position, V3 p,
                                                    Counters& cn, uint32_t mult) {
  V3 v = light_position - p;
  float distance = magnitude(v);
  V3 direction = v / distance;
  cn.shadow++;
  if (COUNT) cn.ref_rays += mult;
  if (COUNT) cn.shape_tests += sc.n_loose;
  const uint32_t first = sc.n_shapes - sc.n_loose;
  // pass 0, X: the closest hit on a shadow-casting shape in front of the light (a CLOSEST search that skips the
  // shadow-less shapes); pass 1: a hit on a shadow-less shape before X in (t, rank) order?  (a CLOSEST_NOSHADOW
  // search that starts from X)
  Sink sink;
  sink.init_closest();
  sink.set_cap(distance);
  int x_shape = -1;
  for (uint32_t pass = 0; pass < 2u; pass++) {
    for (uint32_t k = 0; k < sc.n_loose; k++) {
      const bool casts = (cst(sc.shapes)[first + k].kind_flags & RT_SHAPE_SHADOW) != 0u;
      if (casts == (pass == 0u)) shape_intersect_from(sc, first + k, so.get(k), direction, sink);
    }
    if (pass == 1u) break;
    x_shape = sink.shape;
    if (!wave_any(x_shape >= 0) || !(sc.props & RT_SCENE_HAS_NOSHADOW)) break;  // nobody in the wave has an X, or nothing is shadow-less
    const float tx = sink.t;
    sink.init(SINK_CLOSEST_NOSHADOW);
    sink.t = tx;
    sink.shape = x_shape;
    sink.lim = x_shape >= 0 ? tx : -__builtin_inff();  // (a lane without an X has nothing to look for)
    sink.bound = x_shape >= 0 ? sink.bound : -__builtin_inff();
  }
  return x_shape >= 0 && sink.shape == x_shape;
}
// World::is_shadowed (world.rs:101-111), see Sink::init_any_shadow for the two equivalent searches
template <bool COUNT>
__device__ __forceinline__ bool direct_is_shadowed(const Scene& sc, V3 light_position, V3 p, Counters& cn, uint32_t mult) {
  V3 v = light_position - p;
  float distance = magnitude(v);
  V3 direction = v / distance;
  cn.shadow++;
  if (COUNT) cn.ref_rays += mult;
  Sink sink;
  if (sc.props & RT_SCENE_HAS_NOSHADOW) {
    sink.init_closest();
    sink.set_cap(distance);
    simple_intersect<COUNT>(sc, Ray{p, direction}, sink, cn);
    return sink.shape >= 0 && sink.t < distance && (cst(sc.shapes)[sink.shape].kind_flags & RT_SHAPE_SHADOW);
  }
  sink.init_any_shadow(distance, __builtin_inff(), -1);
  simple_intersect<COUNT>(sc, Ray{p, direction}, sink, cn);
  return sink.found();
}

template <bool COUNT, bool HOIST = false>
__device__ __forceinline__ V3 color_at_direct(const Scene& sc, const RenderParams& rp, Ray primary, uint32_t remaining0,
                              uint32_t pixel, Counters& cn, Hit* primary_hit) {
  PendingRay stack[RT_MAX_RAY_STACK];
  int sp = 0;
  stack[sp++] = PendingRay{primary, 1.f, remaining0, 1u};
  V3 result = v3(0.f, 0.f, 0.f);
  bool first = true;
  const float nl = (float)sc.n_lights;
  while (sp > 0) {
    PendingRay pr = stack[--sp];
    const Ray ray = pr.ray;
    Hit hit = direct_closest<COUNT>(sc, ray, cn);
    if (first) {
      if (primary_hit) *primary_hit = hit;
      first = false;
    }
    if (hit.shape < 0) continue;
    if (COUNT) cn.shaded++;
    const uint32_t mult = COUNT ? ref_multiplicity(sc.n_lights, remaining0 - pr.remaining) : 0u;
    // ---- Computations::prepare (intersections.rs:121-139) ----
    const u4n RT_CONST_AS* raw = cst(reinterpret_cast<const u4n*>(sc.shapes + hit.shape));
    u4n w0 = raw[0], w1 = raw[1];
    uint32_t kf = w0.x, sxf = w0.y, mat_id = w0.z, prim = w1.y;
    int parent = (int)w0.w;
    float smn = __uint_as_float(w1.z), smx = __uint_as_float(w1.w);
    if ((kf & RT_SHAPE_KIND_MASK) == RT_SMOOTH_TRIANGLE && hit.u != hit.u) {
      // hit came through a csg list, which does not carry (u,v): recompute them (same arithmetic)
      Ray local = ray;
      V3 p1, e1, e2;
      load_tri(sc, prim, p1, e1, e2);
      if (sxf) local = xf_ray(sc, sxf, ray);
      float tt;
      isect_triangle(local, p1, e1, e2, tt, hit.u, hit.v);
    }
    V3 point = ray.o + hit.t * ray.d;
    V3 eyev = -ray.d;
    V3 lp = world_to_object(sc, sxf, parent, point);
    V3 ln = local_normal_at(sc, kf, smn, smx, prim, lp, hit.u, hit.v);
    V3 normalv = normal_to_world(sc, sxf, parent, ln);
    if (dot(normalv, eyev) < 0.f) normalv = -normalv;
    V3 reflectv = reflect(ray.d, normalv);
    V3 over_point = point + 1e-4f * normalv;
    V3 under_point = point - 1e-4f * normalv;

    const RtMaterial RT_CONST_AS& m = cst(sc.materials)[mat_id];
    const float m_ambient = m.ambient, m_diffuse = m.diffuse, m_specular = m.specular, m_shininess = m.shininess;
    const float m_reflective = m.reflective, m_transparency = m.transparency;
    V3 mcolor;
    if (m.pattern >= 0) {  // pattern_at_shape at over_point (patterns.rs:79-91, world.rs:72)
      const RtPattern& pat = sc.patterns[m.pattern];
      V3 op = world_to_object(sc, sxf, parent, over_point);
      V3 pp = xf_point(cst(sc.xforms)[pat.xform].m, op);
      mcolor = pattern_at(sc, pat, pp);
    } else {
      mcolor = v3(m.color[0], m.color[1], m.color[2]);
    }

    // ---- World::shade_hit (world.rs:64-89): direct lighting, summed over lights ----
    V3 surface_sum = v3(0.f, 0.f, 0.f);
    for (uint32_t li = 0; li < sc.n_lights; li++) {
      const RtLight RT_CONST_AS& L = cst(sc.lights)[li];
      V3 lcol = v3(L.intensity[0], L.intensity[1], L.intensity[2]);
      V3 effective = mulc(mcolor, lcol);
      V3 ambient = m_ambient * effective;
      V3 surface;
      if (L.kind == RT_LIGHT_POINT) {
        V3 lpos = v3(L.position[0], L.position[1], L.position[2]);
        bool shadowed;
        if constexpr (HOIST)
          shadowed = hoisted_is_shadowed<COUNT>(sc, shadow_origin(sc, over_point), lpos, over_point, cn, mult);
        else
          shadowed = direct_is_shadowed<COUNT>(sc, lpos, over_point, cn, mult);
        float intensity = shadowed ? 0.f : 1.f;  // lights.rs:50-56
        V3 diffuse = v3(0.f, 0.f, 0.f), specular = v3(0.f, 0.f, 0.f);
        phong_sample(lpos, over_point, eyev, normalv, effective, lcol, m_diffuse, m_specular, m_shininess, diffuse,
                     specular);
        surface = ambient + intensity * (diffuse + specular);  // materials.rs:134
      } else {
        // AreaLight::intensity_at (lights.rs:123-134)
        float total = 0.f;
        ShadowOrigin so;
        if (HOIST) so = shadow_origin(sc, over_point);
        for (uint32_t u = 0; u < L.usteps; u++)
          for (uint32_t v = 0; v < L.vsteps; v++) {
            V3 lpos = point_on_light(L, li, u, v, 0u, rp, pixel, pr.path);
            bool shadowed;
            if constexpr (HOIST)
              shadowed = hoisted_is_shadowed<COUNT>(sc, so, lpos, over_point, cn, mult);
            else
              shadowed = direct_is_shadowed<COUNT>(sc, lpos, over_point, cn, mult);
            if (!shadowed) total += 1.f;
          }
        float nb = (float)(L.usteps * L.vsteps);
        float intensity = total / nb;
        // materials.rs:136-175
        V3 sum = v3(0.f, 0.f, 0.f);
        for (uint32_t u = 0; u < L.usteps; u++)
          for (uint32_t v = 0; v < L.vsteps; v++) {
            V3 lpos = point_on_light(L, li, u, v, 1u, rp, pixel, pr.path);
            V3 diffuse, specular;
            if (phong_sample(lpos, over_point, eyev, normalv, effective, lcol, m_diffuse, m_specular, m_shininess,
                             diffuse, specular))
              sum = (sum + diffuse) + specular;
          }
        sum = (1.f / nb) * sum;
        surface = ambient + intensity * sum;
      }
      surface_sum = surface_sum + surface;
    }
    result = result + pr.weight * surface_sum;

    // ---- secondary rays (world.rs:78-86, :113-148); once per light in the reference (Q4) ----
    if (pr.remaining == 0 || sc.n_lights == 0) continue;
    bool want_reflect = m_reflective != 0.f;
    bool want_refract = m_transparency != 0.f;
    if (!want_reflect && !want_refract) continue;
    float reflectance = 1.f, transmittance = 1.f;
    float n1 = 1.f, n2 = 1.f;
    if (want_refract && !(sc.props & RT_SCENE_RI_ALL_ONE)) {
      XsList xl;  // (no side lists in this kernel: a context that has them renders through the persistent one)
      Sink cs;
      cs.init_collect(&xl, hit.t);
      simple_intersect<COUNT>(sc, ray, cs, cn);
      cn.overflow += cs.overflow();
      refractive_indices_of(sc, xl, cs.count(), hit.t, hit.shape, n1, n2);
    }
    if (m_reflective > 0.f && m_transparency > 0.f) {  // Computations::schlick (intersections.rs:177-192)
      float cosv = dot(eyev, normalv);
      bool tir = false;
      if (n1 > n2) {
        float n = n1 / n2;
        float sin2_t = (n * n) * (1.f - cosv * cosv);
        if (sin2_t > 1.f)
          tir = true;
        else
          cosv = __builtin_sqrtf(1.f - sin2_t);
      }
      if (tir) {
        reflectance = 1.f;
      } else {
        float r = (n1 - n2) / (n1 + n2);
        float r0 = r * r;
        float x = 1.f - cosv;
        float x2 = x * x;
        float x4 = x2 * x2;
        reflectance = r0 + (1.f - r0) * (x * x4);
      }
      transmittance = 1.f - reflectance;
    }
    if (want_refract) {  // refracted_color (world.rs:127-148); pushed first so reflection is traced first
      float n_ratio = n1 / n2;
      float cos_i = dot(eyev, normalv);
      float sin2_t = (n_ratio * n_ratio) * (1.f - cos_i * cos_i);
      if (!(sin2_t > 1.f)) {
        float cos_t = __builtin_sqrtf(1.f - sin2_t);
        V3 direction = (n_ratio * cos_i - cos_t) * normalv - n_ratio * eyev;
        float w = pr.weight * nl * m_transparency * transmittance;
        if (sp < RT_MAX_RAY_STACK) {
          stack[sp++] = PendingRay{Ray{under_point, direction}, w, pr.remaining - 1u, pr.path * 2u + 1u};
          cn.refract++;
          if (COUNT) cn.ref_rays += mult * sc.n_lights;
        } else {
          cn.overflow++;
        }
      }
    }
    if (want_reflect) {  // reflected_color (world.rs:113-125)
      float w = pr.weight * nl * m_reflective * reflectance;
      if (sp < RT_MAX_RAY_STACK) {
        stack[sp++] = PendingRay{Ray{over_point, reflectv}, w, pr.remaining - 1u, pr.path * 2u};
        cn.reflect++;
        if (COUNT) cn.ref_rays += mult * sc.n_lights;
      } else {
        cn.overflow++;
      }
    }
  }
  return result;
}

}  // namespace rtd
