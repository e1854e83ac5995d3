// rt_hip.hip — librt_hip.so: the C ABI of include/rt_abi.h and the gfx950 render kernels.
//
// Build (see __graft_entry__.build): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//   -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared
// No FMA contraction and correctly rounded / and sqrt are REQUIRED for parity.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // declarations only: librccl.so is opened when rt_render_multi_device first gathers (rccl_api)

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_accel.h"
#include "rt_accel_device.h"
#include "rt_device.h"

using namespace rtd;

namespace {

#ifndef RT_QUEUE_CLASSES
#define RT_QUEUE_CLASSES 16u  // tile queues per launch, a multiple of 8 (the XCDs): two per XCD (8 / 24 / 32 measured slower at three waves per SIMD)
#endif
#define RT_N_COUNTERS (64 + 8 * RT_QUEUE_CLASSES)  // u64 slots per counter block: 0-10 RtStats, 11-23 and 25-60 diagnostics (RT_PHASE_TIMERS), 24 reference rays, 64.. work queues
#ifndef RT_BLOCKS_PER_CU
#define RT_BLOCKS_PER_CU 3  // 256-thread blocks resident per CU at three waves per SIMD
#endif

struct PartInfo {
  uint32_t tile_h, n_parts, part, rows;  // rows = number of local rows
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void flush_counters(const Counters& cn, unsigned long long* g, unsigned long long* sticky) {
  uint32_t a = wave_sum(cn.primary), b = wave_sum(cn.shadow), c = wave_sum(cn.reflect), d = wave_sum(cn.refract),
           e = wave_sum(cn.overflow);
  uint32_t w0 = wave_sum(cn.node_visits), w1 = wave_sum(cn.tri_tests), w2 = wave_sum(cn.shape_tests),
           w3 = wave_sum(cn.shaded), w4 = wave_sum(cn.xforms), w5 = wave_sum(cn.aux_tests), w6 = wave_sum(cn.ref_rays);
#ifdef RT_PHASE_TIMERS
  for (int k = 0; k < 8; k++) {
    uint32_t v = wave_sum(cn.u[k]);
    if ((threadIdx.x & 63) == 0) atomicAdd(g + 16 + k, (unsigned long long)v);
  }
#endif
  if ((threadIdx.x & 63) == 0) {
#ifdef RT_PHASE_TIMERS
    atomicAdd(g + 11, (unsigned long long)cn.t_world);
    atomicAdd(g + 12, (unsigned long long)cn.t_aux);
    atomicAdd(g + 13, (unsigned long long)cn.t_shapes);
    atomicAdd(g + 10, (unsigned long long)cn.t_load);  // slot of aux_tests: meaningful in non-COUNT runs only
    atomicAdd(g + 15, (unsigned long long)cn.t_jobs);
    atomicAdd(g + 25, (unsigned long long)cn.t_refill);
    atomicAdd(g + 26, (unsigned long long)cn.t_iters);
    atomicAdd(g + 27, (unsigned long long)cn.t_draw);
#endif
    if (w0) atomicAdd(g + 5, (unsigned long long)w0);
    if (w1) atomicAdd(g + 6, (unsigned long long)w1);
    if (w2) atomicAdd(g + 7, (unsigned long long)w2);
    if (w3) atomicAdd(g + 8, (unsigned long long)w3);
    if (w4) atomicAdd(g + 9, (unsigned long long)w4);
    if (w5) atomicAdd(g + 10, (unsigned long long)w5);
    if (w6) atomicAdd(g + 24, (unsigned long long)w6);
    if (a) atomicAdd(g + 0, (unsigned long long)a);
    if (b) atomicAdd(g + 1, (unsigned long long)b);
    if (c) atomicAdd(g + 2, (unsigned long long)c);
    if (d) atomicAdd(g + 3, (unsigned long long)d);
    if (e) {
      atomicAdd(g + 4, (unsigned long long)e);
      atomicAdd(sticky, (unsigned long long)e);  // survives the launch: reported by the next synchronising call (rt_check)
    }
  }
}

// Camera::render (camera.rs:66-84).  Rows are the interleaved scanline tiles of this part.
// Two kernel families:
//   k_render_jobs    job-loop integrator (rt_device.h integrate) on persistent waves: pixels are work items in 8x8
//                    tile order, drawn from a queue as lanes go idle.  Register budget capped at 128 VGPRs so four
//                    waves share a SIMD (profiles/README.md).
//   k_render_direct  direct integrator (color_at_direct), one lane per pixel, one wave per 8x8 tile, for group-less
//                    worlds whose cost is the shadow-sample loop of an area light.
#ifndef RT_WAVES_PER_EU
// Three waves per SIMD (168 VGPRs, no spills) since the auxiliary traversal became wave-cooperative: as fast as four
// (128 VGPRs, ~58 spilled registers and 2 GB of scratch written per frame) on dragons and 5-30 % faster on the lighter
// scenes.  Round 1's per-lane traversal needed the fourth wave to hide its fetch latency (5.6 vs 7.4 ms).
#define RT_WAVES_PER_EU 3
#endif
#ifndef RT_CULL_MIN_SHAPES
#define RT_CULL_MIN_SHAPES 6u
#endif
#define RT_QUEUE_SLOT 64  // d_counters slot of the first work queue of the persistent kernels (one per class, RT_QUEUE_STRIDE apart)

__device__ __forceinline__ Ray ray_for_pixel(const RtCamera& cam, uint32_t px, uint32_t py) {  // camera.rs:45-64
  float xoffset = ((float)px + 0.5f) * cam.pixel_size;
  float yoffset = ((float)py + 0.5f) * cam.pixel_size;
  float world_x = cam.half_width - xoffset;
  float world_y = cam.half_height - yoffset;
  V3 pixel = xf_point(cam.inv_transform, v3(world_x, world_y, -1.f));
  V3 origin = xf_point(cam.inv_transform, v3(0.f, 0.f, 0.f));
  return Ray{origin, normalize(pixel - origin)};
}
struct PixelOut {
  float* __restrict__ out;
  uint8_t* __restrict__ mask;
  int32_t* __restrict__ oid;
  float* __restrict__ ht;
};
__device__ __forceinline__ void write_hit(const PixelOut& o, size_t idx, const Hit& ph) {
  if (o.mask) o.mask[idx] = ph.shape >= 0 ? 1 : 0;
  if (o.oid) o.oid[idx] = ph.shape;
  if (o.ht) o.ht[idx] = ph.shape >= 0 ? ph.t : __builtin_nanf("");
}
template <bool COUNT>
__device__ __forceinline__ void write_colour(const Scene& sc, const PixelOut& o, size_t idx, V3 col, float work,
                                             float work_shadow, float work_collect) {
  o.out[3 * idx + 0] = col.x;
  o.out[3 * idx + 1] = col.y;
  o.out[3 * idx + 2] = col.z;
  if (COUNT && o.ht && (sc.flags & RT_RENDER_WORK_MAP))  // diagnostics: this pixel's traversal work instead of hit t
    o.ht[idx] = (sc.flags & 0x10u) ? work_shadow : (sc.flags & 0x20u) ? work_collect : work;
}
// Work items of a frame: item = (supertile * 256 + tile in supertile) * 64 + pixel in tile.  A tile is 8x8 pixels (one
// wave), a supertile 8x8 tiles (64x64 pixels; 128x128 and 32x32 measured the same within 2 %); supertile L belongs to class L % 8.  Items past the image edge (ragged
// tiles and supertiles) are handed out and skipped.
#ifndef RT_SUPER_LOG2
#define RT_SUPER_LOG2 3u
#endif
#define RT_SUPER (1u << RT_SUPER_LOG2)
template <bool COUNT>
struct PixelSource {
  static constexpr uint32_t kClasses = RT_QUEUE_CLASSES;
  const Scene* sc;
  RtCamera cam;
  PartInfo part;
  PixelOut o;
  uint32_t super_x, super_y;
  __device__ unsigned long long class_items(uint32_t c) const {
    const uint32_t ns = super_x * super_y;
    return (unsigned long long)(ns > c ? (ns - c + kClasses - 1u) / kClasses : 0u) * (RT_SUPER * RT_SUPER * 64u);
  }
  // Supertiles are handed out from the bottom of the image upwards. The frame ends when the last tile drawn is done, so
  // the last tiles should be cheap ones; with a camera above a floor looking slightly down (every sample scene of
  // the reference) the cheap tiles, sky, are at the top. Measured against top-down at 4K: dragons 3.55 -> 3.42 ms,
  // cover 3.58 -> 3.25 ms, dragons 1080p 1.60 -> 1.38 ms; a scene the other way round loses what these gain.
  // Round 2 measured the alternatives again: top-down and middle-rows-outwards lose 4 % (dragons) and 10 % (cover), and
  // the TRUE heaviest-first order -- supertiles sorted by the work the instrumented kernel counted in them -- is no
  // better than bottom-up (dragons 2.53 vs 2.49 ms, cover 2.92 vs 2.90 ms): order has nothing left to give.
  __device__ uint32_t item_of(uint32_t c, uint32_t n) const {
    const uint32_t per = RT_SUPER * RT_SUPER * 64u;
    return ((n / per) * kClasses + c) * per + n % per;
  }
  __device__ void locate(uint32_t item, uint32_t& px, uint32_t& ly) const {
    const uint32_t w = item & 63u, tis = (item >> 6) & (RT_SUPER * RT_SUPER - 1u);
    const uint32_t L = super_x * super_y - 1u - (item >> (2u * RT_SUPER_LOG2 + 6u));
    const uint32_t srow = L / super_x, scol = L % super_x;
    const uint32_t tx = scol * RT_SUPER + (tis & (RT_SUPER - 1u)), ty = srow * RT_SUPER + (RT_SUPER - 1u - tis / RT_SUPER);
    px = tx * 8u + (w & 7u);
    ly = ty * 8u + (w >> 3);
  }
  __device__ uint32_t global_row(uint32_t ly) const {
    return (ly / part.tile_h) * (part.n_parts * part.tile_h) + part.part * part.tile_h + (ly % part.tile_h);
  }
  // The token a lane keeps for its pixel is (row in the part << 16 | column): locate() divides by the supertile count,
  // ~40 instructions, and finish / first_hit / id would each pay it again (rt_scene limits: frames below 65536 x 65536).
  __device__ bool start(uint32_t c, uint32_t n, uint32_t& item, Ray& r) const {
    uint32_t px, ly;
    locate(item_of(c, n), px, ly);
    item = (ly << 16) | px;
    if (px >= cam.hsize || ly >= part.rows) return false;
    r = ray_for_pixel(cam, px, global_row(ly));
    return true;
  }
  __device__ uint32_t id(uint32_t item) const {  // absolute pixel: the same random numbers whatever the partition
    return global_row(item >> 16) * cam.hsize + (item & 0xFFFFu);
  }
  __device__ void first_hit(uint32_t item, const Hit& ph) const {
    write_hit(o, (size_t)(item >> 16) * cam.hsize + (item & 0xFFFFu), ph);
  }
  __device__ void finish(uint32_t item, V3 col, uint32_t work, Counters& cn) const {
    float ws = 0.f, wc = 0.f;
    if (COUNT) {  // per-pixel shares for the work map
      ws = (float)cn.work_shadow, wc = (float)cn.work_collect;
      cn.work_shadow = 0;
      cn.work_collect = 0;
    }
    write_colour<COUNT>(*sc, o, (size_t)(item >> 16) * cam.hsize + (item & 0xFFFFu), col, (float)work, ws, wc);
  }
};
// FEAT: the RT_F_* the kernel is compiled with (jobs_kernel below picks one).  Kernels without the group walk are not
// bound by its registers: RT_FLAT_WAVES_PER_EU
#ifndef RT_FLAT_WAVES_PER_EU
#define RT_FLAT_WAVES_PER_EU 4  // 103-116 VGPRs without the walk: four waves per SIMD without a spill
#endif
// The kernel for worlds of groups and nothing else (dragons.yaml) exists twice: at three waves per SIMD without a
// spill, and (RT_F_WAVES4) at four, 128 VGPRs with ~51 of them spilled and RT_AUX_ROWS_LEAN rows of traversal stack in LDS
// so that four blocks fit a CU (rt_device.h aux_stack).  Four waves hide more of the fetches and win where there is work
// for them (dragons 4K 2.27 -> 2.12 ms, 1080p 0.85 -> 0.83 ms); a launch that is mostly its own tail -- an eighth of the
// frame -- is faster on the waves that do not spill (0.57 against 0.60 ms): jobs_kernel picks by the number of pixels.
#ifndef RT_LEAN_WAVES_PER_EU
#define RT_LEAN_WAVES_PER_EU 4
#endif
#ifndef RT_LEAN_MIN_PIXELS
#define RT_LEAN_MIN_PIXELS 1500000ull
#endif
#define RT_JOBS_WAVES ((FEAT & RT_F_GROUPS) ? ((FEAT & RT_F_WAVES4) ? RT_LEAN_WAVES_PER_EU : RT_WAVES_PER_EU) : RT_FLAT_WAVES_PER_EU)
template <bool COUNT, bool CULL, bool SIDE = false, uint32_t FEAT = RT_F_ALL>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_JOBS_WAVES, RT_JOBS_WAVES)))
void k_render_jobs(Scene sc, RtCamera cam, RenderParams rp, PartInfo part, PixelOut o, unsigned long long* counters) {
#ifdef RT_PHASE_TIMERS
  const uint64_t t_pixel0 = __builtin_readcyclecounter();
  const unsigned long long t_wall0 = wall_clock64();
#endif
  PixelSource<COUNT> src{&sc, cam, part, o, (cam.hsize + 8u * RT_SUPER - 1u) / (8u * RT_SUPER), (part.rows + 8u * RT_SUPER - 1u) / (8u * RT_SUPER)};
  Counters cn = {};
  integrate<COUNT, CULL, SIDE, FEAT>(sc, rp, src, sc.recursion_limit, counters + RT_QUEUE_SLOT, cn);
#ifdef RT_PHASE_TIMERS
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(counters + 14, (unsigned long long)(__builtin_readcyclecounter() - t_pixel0));
    // when the waves of the launch end (100 MHz wall clock): sum, latest, and earliest start as max of the complement
    const unsigned long long t_end = wall_clock64();
    atomicAdd(counters + 30, t_end);
    atomicMax(counters + 31, t_end);
    atomicMax(counters + 29, ~t_wall0);
    atomicAdd(counters + 28, 1ull);
  }
#endif
  flush_counters(cn, counters, sc.sticky);
}
// the direct integrator is force-inlined so that the register budget applies to it (as a called function it took 248
// VGPRs whatever the kernel asked for: soft_shadows 4K 20.9 -> 12.4 ms at four waves per SIMD; 5: same, 6: slower)
#ifndef RT_DIRECT_WAVES_PER_EU
#define RT_DIRECT_WAVES_PER_EU 4
#endif
#ifndef RT_HOIST_WAVES_PER_EU
// the HOIST instantiation (soft_shadows.yaml 4K): 3 / 4 / 5 / 6 / 8 waves per SIMD = 9.66 / 8.51 / 7.92 / 7.75 / 7.79 ms
// with 0 / 38 / 77 / 166 / 204 spilled registers
#define RT_HOIST_WAVES_PER_EU 5
#endif
#define RT_DIRECT_ATTR \
  __attribute__((amdgpu_waves_per_eu(HOIST ? RT_HOIST_WAVES_PER_EU : RT_DIRECT_WAVES_PER_EU, HOIST ? RT_HOIST_WAVES_PER_EU : RT_DIRECT_WAVES_PER_EU)))
// HOIST: worlds of at most RT_HOIST_SHAPES loose shapes and nothing else (rt_device.h hoisted_is_shadowed)
template <bool COUNT, bool HOIST>
__global__ __launch_bounds__(RT_BLOCK) RT_DIRECT_ATTR
void k_render_direct(Scene sc, RtCamera cam, RenderParams rp, PartInfo part, PixelOut o, unsigned long long* counters) {
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#if RT_BLOCK == 256
  const uint32_t px = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t ly = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
#else  // one wave per block
  const uint32_t px = blockIdx.x * 8u + (lane & 7u);
  const uint32_t ly = blockIdx.y * 8u + (lane >> 3);
  (void)wave;
#endif
  Counters cn = {};
  if (px < cam.hsize && ly < part.rows) {
    const uint32_t py = (ly / part.tile_h) * (part.n_parts * part.tile_h) + part.part * part.tile_h + (ly % part.tile_h);
    cn.primary = 1;
    if (COUNT) cn.ref_rays = 1;
    Hit ph;
    V3 col = color_at_direct<COUNT, HOIST>(sc, rp, ray_for_pixel(cam, px, py), sc.recursion_limit, py * cam.hsize + px, cn, &ph);
    write_hit(o, (size_t)ly * cam.hsize + px, ph);
    write_colour<COUNT>(sc, o, (size_t)ly * cam.hsize + px, col,
                        (float)(cn.node_visits + cn.aux_tests + cn.tri_tests + cn.shape_tests), (float)cn.work_shadow,
                        (float)cn.work_collect);
  }
  flush_counters(cn, counters, sc.sticky);
}

// World::color_at for caller-supplied rays (world.rs:91-99).
struct RayOut {
  const float* __restrict__ origins;
  const float* __restrict__ directions;
  uint32_t n;
  float* __restrict__ out;
  float* __restrict__ hit_t;
  int32_t* __restrict__ hit_id;
  float* __restrict__ hit_uv;
};
__device__ __forceinline__ void write_ray_colour(const RayOut& o, uint32_t i, V3 col) {
  if (o.out) {
    o.out[3 * i] = col.x;
    o.out[3 * i + 1] = col.y;
    o.out[3 * i + 2] = col.z;
  }
}
__device__ __forceinline__ void write_ray_hit(const RayOut& o, uint32_t i, const Hit& ph) {
  if (o.hit_t) o.hit_t[i] = ph.shape >= 0 ? ph.t : __builtin_nanf("");
  if (o.hit_id) o.hit_id[i] = ph.shape;
  if (o.hit_uv) {
    o.hit_uv[2 * i] = ph.shape >= 0 && ph.u == ph.u ? ph.u : 0.f;
    o.hit_uv[2 * i + 1] = ph.shape >= 0 && ph.v == ph.v ? ph.v : 0.f;
  }
}
struct RaySource {
  static constexpr uint32_t kClasses = 1u;
  RayOut o;
  __device__ unsigned long long class_items(uint32_t) const { return o.n; }
  __device__ uint32_t item_of(uint32_t, uint32_t n) const { return n; }
  __device__ bool start(uint32_t, uint32_t i, uint32_t& item, Ray& r) const {
    item = i;
    r = Ray{v3(o.origins[3 * i], o.origins[3 * i + 1], o.origins[3 * i + 2]),
            v3(o.directions[3 * i], o.directions[3 * i + 1], o.directions[3 * i + 2])};
    return true;
  }
  __device__ uint32_t id(uint32_t i) const { return i; }
  __device__ void first_hit(uint32_t i, const Hit& ph) const { write_ray_hit(o, i, ph); }
  __device__ void finish(uint32_t i, V3 col, uint32_t, Counters&) const { write_ray_colour(o, i, col); }
};
template <bool COUNT, bool CULL, bool SIDE = false>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, RT_WAVES_PER_EU)))
void k_color_at_jobs(Scene sc, RenderParams rp, RayOut o, uint32_t remaining, unsigned long long* counters) {
  RaySource src{o};
  Counters cn = {};
  integrate<COUNT, CULL, SIDE>(sc, rp, src, remaining, counters + RT_QUEUE_SLOT, cn);
  flush_counters(cn, counters, sc.sticky);
}
__global__ __launch_bounds__(RT_BLOCK) void k_color_at_direct(Scene sc, RenderParams rp, RayOut o, uint32_t remaining,
                                                             unsigned long long* counters) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  Counters cn = {};
  if (i < o.n) {
    Ray r{v3(o.origins[3 * i], o.origins[3 * i + 1], o.origins[3 * i + 2]),
          v3(o.directions[3 * i], o.directions[3 * i + 1], o.directions[3 * i + 2])};
    cn.primary = 1;
    Hit ph;
    V3 col = color_at_direct<false>(sc, rp, r, remaining, i, cn, &ph);
    write_ray_colour(o, i, col);
    write_ray_hit(o, i, ph);
  }
  flush_counters(cn, counters, sc.sticky);
}

// The ids of primary hits on a triangle that is listed more than once (rt_device.h end_of_copy_run): a pass over the
// ids the render kernels wrote, launched only for scenes with such triangles and callers that ask for ids.  Nearly
// every thread is done after one look at its shape's record.
__global__ __launch_bounds__(256) void k_copies_pixels(Scene sc, RtCamera cam, PartInfo part, int32_t* __restrict__ oid) {
  const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (idx >= (size_t)part.rows * cam.hsize) return;
  const int w = oid[idx];
  if (w < 0 || next_copy(sc, w) == w) return;
  const uint32_t ly = (uint32_t)(idx / cam.hsize), px = (uint32_t)(idx % cam.hsize);
  const uint32_t py = (ly / part.tile_h) * (part.n_parts * part.tile_h) + part.part * part.tile_h + (ly % part.tile_h);
  oid[idx] = end_of_copy_run(sc, ray_for_pixel(cam, px, py), w);
}
__global__ __launch_bounds__(256) void k_copies_rays(Scene sc, RayOut o) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= o.n) return;
  const int w = o.hit_id[i];
  if (w < 0 || next_copy(sc, w) == w) return;
  const Ray r{v3(o.origins[3 * i], o.origins[3 * i + 1], o.origins[3 * i + 2]),
              v3(o.directions[3 * i], o.directions[3 * i + 1], o.directions[3 * i + 2])};
  o.hit_id[i] = end_of_copy_run(sc, r, w);
}

// What the reference does with the canvas after Camera::render (scene.rs:215-224), on the device: ordered dithering
// (Canvas::apply_dithering, canvas.rs:37-55) over the Bayer matrix of effects/dithering.rs:12-37, then the 8-bit
// quantisation of Canvas::to_ppm (canvas.rs:73-75).  A 4K frame leaves the GPU as 25 MB instead of 100.
// The matrix entry bayer[i][j] of the reference's recursion (each step: old * 4, then +3 / +2 / +1 for the quadrants
// [size+i][j] / [i][size+j] / [size+i][size+j]) is, bit by bit, sum over levels l of q(bit l of i, bit l of j) * 4^(L-1-l)
// with q(0,0) = 0, q(0,1) = 2, q(1,0) = 3, q(1,1) = 1, times 1/N^2: all exact in f32.
__device__ __forceinline__ float bayer_value(uint32_t i, uint32_t j, uint32_t levels) {
  uint32_t v = 0;
  for (uint32_t l = 0; l < levels; l++) {
    const uint32_t ib = (i >> l) & 1u, jb = (j >> l) & 1u;
    const uint32_t q = ib ? (jb ? 1u : 3u) : (jb ? 2u : 0u);
    v |= q << (2u * (levels - 1u - l));
  }
  return (float)v * (1.0f / (float)(1u << (2u * levels)));
}
__device__ __forceinline__ uint32_t quantise_u8(float c) {  // (0_f32.max(c).min(1.) * 255.).round() as u8
  const float lo = c > 0.f ? c : 0.f;  // NaN -> 0, as f32::max does
  const float cl = lo < 1.f ? lo : 1.f;
  return (uint32_t)roundf(cl * 255.f);  // half away from zero
}
__device__ __forceinline__ void post_pixel(float& r, float& g, float& b, uint32_t x, uint32_t y, uint32_t dither) {
  if (dither == RT_DITHER_NONE) return;
  const uint32_t levels = dither == RT_DITHER_BAYER_COLOR ? 2u : dither;  // Bayer2/4/8/16 = 1..4 levels; colour: N = 4
  const uint32_t mask = (1u << levels) - 1u;
  const float d = bayer_value(x & mask, y & mask, levels);  // bayer[x % N][y % N] (canvas.rs:41)
  if (dither == RT_DITHER_BAYER_COLOR) {
    r = r + d, g = g + d, b = b + d;
  } else {
    const bool black = r < d || g < d || b < d;
    r = g = b = black ? 0.f : 1.f;
  }
}
// four pixels per thread: three 16-B loads, three 4-B stores; `n` pixels in all, `hsize` per row
__global__ __launch_bounds__(256) void k_canvas_to_u8(const float* __restrict__ rgb, uint8_t* __restrict__ out, uint64_t n,
                                                      uint32_t hsize, PartInfo part, uint32_t dither) {
  const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (p0 >= n) return;
  float v[12];
  const bool aligned = (((uintptr_t)rgb & 15u) | ((uintptr_t)out & 3u)) == 0u;
  const uint32_t cnt = n - p0 >= 4u ? 4u : (uint32_t)(n - p0);
  if (cnt == 4u && aligned) {
    const float4* q = reinterpret_cast<const float4*>(rgb + 3u * p0);
    const float4 a = q[0], b = q[1], c = q[2];
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    v[8] = c.x, v[9] = c.y, v[10] = c.z, v[11] = c.w;
  } else {
    for (uint32_t k = 0; k < 12; k++) v[k] = k < 3u * cnt ? rgb[3u * p0 + k] : 0.f;
  }
  uint32_t bytes[12];
  for (uint32_t k = 0; k < 4; k++) {
    const uint64_t p = p0 + k;
    const uint32_t x = (uint32_t)(p % hsize), ly = (uint32_t)(p / hsize);
    const uint32_t y = (ly / part.tile_h) * (part.n_parts * part.tile_h) + part.part * part.tile_h + (ly % part.tile_h);
    post_pixel(v[3 * k], v[3 * k + 1], v[3 * k + 2], x, y, dither);
    for (uint32_t c = 0; c < 3; c++) bytes[3 * k + c] = quantise_u8(v[3 * k + c]);
  }
  if (cnt == 4u && aligned) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out + 3u * p0);
    for (uint32_t wd = 0; wd < 3; wd++)
      o[wd] = bytes[4 * wd] | (bytes[4 * wd + 1] << 8) | (bytes[4 * wd + 2] << 16) | (bytes[4 * wd + 3] << 24);
  } else {
    for (uint32_t k = 0; k < 3u * cnt; k++) out[3u * p0 + k] = (uint8_t)bytes[k];
  }
}

// The canvas of a frame that n GPUs rendered as interleaved parts, from the parts laid one after the other
// in `parts` (part g's rows_of(g) compacted rows from row first[g] on): row r of the canvas is row
// (r / (tile_h n)) tile_h + r % tile_h of part (r / tile_h) % n.  16 bytes per thread where the rows allow it.
struct PartLayout {
  uint32_t n, tile_h, vsize, row_floats;
  uint32_t first[16];  // RT_MULTI_MAX parts
};
__global__ __launch_bounds__(256) void k_deinterleave(const float* __restrict__ parts, float* __restrict__ canvas, PartLayout L) {
  const uint32_t r = blockIdx.y;
  const uint32_t g = (r / L.tile_h) % L.n;
  const uint32_t lr = (r / (L.tile_h * L.n)) * L.tile_h + r % L.tile_h;
  const float* src = parts + (size_t)(L.first[g] + lr) * L.row_floats;
  float* dst = canvas + (size_t)r * L.row_floats;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((L.row_floats & 3u) == 0u && ((((uintptr_t)parts) | ((uintptr_t)canvas)) & 15u) == 0u) {
    if (i * 4u < L.row_floats) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  } else {
    for (uint32_t k = i * 4u; k < i * 4u + 4u && k < L.row_floats; k++) dst[k] = src[k];
  }
}

}  // namespace

// =========================================================================================
// rt_color_at_static: staging and counters of the allocation-free entry point.  One array per GPU the library runs
// on, held until the process ends.  In host code the NAME below is only a stand-in for the symbol, not a device
// address: rt_context_create asks the runtime for this device's address (hipGetSymbolAddress) and everything --
// kernel arguments, memsets, copies -- uses that pointer.
//   layout (RT_STATIC_MAX_RAYS = N): origins 3N f32 | directions 3N | colours 3N | hit t N | hit uv 2N | hit id N i32
//                                    | counter block (RT_N_COUNTERS u64, 256-B aligned)
// =========================================================================================
#define RT_STATIC_WORDS (13u * RT_STATIC_MAX_RAYS)
#define RT_STATIC_BYTES (RT_STATIC_WORDS * 4u + RT_N_COUNTERS * 8u)
static __device__ __align__(256) unsigned char scratch[RT_STATIC_BYTES];

// =========================================================================================
// host side of the ABI
// =========================================================================================
#define RT_RING 8       // counter blocks of the asynchronous launches
#define RT_MAX_BANDS 4  // rt_render renders a full frame as up to this many row bands, copying each back while the next renders
struct RtContext {
  int device = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_band[RT_MAX_BANDS] = {nullptr, nullptr, nullptr, nullptr};
  // Counter blocks (RT_N_COUNTERS slots each; one slot of a block is the launch's work queue).  Every launch of
  // rt_render_device / rt_color_at takes the next block of a ring of RT_RING, so launches in flight on different
  // streams never share a queue; before a block is reused the new stream waits for the launch that used it last.
  // Then one block per band of rt_render, then the sticky overflow word.
  unsigned long long* d_counters = nullptr;
  hipEvent_t ev_ring[RT_RING] = {};
  bool ring_used[RT_RING] = {};
  unsigned ring_next = 0;
  unsigned long long* d_sticky = nullptr;
  unsigned char* static_scratch = nullptr;  // this device's copy of `scratch` (hipGetSymbolAddress), never the symbol's name
  uint32_t n_cus = 256;
  // frame buffers of rt_render, kept between calls (grow only)
  void* d_frame[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // rgb, mask, object id, hit t, 8-bit rgb
  size_t frame_bytes[5] = {0, 0, 0, 0, 0};
  // Side lists: room in device memory for the n1/n2 intersection lists of rays that pass RT_XS_CAP entries (the
  // reference's list is unbounded, intersections.rs:66-73).  None until a launch on this context overflows; then
  // side_cap entries per list and resident lane (grow_side), four times as many after each further overflow.  While it
  // exists every render goes through the persistent kernels (a lane there owns a fixed region of it), one launch at a
  // time (ev_side orders them).
  char* d_side = nullptr;
  uint32_t side_cap = 0;
  hipEvent_t ev_side = nullptr;
  bool side_used = false;
  std::string error;
};
struct RtDeviceScene {
  Scene sc{};
  std::vector<void*> allocs;
  RtContext* owner = nullptr;
  size_t aux_nodes = 0, aux_tris = 0;
  // the world passes a capacity of the ordinary kernels (rt_device.h Lim) but not of the slow-path ones: it renders
  // through those (need_big gives the context side lists, which is what selects them)
  bool needs_big = false;
};

namespace {

int fail(RtContext* c, int code, const std::string& msg) {
  if (c) c->error = msg;
  return code;
}
#define HIP_TRY(ctx, expr)                                                                                  \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      return fail(ctx, e_ == hipErrorOutOfMemory ? RT_E_OOM : RT_E_DEVICE,                                  \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                       \
  } while (0)

template <class T>
int upload(RtContext* ctx, RtDeviceScene* ds, const T* src, size_t count, const T** dst) {
  *dst = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
  void* p = nullptr;
  HIP_TRY(ctx, hipMalloc(&p, bytes));
  ds->allocs.push_back(p);
  if (count) HIP_TRY(ctx, hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  *dst = static_cast<const T*>(p);
  return RT_OK;
}

// Reject anything that could make a kernel index out of bounds or overflow a fixed device stack.
int validate_scene(RtContext* ctx, const RtScene* s, bool* needs_big) {
  auto bad = [&](const char* what) { return fail(ctx, RT_E_SCENE, std::string("invalid scene: ") + what); };
  *needs_big = false;
  if (s->abi_version != RT_ABI_VERSION) return fail(ctx, RT_E_ABI, "abi_version mismatch");
  if (s->recursion_limit + 1 > RT_BIG_RAY_STACK) return bad("recursion_limit exceeds the device ray stack (16 levels)");
  if (s->recursion_limit + 1 > RT_MAX_RAY_STACK) *needs_big = true;
  if (s->n_xforms < 1) return bad("xforms[0] (identity) missing");
  if (s->n_loose_shapes > s->n_shapes) return bad("n_loose_shapes > n_shapes");
  {
    // (5.2 M records for dragons.yaml: in pieces on threads; the complaint is the one about the first bad shape)
    const uint32_t n_pieces = s->n_shapes >= (1u << 18) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
    std::vector<std::pair<uint32_t, const char*>> first_bad(n_pieces, {0xFFFFFFFFu, nullptr});
    auto piece = [&](uint32_t k) {
      const uint32_t b = (uint32_t)((uint64_t)s->n_shapes * k / n_pieces), e = (uint32_t)((uint64_t)s->n_shapes * (k + 1) / n_pieces);
      for (uint32_t i = b; i < e; i++) {
        const RtShape& sh = s->shapes[i];
        const uint32_t kind = sh.kind_flags & RT_SHAPE_KIND_MASK;
        const char* what = kind > RT_SMOOTH_TRIANGLE                                 ? "shape kind"
                           : sh.xform >= s->n_xforms                                 ? "shape xform index"
                           : sh.material >= s->n_materials                           ? "shape material index"
                           : (sh.parent >= (int32_t)s->n_nodes || sh.parent < -1)    ? "shape parent index"
                           : (kind >= RT_TRIANGLE && sh.prim >= s->n_tris)           ? "shape prim index"
                                                                                     : nullptr;
        if (what) {
          first_bad[k] = {i, what};
          return;
        }
      }
    };
    std::vector<std::thread> pool;
    for (uint32_t k = 1; k < n_pieces; k++) pool.emplace_back(piece, k);
    piece(0);
    for (auto& t : pool) t.join();
    for (const auto& fb : first_bad)
      if (fb.second) return bad(fb.second);
  }
  std::vector<uint32_t> depth(s->n_nodes, 0);
  for (uint32_t i = 0; i < s->n_nodes; i++) {
    const RtNode& n = s->nodes[i];
    if (n.xform >= s->n_xforms) return bad("node xform index");
    if (n.skip <= i || n.skip > s->n_nodes) return bad("node skip pointer");
    if (n.parent >= (int32_t)i || n.parent < -1) return bad("node parent index");
    if ((uint64_t)n.first_shape + n.n_shapes > s->n_shapes) return bad("node shape range");
    if ((n.flags & RT_NODE_ALL_TRIS) && (uint64_t)n.first_tri + n.n_shapes > s->n_tris) return bad("node tri range");
    if (n.next_xf >= (int32_t)s->n_nodes || n.next_xf < -1) return bad("node next_xf");
    if (n.next_xf >= 0 && (uint32_t)n.next_xf > i) return bad("node next_xf is not an ancestor");
    if (n.next_xf >= 0 && s->nodes[n.next_xf].xform == 0) return bad("node next_xf has identity transform");
    depth[i] = (n.parent >= 0 ? depth[n.parent] : 0) + (n.xform ? 1 : 0);
    if (depth[i] > RT_BIG_XF_DEPTH) return bad("more nested group transforms than RT_BIG_XF_DEPTH (32)");
    if (depth[i] > RT_MAX_XF_DEPTH) *needs_big = true;
  }
  for (uint32_t r = 0; r < s->n_csg_roots; r++) {
    const RtCsgRoot& root = s->csg_roots[r];
    if ((uint64_t)root.first + root.count > s->n_csg_nodes || root.count == 0) return bad("csg root range");
    int sp = 0;
    // the most intersections one ray can have with the leaves of this tree (sphere, cube: 2; plane, triangle: 1;
    // cylinder, cone: walls and caps, 4): the per-ray list of csg_eval (RT_CSG_CAP entries) must hold them all
    uint32_t most = 0;
    auto leaf_hits = [&](int32_t shape) {
      const uint32_t kind = s->shapes[shape].kind_flags & RT_SHAPE_KIND_MASK;
      return kind == RT_CYLINDER || kind == RT_CONE ? 4u : (kind == RT_SPHERE || kind == RT_CUBE ? 2u : 1u);
    };
    for (uint32_t k = root.first; k < root.first + root.count; k++) {
      const RtCsgNode& n = s->csg_nodes[k];
      if (n.op > RT_CSG_DIFFERENCE) return bad("csg op");
      int nseg = 0;
      if (n.flags & RT_CSG_LEFT_IS_SHAPE) {
        if (n.left < 0 || (uint32_t)n.left >= s->n_shapes) return bad("csg left shape");
        most += leaf_hits(n.left);
      } else {
        nseg++;
        if (n.left_lo > n.left_hi || n.left_hi > s->n_shapes) return bad("csg left leaf range");
      }
      if (n.flags & RT_CSG_RIGHT_IS_SHAPE) {
        if (n.right < 0 || (uint32_t)n.right >= s->n_shapes) return bad("csg right shape");
        most += leaf_hits(n.right);
      } else {
        nseg++;
      }
      if (sp < nseg) return bad("csg nodes are not in post-order");
      sp = sp - nseg + 1;
      if (sp > RT_BIG_CSG_DEPTH) return bad("csg tree deeper than RT_BIG_CSG_DEPTH (32)");
      if (sp > RT_CSG_DEPTH) *needs_big = true;
    }
    if (sp != 1) return bad("csg root does not reduce to one list");
    if (most > RT_BIG_CSG_CAP)
      return fail(ctx, RT_E_SCENE, "invalid scene: csg tree " + std::to_string(r) + " can give one ray " + std::to_string(most) +
                                       " intersections, more than RT_BIG_CSG_CAP (" + std::to_string(RT_BIG_CSG_CAP) +
                                       "); rebuild the library with a larger -DRT_BIG_CSG_CAP");
    if (most > RT_CSG_CAP) *needs_big = true;
  }
  for (uint32_t i = 0; i < s->n_materials; i++)
    if (s->materials[i].pattern >= (int32_t)s->n_patterns || s->materials[i].pattern < -1) return bad("material pattern");
  for (uint32_t i = 0; i < s->n_patterns; i++) {
    const RtPattern& p = s->patterns[i];
    if (p.kind > RT_PAT_CUBE_MAP || p.xform >= s->n_xforms) return bad("pattern");
    int nuv = p.kind == RT_PAT_TEXTURE_MAP ? 1 : (p.kind == RT_PAT_CUBE_MAP ? 6 : 0);
    for (int k = 0; k < nuv; k++)
      if (p.uv[k] < 0 || p.uv[k] >= (int32_t)s->n_uvpatterns) return bad("pattern uv index");
  }
  for (uint32_t i = 0; i < s->n_uvpatterns; i++) {
    const RtUvPattern& u = s->uvpatterns[i];
    if (u.kind > RT_UV_IMAGE) return bad("uv pattern kind");
    if (u.kind == RT_UV_IMAGE && (u.image < 0 || u.image >= (int32_t)s->n_images)) return bad("uv image index");
  }
  for (uint32_t i = 0; i < s->n_images; i++) {
    const RtImage& im = s->images[i];
    if (im.width == 0 || im.height == 0 || im.offset + (uint64_t)im.width * im.height * 3 > s->n_texels)
      return bad("image texel range");
  }
  for (uint32_t i = 0; i < s->n_lights; i++)
    if (s->lights[i].kind > RT_LIGHT_AREA) return bad("light kind");
  return RT_OK;
}

PartInfo make_part(const RtCamera* cam, const RtRenderOptions* o) {
  PartInfo p;
  p.tile_h = (o && o->tile_h) ? o->tile_h : 1;
  p.n_parts = (o && o->n_parts) ? o->n_parts : 1;
  p.part = o ? o->part : 0;
  uint32_t rows = 0;
  for (uint32_t r = 0; r < cam->vsize; r++)
    if ((r / p.tile_h) % p.n_parts == p.part) rows++;
  p.rows = rows;
  return p;
}

RenderParams make_params(const RtRenderOptions* o) {
  RenderParams rp;
  rp.jitter_mode = o ? o->jitter_mode : RT_JITTER_CONST;
  rp.jitter_value = o ? o->jitter_value : 0.5f;
  rp.seed = o ? o->seed : 0;
  rp.queue_chunk = 1;
  return rp;
}

int read_stats(RtContext* ctx, hipStream_t stream, RtStats* stats, double total_ms, int n_blocks = 1,
               const unsigned long long* d_first = nullptr) {
  unsigned long long h[RT_N_COUNTERS], hb[RT_MAX_BANDS * RT_N_COUNTERS];
  if (n_blocks < 1 || n_blocks > RT_MAX_BANDS) return fail(ctx, RT_E_ABI, "bad counter block count");
  HIP_TRY(ctx, hipMemcpyAsync(hb, d_first ? d_first : ctx->d_counters, (size_t)n_blocks * sizeof(h), hipMemcpyDeviceToHost,
                              stream));
  HIP_TRY(ctx, hipStreamSynchronize(stream));
  for (int k = 0; k < RT_N_COUNTERS; k++) {
    h[k] = 0;
    for (int b = 0; b < n_blocks; b++) h[k] += hb[b * RT_N_COUNTERS + k];
  }
  HIP_TRY(ctx, hipStreamSynchronize(stream));
  float ms = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  stats->primary = h[0];
  stats->shadow = h[1];
  stats->reflect = h[2];
  stats->refract = h[3];
  stats->overflow = h[4];
  stats->node_visits = h[5];
  stats->tri_tests = h[6];
  stats->shape_tests = h[7];
  stats->shaded_hits = h[8];
  stats->node_xforms = h[9];
  stats->aux_tests = h[10];
  stats->rays_reference = h[24];
#ifdef RT_PHASE_TIMERS
  {
    const char* nm[4] = {"aux node", "aux leaf", "walk visit", "walk shape"};
    for (int k = 0; k < 4; k++)
      fprintf(stderr, "%s steps: lanes %llu waves %llu utilisation %.3f\n", nm[k], h[16 + 2 * k], h[17 + 2 * k],
              h[17 + 2 * k] ? (double)h[16 + 2 * k] / (64.0 * h[17 + 2 * k]) : 0.0);
  }
  if (h[28]) {
    const double t0 = (double)(~h[29]), t_last = (double)h[31], t_mean = (double)h[30] / (double)h[28];
    fprintf(stderr, "wave ends (100 MHz ticks): launch %.0f, mean end %.0f, last end %.0f => %.1f %% of wave-time idle at the end\n",
            t_last - t0, t_mean - t0, t_last - t0, 100.0 * (t_last - t_mean) / (t_last - t0));
  }
  {
    fprintf(stderr, "draws by duration (log2 of 10 ns ticks: count):");
    for (int k = 0; k < 28; k++)
      if (h[32 + k]) fprintf(stderr, " %d:%llu", k, h[32 + k]);
    fprintf(stderr, "; longest %.1f us, first item %llu\n", (double)(h[60] >> 32) * 0.01, h[60] & 0xFFFFFFFFull);
  }
  fprintf(stderr, "phase clocks (wave sums): pixel %llu  world_intersect %llu  aux %llu  shapes %llu  refill %llu (queue draw %llu)  node load %llu; job-loop iterations %llu, walk iterations %llu\n", h[14], h[11], h[12], h[13], h[25], h[27], h[10], h[15], h[26]);
#endif
  stats->kernel_ms = ms;
  stats->total_ms = total_ms;
  return RT_OK;
}

// The counter block of the next asynchronous launch on `stream` (see RtContext::d_counters).
int ring_acquire(RtContext* ctx, hipStream_t stream, unsigned* slot_out, unsigned long long** block_out) {
  const unsigned slot = ctx->ring_next++ % RT_RING;
  if (ctx->ring_used[slot]) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_ring[slot], 0));
  *slot_out = slot;
  *block_out = ctx->d_counters + (size_t)slot * RT_N_COUNTERS;
  return RT_OK;
}
int ring_release(RtContext* ctx, hipStream_t stream, unsigned slot) {
  HIP_TRY(ctx, hipEventRecord(ctx->ev_ring[slot], stream));
  ctx->ring_used[slot] = true;
  return RT_OK;
}
// After `stream` has been synchronised: the number of list overflows of every launch on this context since the last
// check (the context's sticky word), which is cleared.  An overflow in an asynchronous call (no stats) is therefore
// reported by the next call that synchronises, never lost.
int take_sticky(RtContext* ctx, hipStream_t stream, unsigned long long* v) {
  *v = 0;
  HIP_TRY(ctx, hipMemcpyAsync(v, ctx->d_sticky, sizeof(*v), hipMemcpyDeviceToHost, stream));
  HIP_TRY(ctx, hipStreamSynchronize(stream));
  if (!*v) return RT_OK;
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_sticky, 0, sizeof(*v), stream));
  HIP_TRY(ctx, hipStreamSynchronize(stream));
  return RT_OK;
}

#ifndef RT_SIDE_FIRST
#define RT_SIDE_FIRST 512u  // entries per list and lane of the first side-list allocation
#endif
#ifndef RT_SIDE_MAX
#define RT_SIDE_MAX 8192u   // the largest: 12 B x 8192 x 196608 resident lanes = 19 GB of the 288
#endif
size_t side_lanes(const RtContext* ctx) { return (size_t)ctx->n_cus * RT_BLOCKS_PER_CU * RT_BLOCK; }
// Gives the context side lists, or larger ones.  Synchronises the device: no launch may be using the old ones.
int grow_side(RtContext* ctx) {
  // RT_SIDE_MAX in the environment lowers (or raises) the limit: a host that would rather fail than give 19 GB away
  uint32_t most = RT_SIDE_MAX;
  if (const char* e = std::getenv("RT_SIDE_MAX")) most = (uint32_t)std::strtoul(e, nullptr, 10);
  if (ctx->side_cap >= most)
    return fail(ctx, RT_E_OVERFLOW, "a ray's intersection list passed " + std::to_string(std::max(most, (uint32_t)RT_XS_CAP)) +
                                    " entries, the largest list this library keeps (RT_SIDE_MAX): the image is invalid");
  HIP_TRY(ctx, hipDeviceSynchronize());
  const uint32_t cap = std::min(most, ctx->side_cap ? ctx->side_cap * 4u : RT_SIDE_FIRST);
  if (ctx->d_side) (void)hipFree(ctx->d_side);
  ctx->d_side = nullptr;
  ctx->side_cap = 0;
  if (!ctx->ev_side) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
  const size_t bytes = side_lanes(ctx) * SIDE_LISTS * 4u * cap;  // 12 B per entry: t, shape, container
  hipError_t e = hipMalloc((void**)&ctx->d_side, bytes);
  if (e != hipSuccess) {
    ctx->d_side = nullptr;
    (void)hipGetLastError();
    return fail(ctx, RT_E_OVERFLOW, "a ray's intersection list overflowed and the side lists (" + std::to_string(bytes >> 20) +
                                    " MiB) could not be allocated: " + hipGetErrorString(e));
  }
  ctx->side_cap = cap;
  ctx->side_used = false;
  return RT_OK;
}
// What a synchronous call does about overflows once its stream is idle.  `own` = the overflows its own launch(es)
// counted.  Its own overflow is cured by giving the context (larger) side lists and running the call again (*retry);
// overflows left behind by EARLIER asynchronous calls cannot be cured here -- those frames are gone -- and are reported.
int settle_overflow(RtContext* ctx, hipStream_t stream, unsigned long long own, bool may_grow, bool* retry,
                    unsigned long long* earlier) {
  *retry = false;
  unsigned long long v = 0;
  int rc = take_sticky(ctx, stream, &v);
  if (rc != RT_OK) return rc;
  if (v > own) *earlier += v - own;
  if (own) {
    if (!may_grow)
      return fail(ctx, RT_E_OVERFLOW, "per-ray intersection list overflow (" + std::to_string(own) + " rays) and this entry "
                                      "point may not allocate side lists: use rt_color_at");
    rc = grow_side(ctx);
    if (rc != RT_OK) return rc;
    *retry = true;
    return RT_OK;
  }
  if (*earlier)
    return fail(ctx, RT_E_OVERFLOW, "per-ray intersection list overflow (" + std::to_string(*earlier) + " rays) in an earlier "
                                    "asynchronous call on this context: those images are invalid (render them again: the "
                                    "context now holds side lists of " + std::to_string(ctx->side_cap) + " entries)");
  return RT_OK;
}
// Launches that use the side lists run one after another, whatever their streams.
int side_enter(RtContext* ctx, hipStream_t stream) {
  if (ctx->side_cap && ctx->side_used) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ev_side, 0));
  return RT_OK;
}
int side_leave(RtContext* ctx, hipStream_t stream) {
  if (ctx->side_cap) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_side, stream));
    ctx->side_used = true;
  }
  return RT_OK;
}
void attach_side(const RtContext* ctx, Scene& sc) {
  sc.side = ctx->d_side;
  sc.side_cap = ctx->side_cap;
}
// A world (or a rt_color_at depth) beyond a capacity of the ordinary kernels renders through the slow-path ones, which a
// context selects by holding side lists: give it some.  may_allocate false (rt_color_at_static): it must have them already.
int need_big(RtContext* ctx, bool needed, bool may_allocate) {
  if (!needed || ctx->side_cap) return RT_OK;
  if (!may_allocate)
    return fail(ctx, RT_E_LIMIT, "this world (or depth) needs the slow-path kernels, whose lists this entry point may not "
                                 "allocate: render it once through another entry point of the same context first");
  return grow_side(ctx);
}

}  // namespace

// The job loop with everything in it is ~14 000 instructions, more than the instruction cache of a pair of CUs holds,
// and what a scene does not use still costs: registers, scratch, fetches around the dead code.  The sample scenes of
// the reference use one feature each (dragons: groups; cover: refraction; csg: csg; checkered_*, basic_scene: patterns),
// so there is one instantiation per single feature and the complete one for every mixture.  RT_NO_LEAN=1 in the
// environment takes the complete one always (A/B timing, tests).
static uint32_t scene_features(const Scene& sc) {
  uint32_t f = 0;
  if (sc.n_csg_roots) f |= RT_F_CSG;
  if (!(sc.props & RT_SCENE_RI_ALL_ONE)) f |= RT_F_COLLECT;
  if (sc.props & RT_SCENE_HAS_PATTERN) f |= RT_F_PATTERN;
  if (sc.props & RT_SCENE_OTHER_LIGHT) f |= RT_F_AREA;
  if (sc.n_nodes) f |= RT_F_GROUPS;
  return f;
}
typedef void (*JobsKernel)(Scene, RtCamera, RenderParams, PartInfo, PixelOut, unsigned long long*);
struct JobsPick {
  JobsKernel kern;
  uint32_t blocks_per_cu;  // resident 256-thread blocks per CU = the kernel's waves per SIMD
};
template <bool CULL>
static JobsPick jobs_kernel(uint32_t need, uint64_t pixels) {
  const bool complete = std::getenv("RT_NO_LEAN") != nullptr;
  // (RT_LEAN_MIN_PIXELS in the environment moves the threshold: 0 sends every frame through the four-wave kernel, for tests)
  static const uint64_t min_pixels = [] {
    const char* e = std::getenv("RT_LEAN_MIN_PIXELS");
    return e ? std::strtoull(e, nullptr, 10) : (uint64_t)RT_LEAN_MIN_PIXELS;
  }();
  const bool big = pixels >= min_pixels;
  if (!complete) {
    if (!(need & ~RT_F_PATTERN)) return {k_render_jobs<false, CULL, false, RT_F_PATTERN>, RT_FLAT_WAVES_PER_EU};
    if (!(need & ~RT_F_COLLECT)) return {k_render_jobs<false, CULL, false, RT_F_COLLECT>, RT_FLAT_WAVES_PER_EU};
    if (!(need & ~RT_F_CSG)) return {k_render_jobs<false, CULL, false, RT_F_CSG>, RT_FLAT_WAVES_PER_EU};
    if (!(need & ~RT_F_GROUPS))
      return big ? JobsPick{k_render_jobs<false, CULL, false, RT_F_GROUPS | RT_F_WAVES4>, RT_LEAN_WAVES_PER_EU}
                 : JobsPick{k_render_jobs<false, CULL, false, RT_F_GROUPS>, RT_BLOCKS_PER_CU};
    // mixtures: everything but the groups (a patterned floor under glass spheres), or everything
    if (!(need & RT_F_GROUPS)) return {k_render_jobs<false, CULL, false, RT_F_ALL & ~RT_F_GROUPS>, RT_FLAT_WAVES_PER_EU};
  }
  // (the complete kernel gains nothing from a fourth wave: 80 registers spilled, dragons 2.39 ms either way)
  return {k_render_jobs<false, CULL>, RT_BLOCKS_PER_CU};
}

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_context_create(int device, RtContext** out) {
  if (!out) return RT_E_ABI;
  *out = nullptr;
  RtContext* c = new RtContext();
  c->device = device;
  auto bail = [&](hipError_t e, const char* what) {
    std::fprintf(stderr, "rt_context_create: %s: %s\n", what, hipGetErrorString(e));
    rt_context_destroy(c);
    return RT_E_DEVICE;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  {
    int cus = 0;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess)
      return bail(e, "hipDeviceGetAttribute");
    c->n_cus = cus > 0 ? (uint32_t)cus : 256u;
  }
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
  {
    const size_t words = (size_t)(RT_RING + RT_MAX_BANDS) * RT_N_COUNTERS + 16;
    if ((e = hipMalloc((void**)&c->d_counters, words * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(c->d_counters, 0, words * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset");
    c->d_sticky = c->d_counters + (size_t)(RT_RING + RT_MAX_BANDS) * RT_N_COUNTERS;
  }
  for (int b = 0; b < RT_RING; b++)
    if ((e = hipEventCreateWithFlags(&c->ev_ring[b], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
  // each GPU holds its own copy of the static scratch array at its own address: look it up with this device current
  if ((e = hipGetSymbolAddress((void**)&c->static_scratch, HIP_SYMBOL(scratch))) != hipSuccess || !c->static_scratch)
    return bail(e, "hipGetSymbolAddress(scratch)");
  if ((e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  for (int b = 0; b < RT_MAX_BANDS; b++)
    if ((e = hipEventCreateWithFlags(&c->ev_band[b], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
  *out = c;
  return RT_OK;
}

void rt_context_destroy(RtContext* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->d_side) (void)hipFree(c->d_side);
  if (c->ev_side) (void)hipEventDestroy(c->ev_side);
  for (void* p : c->d_frame)
    if (p) (void)hipFree(p);
  for (hipEvent_t ev : c->ev_band)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_ring)
    if (ev) (void)hipEventDestroy(ev);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* rt_last_error(const RtContext* c) { return c ? c->error.c_str() : "null context"; }

int rt_scene_create(RtContext* ctx, const RtScene* s, RtDeviceScene** out) {
  if (!ctx || !s || !out) return RT_E_ABI;
  *out = nullptr;
  // RT_TIMING=1 in the environment: stage times of the scene set-up on stderr
  const bool timing = std::getenv("RT_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "rt_scene_create: %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  bool needs_big = false;
  int rc = validate_scene(ctx, s, &needs_big);
  stage("validate");
  if (rc != RT_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  RtDeviceScene* ds = new RtDeviceScene();
  ds->owner = ctx;
  ds->needs_big = needs_big;
  Scene& d = ds->sc;
  const RtTriGeom* tg = nullptr;
  const RtTriNormals* tn = nullptr;
#define UP(field, src, count)                                 \
  if ((rc = upload(ctx, ds, src, count, &field)) != RT_OK) {  \
    rt_scene_destroy(ctx, ds);                                \
    return rc;                                                \
  }
  UP(d.xforms, s->xforms, (size_t)s->n_xforms)
  // the library's own acceleration data -- auxiliary BVHs under triangle-heavy groups and the cull records of transformed
  // groups (rt_accel.h) -- depend on the caller's arrays only: they are built on threads of their own while this one
  // prepares and uploads the shapes (0.22 + 0.04 s of a 0.34 s call for dragons.yaml)
  // (the big trees themselves on the device, rt_accel_device.h, on a stream of the builder's own; RT_HOST_ACCEL_BUILD=1
  // in the environment keeps them on host threads: the two give the same trees)
  const int dev_for_trees = ctx->device;
  const bool host_trees = std::getenv("RT_HOST_ACCEL_BUILD") != nullptr;
  std::future<rtaccel::Result> acc_job = std::async(std::launch::async, [s, dev_for_trees, host_trees] {
    if (host_trees) return rtaccel::build(s);
    return rtaccel::build(s, [dev_for_trees](const RtScene* sc, uint32_t root, rtaccel::TreeBuilder& tb) {
      return rtaccel_dev::build_tree(dev_for_trees, sc, root, tb);
    });
  });
  std::future<rtaccel::NodeCull> cull_job = std::async(std::launch::async, [s] { return rtaccel::build_node_cull(s); });
  std::vector<std::vector<uint32_t>> piece_noshadow;  // shapes with shadow() == false, per piece of the pass below
  {
    // device copy of the shapes: bounded analytic shapes get a cull record (rt_accel.h), its index in `prim`.
    // 5.2 M records for dragons.yaml: the passes over all of them run in pieces on threads; whatever decides an index
    // (cull records, duplicate links) is put together in piece order, so the result does not depend on the threads.
    std::unique_ptr<RtShape[]> shapes_mem(new RtShape[std::max<size_t>(s->n_shapes, 1)]);  // (not zeroed first: every piece fills its own)
    RtShape* const shapes = shapes_mem.get();
    const uint32_t n_pieces = s->n_shapes >= (1u << 18) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
    auto in_pieces = [&](const std::function<void(uint32_t, uint32_t, uint32_t)>& f) {
      std::vector<std::thread> pool;
      for (uint32_t k = 1; k < n_pieces; k++)
        pool.emplace_back(f, k, (uint32_t)((uint64_t)s->n_shapes * k / n_pieces), (uint32_t)((uint64_t)s->n_shapes * (k + 1) / n_pieces));
      f(0, 0, (uint32_t)((uint64_t)s->n_shapes / n_pieces));
      for (auto& t : pool) t.join();
    };
    std::vector<float> cull;
    // worth its ~25 operations per test only where a ray meets many shapes in a row: the loose shapes of the world or
    // the own shapes of one group, when there are at least RT_CULL_MIN_SHAPES of them
    {
      std::vector<std::vector<uint32_t>> candidates(n_pieces);  // cubes and spheres with enough siblings
      std::vector<uint32_t> piece_max(n_pieces, 0);
      piece_noshadow.resize(n_pieces);
      in_pieces([&](uint32_t piece, uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) {
          shapes[i] = s->shapes[i];
          const uint32_t kind = shapes[i].kind_flags & RT_SHAPE_KIND_MASK;
          // next_copy: itself, until a later duplicate is found (in a slot the kind does not use)
          memcpy(kind == RT_CYLINDER || kind == RT_CONE ? (void*)&shapes[i].prim : (void*)&shapes[i].mn, &i, sizeof(uint32_t));
          if (kind == RT_CUBE || kind == RT_SPHERE) {
            const int32_t par = shapes[i].parent;
            const uint32_t siblings = par >= 0 ? s->nodes[par].n_shapes : s->n_loose_shapes;
            if (siblings >= RT_CULL_MIN_SHAPES) candidates[piece].push_back(i);
          }
          if (shapes[i].vclass != RT_VCLASS_NEVER) piece_max[piece] = std::max(piece_max[piece], shapes[i].vclass);
          if (!(shapes[i].kind_flags & RT_SHAPE_SHADOW)) piece_noshadow[piece].push_back(i);
        }
      });
      for (const std::vector<uint32_t>& list : candidates)
        for (uint32_t i : list) {
          float rec[12];
          if (rtaccel::shape_cull_record(s, shapes[i], rec)) {
            shapes[i].prim = (uint32_t)(cull.size() / 12);
            shapes[i].kind_flags |= RT_SHAPE_CULLABLE;
            cull.insert(cull.end(), rec, rec + 12);
          }
        }
      // Which of several value-identical shapes is reported as hit (rt_device.h end_of_copy_run). A mesh that lists
      // a triangle twice gives two entries equal in (t, Shape, u, v), and of such neighbours the reference's
      // Intersections keeps the LATER one (BTreeMap bulk build), whereas ties between different shapes
      // go to the earlier. The device copy of a shape holds the NEXT shape of its value class, wherever it
      // sits, in a slot its kind does not use (itself: none); whether that one is an entry at the same t for a given ray --
      // always, under the same transformed ancestor; hardly ever, under another -- is decided on the device.
      // only classes that occur more than once can have duplicates: count first (one pass on threads); only then the
      // links, in one pass from the last shape to the first (per class: the index met last, i.e. the next one)
      uint32_t max_class = 0;
      for (uint32_t m : piece_max) max_class = std::max(max_class, m);
      std::vector<uint8_t> seen((size_t)max_class + 1, 0);  // per class: 1 = seen once, 2 and more = more than once
      std::vector<uint8_t> piece_dup(n_pieces, 0);
      in_pieces([&](uint32_t piece, uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++)
          // a shape that holds a NaN equals no shape, itself included: it is nobody's duplicate
          // (a count stops being raised once somebody sees 2: with one thread per piece it stays below 2 + their number)
          if (shapes[i].vclass != RT_VCLASS_NEVER && __atomic_load_n(&seen[shapes[i].vclass], __ATOMIC_RELAXED) < 2 &&
              __atomic_fetch_add(&seen[shapes[i].vclass], (uint8_t)1, __ATOMIC_RELAXED) >= 1)
            piece_dup[piece] = 1;
      });
      if (std::find(piece_dup.begin(), piece_dup.end(), (uint8_t)1) != piece_dup.end()) {
        d.props |= RT_SCENE_HAS_COPIES;
        std::vector<uint32_t> later((size_t)max_class + 1, 0xFFFFFFFFu);
        for (uint32_t i = s->n_shapes; i-- > 0;) {
          if (shapes[i].vclass == RT_VCLASS_NEVER || seen[shapes[i].vclass] < 2) continue;
          uint32_t& l = later[shapes[i].vclass];
          const uint32_t kind = shapes[i].kind_flags & RT_SHAPE_KIND_MASK;
          if (l != 0xFFFFFFFFu)
            memcpy(kind == RT_CYLINDER || kind == RT_CONE ? (void*)&shapes[i].prim : (void*)&shapes[i].mn, &l, sizeof(uint32_t));
          l = i;
        }
      }
    }
    stage("cull records, duplicate links");
    const float* dc = nullptr;
    UP(d.shapes, (const RtShape*)shapes, (size_t)s->n_shapes)
    UP(dc, cull.data(), cull.size())
    d.shape_cull = reinterpret_cast<const float4*>(dc);
    if (!cull.empty()) d.props |= RT_SCENE_SHAPE_CULL;
    hipError_t e = hipStreamSynchronize(ctx->stream);  // the copies above read these temporaries
    if (e != hipSuccess) {
      rt_scene_destroy(ctx, ds);
      return fail(ctx, RT_E_DEVICE, std::string("scene upload: ") + hipGetErrorString(e));
    }
  }
  stage("upload shapes");
  UP(tg, s->tri_geom, (size_t)s->n_tris)
  UP(tn, s->tri_normals, (size_t)s->n_tris)
  stage("upload triangles");
  {
    // the library's own acceleration data: auxiliary BVHs under triangle-heavy groups (rt_accel.h);
    // the device copy of a node carries its aux root in the slot of n_children (unused on the device)
    rtaccel::Result acc = acc_job.get();
    stage("auxiliary trees (wait)");
    std::vector<RtNode> nodes(s->nodes, s->nodes + s->n_nodes);
    // groups that hold a shape with shadow() == false somewhere below them (any-shadow search, rt_device.h)
    for (const std::vector<uint32_t>& list : piece_noshadow)  // (noted by the pass over the shapes above)
      for (uint32_t i : list) {
        d.props |= RT_SCENE_HAS_NOSHADOW;
        for (int32_t g = s->shapes[i].parent; g >= 0 && !(nodes[g].flags & RT_NODE_HAS_NOSHADOW); g = nodes[g].parent)
          nodes[g].flags |= RT_NODE_HAS_NOSHADOW;
      }
    bool ri_one = true;
    for (uint32_t i = 0; i < s->n_materials; i++) ri_one = ri_one && s->materials[i].refractive_index == 1.0f;
    if (ri_one) d.props |= RT_SCENE_RI_ALL_ONE;
    for (uint32_t i = 0; i < s->n_materials; i++)
      if (s->materials[i].pattern >= 0) d.props |= RT_SCENE_HAS_PATTERN;
    for (uint32_t i = 0; i < s->n_lights; i++)
      if (s->lights[i].kind != RT_LIGHT_POINT) d.props |= RT_SCENE_OTHER_LIGHT;
    for (uint32_t i = 0; i < s->n_lights; i++)
      if (s->lights[i].kind == RT_LIGHT_AREA && s->lights[i].usteps * s->lights[i].vsteps > 1) d.props |= RT_SCENE_AREA_LIGHT;
    if ((d.props & RT_SCENE_HAS_NOSHADOW) && !acc.nodes.empty()) d.props |= RT_SCENE_SHADOW_TWO_PASS;
    for (uint32_t i = 0; i < s->n_nodes; i++) {
      if (acc.root_of[i] != rtaccel::kNone) {
        nodes[i].n_children = acc.root_of[i];
        nodes[i].flags |= RT_NODE_AUX;
      } else {
        nodes[i].n_children = acc.gate_next[i];
      }
    }
    // cull records of transformed groups: the record index rides in the idrun slot (0 for every transformed node)
    stage("node flags");
    rtaccel::NodeCull nc = cull_job.get();
    stage("subtree cull records (wait)");
    for (uint32_t i = 0; i < s->n_nodes; i++)
      if (nc.rec_of[i] != rtaccel::kNone) {
        nodes[i].idrun = nc.rec_of[i];
        nodes[i].flags |= RT_NODE_CULLREC;
      }
    const float* dnc = nullptr;
    UP(dnc, nc.recs.data(), nc.recs.size())
    d.node_cull = reinterpret_cast<const float4*>(dnc);
    const rtaccel::Node4* an = nullptr;
    const rtaccel::Tri* at = nullptr;
    UP(d.nodes, nodes.data(), nodes.size())
    UP(an, acc.nodes.data(), acc.nodes.size())
    UP(at, acc.tris.data(), acc.tris.size())
    d.aux_nodes = reinterpret_cast<const float4*>(an);
    d.aux_tris = reinterpret_cast<const float4*>(at);
    ds->aux_nodes = acc.nodes.size();
    ds->aux_tris = acc.tris.size();
    // uploads above are asynchronous copies from these temporaries
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      rt_scene_destroy(ctx, ds);
      return fail(ctx, RT_E_DEVICE, std::string("scene upload: ") + hipGetErrorString(e));
    }
  }
  stage("upload nodes and trees");
  UP(d.csg_nodes, s->csg_nodes, (size_t)s->n_csg_nodes)
  UP(d.csg_roots, s->csg_roots, (size_t)s->n_csg_roots)
  UP(d.materials, s->materials, (size_t)s->n_materials)
  UP(d.patterns, s->patterns, (size_t)s->n_patterns)
  UP(d.uvpatterns, s->uvpatterns, (size_t)s->n_uvpatterns)
  UP(d.images, s->images, (size_t)s->n_images)
  UP(d.texels, s->texels, (size_t)s->n_texels)
  UP(d.lights, s->lights, (size_t)s->n_lights)
#undef UP
  d.tri_geom = reinterpret_cast<const float4*>(tg);
  d.tri_normals = reinterpret_cast<const float4*>(tn);
  d.n_shapes = s->n_shapes;
  d.n_nodes = s->n_nodes;
  d.n_csg_roots = s->n_csg_roots;
  d.n_lights = s->n_lights;
  d.n_loose = s->n_loose_shapes;
  d.recursion_limit = s->recursion_limit;
  d.flags = 0;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    rt_scene_destroy(ctx, ds);
    return fail(ctx, RT_E_DEVICE, std::string("scene upload: ") + hipGetErrorString(e));
  }
  *out = ds;
  return RT_OK;
}

void rt_scene_destroy(RtContext* ctx, RtDeviceScene* ds) {
  if (!ds) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  for (void* p : ds->allocs) (void)hipFree(p);
  delete ds;
}

static int accel_view(rtaccel::Result* r, RtAccelView* out);
int rt_accel_build(const RtScene* s, RtAccelView* out) {
  if (!s || !out) return RT_E_ABI;
  return accel_view(new rtaccel::Result(rtaccel::build(s)), out);
}
int rt_accel_build_device(RtContext* ctx, const RtScene* s, RtAccelView* out) {
  if (!ctx || !s || !out) return RT_E_ABI;
  bool needs_big = false;
  int rc = validate_scene(ctx, s, &needs_big);
  if (rc != RT_OK) return rc;
  const int dev = ctx->device;
  uint32_t declined = 0;
  rtaccel::Result* r = new rtaccel::Result(rtaccel::build(s, [dev, &declined](const RtScene* sc, uint32_t root, rtaccel::TreeBuilder& tb) {
    const bool ok = rtaccel_dev::build_tree(dev, sc, root, tb);
    if (!ok) declined++;
    return ok;
  }));
  (void)hipSetDevice(ctx->device);
  rc = accel_view(r, out);
  // (a tree the device builder declined was built on host threads: the view is complete either way)
  if (declined) ctx->error = std::to_string(declined) + " tree(s) built on the host";
  return rc;
}
static int accel_view(rtaccel::Result* r, RtAccelView* out) {
  out->n_nodes = r->nodes.size();
  out->n_tris = r->tris.size();
  out->n_roots = r->root_of.size();
  out->nodes = r->nodes.data();
  out->tris = r->tris.data();
  out->root_of = r->root_of.data();
  out->gate_next = r->gate_next.data();
  out->owner = r;
  return RT_OK;
}
void rt_accel_free(RtAccelView* v) {
  if (v && v->owner) delete static_cast<rtaccel::Result*>(v->owner);
  if (v) *v = RtAccelView{};
}

uint32_t rt_part_rows(const RtCamera* cam, const RtRenderOptions* o) { return cam ? make_part(cam, o).rows : 0; }

// One kernel launch: the rows of `part` into d_out_rgb / d_aux, counters (and the work queue) in `d_counters`.
#define RT_LIGHT_SCENE_SHAPES 16u
#ifndef RT_LIGHT_SCENE_DRAWS
#define RT_LIGHT_SCENE_DRAWS 2u
#endif
#ifndef RT_LIGHT_SCENE_CHUNK
#define RT_LIGHT_SCENE_CHUNK 4u
#endif
static int launch_part(RtContext* ctx, const Scene& sc, const RtCamera* cam, const RenderParams& rp, const PartInfo& part,
                       float* d_out_rgb, const RtAux* d_aux, hipStream_t stream, unsigned long long* d_counters) {
  HIP_TRY(ctx, hipMemsetAsync(d_counters, 0, RT_N_COUNTERS * sizeof(unsigned long long), stream));
  if (part.rows == 0) return RT_OK;
  int rc_side = side_enter(ctx, stream);
  if (rc_side != RT_OK) return rc_side;
  // both integrators give the same bits; the direct one wins where an area light's sample loop dominates
  // (a context with side lists renders through the persistent kernel only: its lanes own fixed regions of them)
  const bool direct = sc.n_nodes == 0 && (sc.props & RT_SCENE_AREA_LIGHT) && !sc.side_cap;
  const bool count = (sc.flags & RT_RENDER_COUNT_WORK) != 0;
  PixelOut po{d_out_rgb, d_aux ? d_aux->hit_mask : nullptr, d_aux ? d_aux->object_id : nullptr,
              d_aux ? d_aux->hit_t : nullptr};
  if (direct) {
    const uint32_t tile = RT_BLOCK == 256 ? 16u : 8u;
    dim3 grid((cam->hsize + tile - 1) / tile, (part.rows + tile - 1) / tile);
    const bool hoist = sc.n_csg_roots == 0 && sc.n_loose <= RT_HOIST_SHAPES && !std::getenv("RT_NO_HOIST");
    auto kern = count ? (hoist ? k_render_direct<true, true> : k_render_direct<true, false>)
                      : (hoist ? k_render_direct<false, true> : k_render_direct<false, false>);
    hipLaunchKernelGGL(kern, grid, dim3(RT_BLOCK), 0, stream, sc, *cam, rp, part, po, d_counters);
  } else {
    // persistent waves: enough blocks to fill the device at four waves per SIMD, never more than the work
    const uint64_t items = (uint64_t)((cam->hsize + 7u) / 8u) * ((part.rows + 7u) / 8u) * 64u;
    // item ids run over whole 128x128 supertiles (PixelSource) and must fit 32 bits
    const uint64_t super_items = (uint64_t)((cam->hsize + 8u * RT_SUPER - 1u) / (8u * RT_SUPER)) * ((part.rows + 8u * RT_SUPER - 1u) / (8u * RT_SUPER)) * (RT_SUPER * RT_SUPER * 64u);
    if (super_items >= 0xFFFFFFFFull) return fail(ctx, RT_E_ABI, "image too large");
    // ... and a lane keeps its pixel as (row << 16 | column) (PixelSource::start)
    if (cam->hsize > 0xFFFFu || part.rows > 0xFFFFu) return fail(ctx, RT_E_ABI, "image too large (65535 pixels a side)");
    const bool cull = (sc.props & RT_SCENE_SHAPE_CULL) != 0;  // the scene has cull records (rt_scene_create)
    JobsPick pick = count ? (cull ? JobsPick{k_render_jobs<true, true>, RT_BLOCKS_PER_CU} : JobsPick{k_render_jobs<true, false>, RT_BLOCKS_PER_CU})
                          : (cull ? jobs_kernel<true>(scene_features(sc), (uint64_t)cam->hsize * part.rows)
                                  : jobs_kernel<false>(scene_features(sc), (uint64_t)cam->hsize * part.rows));
    // a context with side lists: the one instantiation that keeps the n1/n2 lists there (it counts work and tests cull
    // records whether asked to or not: it is the slow path, and one kernel instead of four keeps the build short)
    if (sc.side_cap) pick = JobsPick{k_render_jobs<true, true, true>, RT_BLOCKS_PER_CU};
    const auto kern = pick.kern;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((items + RT_BLOCK - 1) / RT_BLOCK, (uint64_t)ctx->n_cus * pick.blocks_per_cu);
    // A scene of a few shapes has cheap tiles throughout, and its frame time is the queue's atomics: draw up to
    // RT_LIGHT_SCENE_CHUNK tiles at a time, as long as every wave still gets RT_LIGHT_SCENE_DRAWS draws
    // (integrate's refill).
    RenderParams rq = rp;
    const uint64_t waves = (uint64_t)blocks * (RT_BLOCK / 64u);
    if (sc.n_nodes == 0 && sc.n_shapes <= RT_LIGHT_SCENE_SHAPES)
      rq.queue_chunk = (uint32_t)std::clamp<uint64_t>(items / 64u / (waves * RT_LIGHT_SCENE_DRAWS), 1u, RT_LIGHT_SCENE_CHUNK);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(RT_BLOCK), 0, stream, sc, *cam, rq, part, po, d_counters);
  }
  if ((sc.props & RT_SCENE_HAS_COPIES) && po.oid) {
    const size_t n = (size_t)part.rows * cam->hsize;
    hipLaunchKernelGGL(k_copies_pixels, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, sc, *cam, part, po.oid);
  }
  HIP_TRY(ctx, hipGetLastError());
  return side_leave(ctx, stream);
}

int rt_render_device(RtContext* ctx, const RtDeviceScene* ds, const RtCamera* cam, const RtRenderOptions* opts,
                     float* d_out_rgb, RtAux* d_aux, void* stream_, RtStats* stats) {
  if (!ctx || !ds || !cam || !d_out_rgb) return RT_E_ABI;
  if (ds->owner != ctx) return fail(ctx, RT_E_ABI, "scene belongs to another context");
  if (cam->hsize == 0 || cam->vsize == 0) return fail(ctx, RT_E_ABI, "empty camera");
  PartInfo part = make_part(cam, opts);
  if (part.part >= part.n_parts) return fail(ctx, RT_E_ABI, "part >= n_parts");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_big = need_big(ctx, ds->needs_big, true);
    if (rc_big != RT_OK) return rc_big;
  }
  hipStream_t stream = stream_ ? (hipStream_t)stream_ : ctx->stream;
  Scene sc = ds->sc;
  sc.flags = opts ? opts->flags : 0;
  sc.sticky = ctx->d_sticky;
  RenderParams rp = make_params(opts);
  unsigned long long earlier = 0;
  while (true) {
    attach_side(ctx, sc);
    unsigned slot = 0;
    unsigned long long* block = nullptr;
    int rc = ring_acquire(ctx, stream, &slot, &block);
    if (rc != RT_OK) return rc;
    if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev0, stream));
    rc = launch_part(ctx, sc, cam, rp, part, d_out_rgb, d_aux, stream, block);
    if (rc != RT_OK) return rc;
    if (stats) HIP_TRY(ctx, hipEventRecord(ctx->ev1, stream));
    rc = ring_release(ctx, stream, slot);
    if (rc != RT_OK) return rc;
    if (!stats) return RT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rc = read_stats(ctx, stream, stats, total, 1, block);
    if (rc != RT_OK) return rc;
    bool retry = false;
    rc = settle_overflow(ctx, stream, stats->overflow, true, &retry, &earlier);
    if (!retry) return rc;  // else: the context has (larger) side lists now; the same launch again
  }
}

int rt_check(RtContext* ctx, void* stream_) {
  if (!ctx) return RT_E_ABI;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (stream_)
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream_));
  else
    HIP_TRY(ctx, hipDeviceSynchronize());  // every stream of this GPU, the context's own included
  unsigned long long v = 0;
  int rc = take_sticky(ctx, ctx->stream, &v);
  if (rc != RT_OK || !v) return rc;
  // the frames are gone, but the next attempt need not fail: give the context (larger) side lists now
  const int grown = grow_side(ctx);
  return fail(ctx, RT_E_OVERFLOW,
              "per-ray intersection list overflow (" + std::to_string(v) + " rays) in asynchronous calls on this context since "
              "the last check: those images are invalid" +
              (grown == RT_OK ? "; render them again, the context now holds side lists of " + std::to_string(ctx->side_cap) +
                                    " entries per ray"
                              : std::string("; ") + ctx->error));
}

static int launch_quantize(RtContext* ctx, const float* d_rgb, uint8_t* d_out, uint64_t n_pixels, uint32_t hsize,
                           const PartInfo& part, uint32_t dither, hipStream_t stream) {
  if (n_pixels == 0) return RT_OK;
  const uint64_t threads = (n_pixels + 3u) / 4u;
  hipLaunchKernelGGL(k_canvas_to_u8, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, stream, d_rgb, d_out, n_pixels, hsize,
                     part, dither);
  HIP_TRY(ctx, hipGetLastError());
  return RT_OK;
}

int rt_canvas_to_u8_device(RtContext* ctx, const RtCamera* cam, const RtRenderOptions* opts, uint32_t dither,
                           const float* d_rgb, uint8_t* d_out_u8, void* stream_) {
  if (!ctx || !cam || !d_rgb || !d_out_u8) return RT_E_ABI;
  if (dither > RT_DITHER_BAYER_COLOR) return fail(ctx, RT_E_ABI, "unknown dither mode");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const PartInfo part = make_part(cam, opts);
  if (part.part >= part.n_parts) return fail(ctx, RT_E_ABI, "part >= n_parts");
  return launch_quantize(ctx, d_rgb, d_out_u8, (uint64_t)part.rows * cam->hsize, cam->hsize, part, dither,
                         stream_ ? (hipStream_t)stream_ : ctx->stream);
}

// Host-memory entry points.  The frame buffers live in the context between calls; a full frame is rendered as up to
// RT_MAX_BANDS bands of rows, each copied back on a second stream while the next one renders (the copy of a 4K
// f32 frame takes longer than its kernel).  out_u8 != NULL: each band is dithered / quantised on the device and the
// 8-bit rows are what is copied (a quarter of the bytes).
static int render_host(RtContext* ctx, const RtDeviceScene* ds, const RtCamera* cam, const RtRenderOptions* opts,
                       float* out_rgb, uint8_t* out_u8, uint32_t dither, RtAux* aux, RtStats* stats) {
  if (!ctx || !ds || !cam || (!out_rgb && !out_u8)) return RT_E_ABI;
  if (ds->owner != ctx) return fail(ctx, RT_E_ABI, "scene belongs to another context");
  if (cam->hsize == 0 || cam->vsize == 0) return fail(ctx, RT_E_ABI, "empty camera");
  if (dither > RT_DITHER_BAYER_COLOR) return fail(ctx, RT_E_ABI, "unknown dither mode");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    int rc_big = need_big(ctx, ds->needs_big, true);
    if (rc_big != RT_OK) return rc_big;
  }
  const PartInfo whole = make_part(cam, opts);
  if (whole.part >= whole.n_parts) return fail(ctx, RT_E_ABI, "part >= n_parts");
  const size_t npix = (size_t)whole.rows * cam->hsize;
  const size_t elem[5] = {3 * sizeof(float), 1, sizeof(int32_t), sizeof(float), 3};
  void* host[5] = {out_rgb, aux ? (void*)aux->hit_mask : nullptr, aux ? (void*)aux->object_id : nullptr,
                   aux ? (void*)aux->hit_t : nullptr, out_u8};
  for (int k = 0; k < 5; k++) {
    if (!host[k] && !(k == 0 && out_u8)) continue;  // the f32 frame is rendered even when only its 8-bit form is wanted
    size_t need = (npix ? npix : 1) * elem[k];
    if (ctx->frame_bytes[k] < need) {
      if (ctx->d_frame[k]) (void)hipFree(ctx->d_frame[k]);
      ctx->d_frame[k] = nullptr;
      ctx->frame_bytes[k] = 0;
      hipError_t e = hipMalloc(&ctx->d_frame[k], need);
      if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_DEVICE, std::string("frame buffer: ") + hipGetErrorString(e));
      ctx->frame_bytes[k] = need;
    }
  }
  // bands: a full frame splits into contiguous bands of rows (a part with tile_h = band height); a part of a
  // multi-GPU frame is rendered in one piece
  PartInfo bands[RT_MAX_BANDS];
  int nb = 1;
  bands[0] = whole;
  if (whole.n_parts == 1 && cam->vsize >= 64) {
    // Every extra launch has a tail of its own, every band's copy overlaps the next band's kernel: how many bands pay
    // depends on how long the copy is.  Measured on dragons 4K (tools/gpu_bands.py, ms per call for 1/2/3/4 bands):
    // f32 pageable 8.5/7.1/6.5/6.4, f32 page-locked 4.5/4.0/3.9/4.2, u8 pageable 4.1/3.8/3.9/4.1, u8 page-locked
    // 3.1/3.3/3.5/3.9.
    bool locked = false;
    {
      hipPointerAttribute_t at;
      const void* dst = out_u8 && !out_rgb ? (const void*)out_u8 : (const void*)out_rgb;
      if (hipPointerGetAttributes(&at, dst) == hipSuccess)
        locked = at.type == hipMemoryTypeHost;
      else
        (void)hipGetLastError();  // an ordinary (pageable) pointer is "invalid" to the runtime
    }
    const bool bytes_only = out_u8 && !out_rgb;
    uint32_t want = bytes_only ? (locked ? 1u : 2u) : (locked ? 2u : (uint32_t)RT_MAX_BANDS);
    if (const char* e = std::getenv("RT_BANDS")) want = std::min<uint32_t>(RT_MAX_BANDS, std::max(1, std::atoi(e)));
    uint32_t bh = ((cam->vsize + want - 1) / want + 7u) & ~7u;
    nb = (int)((cam->vsize + bh - 1) / bh);
    for (int b = 0; b < nb; b++)
      bands[b] = PartInfo{bh, (uint32_t)nb, (uint32_t)b, std::min(bh, cam->vsize - (uint32_t)b * bh)};
  }
  Scene sc = ds->sc;
  sc.flags = opts ? opts->flags : 0;
  sc.sticky = ctx->d_sticky;
  RenderParams rp = make_params(opts);
  unsigned long long* d_blocks = ctx->d_counters + (size_t)RT_RING * RT_N_COUNTERS;
  unsigned long long earlier = 0;
render_again:
  attach_side(ctx, sc);
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  size_t row0 = 0;
  for (int b = 0; b < nb; b++) {
    const size_t p0 = row0 * cam->hsize;
    RtAux da{host[1] ? (uint8_t*)ctx->d_frame[1] + p0 : nullptr, host[2] ? (int32_t*)ctx->d_frame[2] + p0 : nullptr,
             host[3] ? (float*)ctx->d_frame[3] + p0 : nullptr};
    int rc = launch_part(ctx, sc, cam, rp, bands[b], (float*)ctx->d_frame[0] + 3 * p0, aux ? &da : nullptr, ctx->stream,
                         d_blocks + (size_t)b * RT_N_COUNTERS);
    if (rc == RT_OK && out_u8)
      rc = launch_quantize(ctx, (float*)ctx->d_frame[0] + 3 * p0, (uint8_t*)ctx->d_frame[4] + 3 * p0,
                           (uint64_t)bands[b].rows * cam->hsize, cam->hsize, bands[b], dither, ctx->stream);
    if (rc != RT_OK) {
      (void)hipStreamSynchronize(ctx->stream);
      return rc;
    }
    if (b + 1 == nb) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_band[b], ctx->stream));
    row0 += bands[b].rows;
  }
  // all bands are queued; now the copies, each as soon as its band is done (a copy to pageable memory keeps the
  // host busy, so none is issued before the last launch)
  row0 = 0;
  for (int b = 0; b < nb; b++) {
    const size_t p0 = row0 * cam->hsize;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_band[b], 0));
    const size_t n = (size_t)bands[b].rows * cam->hsize;
    for (int k = 0; k < 5; k++)
      if (host[k] && n)
        HIP_TRY(ctx, hipMemcpyAsync((char*)host[k] + p0 * elem[k], (char*)ctx->d_frame[k] + p0 * elem[k], n * elem[k],
                                    hipMemcpyDeviceToHost, ctx->copy_stream));
    row0 += bands[b].rows;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  RtStats local{};
  double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int rc = read_stats(ctx, ctx->stream, &local, total, nb, d_blocks);
  if (rc != RT_OK) return rc;
  if (stats) *stats = local;
  bool retry = false;
  rc = settle_overflow(ctx, ctx->stream, local.overflow, true, &retry, &earlier);
  if (retry) goto render_again;  // the context has (larger) side lists now: the same frame again
  return rc;
}

void* rt_host_alloc(RtContext* ctx, uint64_t bytes) {
  if (!ctx || bytes == 0) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    fail(ctx, RT_E_OOM, "rt_host_alloc: page-locked allocation of " + std::to_string(bytes) + " bytes failed");
    return nullptr;
  }
  return p;
}
void rt_host_free(RtContext* ctx, void* p) {
  if (!ctx || !p) return;
  (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}

int rt_render(RtContext* ctx, const RtDeviceScene* ds, const RtCamera* cam, const RtRenderOptions* opts,
              float* out_rgb, RtAux* aux, RtStats* stats) {
  if (!out_rgb) return RT_E_ABI;
  return render_host(ctx, ds, cam, opts, out_rgb, nullptr, RT_DITHER_NONE, aux, stats);
}

int rt_render_u8(RtContext* ctx, const RtDeviceScene* ds, const RtCamera* cam, const RtRenderOptions* opts, uint32_t dither,
                 uint8_t* out_u8, RtAux* aux, RtStats* stats) {
  if (!out_u8) return RT_E_ABI;
  return render_host(ctx, ds, cam, opts, nullptr, out_u8, dither, aux, stats);
}

// =========================================================================================
// Camera::render over several GPUs of one host: context g renders the rows r with
// (r / tile_h) % n == g of the frame, every context on a stream of its own, all launched before any is waited for.
//   rt_render_multi         canvas in HOST memory (what Camera::render returns): every GPU copies its stripes straight
//                           to where they belong in the caller's canvas (a strided copy over its own PCIe link); no
//                           GPU-to-GPU traffic at all.
//   rt_render_multi_device  canvas in the first context's HBM: the parts are gathered there over xGMI (RCCL send /
//                           recv in one group) and put in order by k_deinterleave.
// =========================================================================================
namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};
// librccl.so is opened on first use: a host that renders on one GPU never loads it.  RT_RCCL_LIB names another file.
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* name = std::getenv("RT_RCCL_LIB");
    for (const char* cand : {name, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (!cand) continue;
      api.lib = dlopen(cand, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      api.error = std::string("librccl.so could not be opened: ") + (dlerror() ? dlerror() : "not found");
      return;
    }
    bool ok = true;
    auto sym = [&](const char* n) {
      void* p = dlsym(api.lib, n);
      if (!p) {
        ok = false;
        api.error = std::string("librccl.so lacks ") + n;
      }
      return p;
    };
    api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return &api;
}
// One communicator per distinct list of devices, made when that list first gathers and kept until the process ends.
struct CommSet {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
};
std::mutex g_comm_mutex;
std::vector<CommSet*> g_comm_sets;
int multi_comms(RtContext* c0, const std::vector<int>& devices, CommSet** out) {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  for (CommSet* cs : g_comm_sets)
    if (cs->devices == devices) {
      *out = cs;
      return RT_OK;
    }
  RcclApi* api = rccl_api();
  if (!api->lib) return fail(c0, RT_E_DEVICE, "rt_render_multi_device: " + api->error);
  CommSet* cs = new CommSet();
  cs->devices = devices;
  cs->comms.resize(devices.size());
  ncclResult_t r = api->CommInitAll(cs->comms.data(), (int)devices.size(), devices.data());
  if (r != ncclSuccess) {
    delete cs;
    return fail(c0, RT_E_DEVICE, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
  }
  g_comm_sets.push_back(cs);
  *out = cs;
  return RT_OK;
}

// What one GPU does for a multi-GPU frame: its part into its own frame buffer (or `d_direct`), asynchronously.
struct MultiPart {
  PartInfo part;
  unsigned slot = 0;
  unsigned long long* block = nullptr;
  float* d_rgb = nullptr;
  bool launched = false;
};
int multi_frame_buffer(RtContext* ctx, size_t floats) {
  const size_t need = (floats ? floats : 1) * sizeof(float);
  if (ctx->frame_bytes[0] >= need) return RT_OK;
  if (ctx->d_frame[0]) (void)hipFree(ctx->d_frame[0]);
  ctx->d_frame[0] = nullptr;
  ctx->frame_bytes[0] = 0;
  hipError_t e = hipMalloc(&ctx->d_frame[0], need);
  if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_DEVICE, std::string("frame buffer: ") + hipGetErrorString(e));
  ctx->frame_bytes[0] = need;
  return RT_OK;
}
int multi_launch(RtContext* ctx, const RtDeviceScene* ds, const RtCamera* cam, const RtRenderOptions* opts, MultiPart& mp,
                 bool want_events) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = need_big(ctx, ds->needs_big, true);
  if (rc != RT_OK) return rc;
  Scene sc = ds->sc;
  sc.flags = opts ? opts->flags : 0;
  sc.sticky = ctx->d_sticky;
  attach_side(ctx, sc);
  RenderParams rp = make_params(opts);
  rc = ring_acquire(ctx, ctx->stream, &mp.slot, &mp.block);
  if (rc != RT_OK) return rc;
  if (want_events) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  rc = launch_part(ctx, sc, cam, rp, mp.part, mp.d_rgb, nullptr, ctx->stream, mp.block);
  if (rc != RT_OK) return rc;
  if (want_events) HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  mp.launched = true;
  return ring_release(ctx, ctx->stream, mp.slot);
}
// The rows of part g, compacted on the device, to their places in the caller's canvas: stripe k of the part is rows
// [(k n + g) tile_h, + tile_h) of the frame, so the whole stripes are ONE strided copy ("rows" of tile_h image rows,
// n tile_h image rows apart) and a last, shorter stripe is a second one.
int multi_copy_out(RtContext* ctx, const RtCamera* cam, const MultiPart& mp, float* out_rgb) {
  const size_t row_bytes = (size_t)cam->hsize * 3 * sizeof(float);
  const PartInfo& p = mp.part;
  const uint32_t whole = p.rows / p.tile_h, rest = p.rows % p.tile_h;
  char* dst = reinterpret_cast<char*>(out_rgb) + (size_t)p.part * p.tile_h * row_bytes;
  if (whole)
    HIP_TRY(ctx, hipMemcpy2DAsync(dst, (size_t)p.n_parts * p.tile_h * row_bytes, mp.d_rgb, (size_t)p.tile_h * row_bytes,
                                  (size_t)p.tile_h * row_bytes, whole, hipMemcpyDeviceToHost, ctx->stream));
  if (rest)
    HIP_TRY(ctx, hipMemcpyAsync(dst + (size_t)whole * p.n_parts * p.tile_h * row_bytes,
                                reinterpret_cast<const char*>(mp.d_rgb) + (size_t)whole * p.tile_h * row_bytes, rest * row_bytes,
                                hipMemcpyDeviceToHost, ctx->stream));
  return RT_OK;
}
// After the part's stream is idle: counters, and the cure for a list overflow (side lists, the part again).  `redo`
// re-renders and re-delivers the part synchronously.
int multi_settle(RtContext* ctx, MultiPart& mp, RtStats* stats, double total_ms, const std::function<int()>& redo) {
  unsigned long long earlier = 0;
  while (true) {
    RtStats local{};
    int rc = read_stats(ctx, ctx->stream, &local, total_ms, 1, mp.block);
    if (rc != RT_OK) return rc;
    if (stats) *stats = local;
    bool retry = false;
    rc = settle_overflow(ctx, ctx->stream, local.overflow, true, &retry, &earlier);
    if (!retry) return rc;
    rc = redo();
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
}
int multi_check_args(RtContext* const* ctxs, const RtDeviceScene* const* scenes, uint32_t n, const RtCamera* cam,
                     const void* out) {
  if (!ctxs || !scenes || !cam || !out || n == 0 || n > RT_MULTI_MAX) return RT_E_ABI;
  for (uint32_t g = 0; g < n; g++) {
    if (!ctxs[g] || !scenes[g]) return RT_E_ABI;
    if (scenes[g]->owner != ctxs[g]) return fail(ctxs[g], RT_E_ABI, "scene belongs to another context");
    for (uint32_t k = 0; k < g; k++)
      if (ctxs[k] == ctxs[g]) return fail(ctxs[0], RT_E_ABI, "rt_render_multi: a context is listed twice");
  }
  if (cam->hsize == 0 || cam->vsize == 0) return fail(ctxs[0], RT_E_ABI, "empty camera");
  return RT_OK;
}
RtRenderOptions multi_options(const RtRenderOptions* opts, uint32_t n, uint32_t g) {
  RtRenderOptions o{};
  if (opts) o = *opts;
  if (!opts) {
    o.jitter_mode = RT_JITTER_CONST;
    o.jitter_value = 0.5f;
  }
  if (o.tile_h == 0) o.tile_h = 8;  // stripes as tall as the kernel's 8x8 tiles
  o.n_parts = n;
  o.part = g;
  return o;
}
}  // namespace

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int rt_render_multi(RtContext* const* ctxs, const RtDeviceScene* const* scenes, uint32_t n, const RtCamera* cam,
                    const RtRenderOptions* opts, float* out_rgb, RtStats* stats) {
  int rc = multi_check_args(ctxs, scenes, n, cam, out_rgb);
  if (rc != RT_OK) return rc;
  auto t0 = std::chrono::steady_clock::now();
  std::vector<MultiPart> mp(n);
  std::vector<RtRenderOptions> po(n);
  // a copy into pageable memory keeps the calling thread busy for its whole length: one thread per GPU then
  bool locked = false;
  {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, out_rgb) == hipSuccess)
      locked = at.type == hipMemoryTypeHost;
    else
      (void)hipGetLastError();
  }
  auto one = [&](uint32_t g) -> int {
    RtContext* ctx = ctxs[g];
    po[g] = multi_options(opts, n, g);
    mp[g].part = make_part(cam, &po[g]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int r = multi_frame_buffer(ctx, (size_t)mp[g].part.rows * cam->hsize * 3);
    if (r != RT_OK) return r;
    mp[g].d_rgb = static_cast<float*>(ctx->d_frame[0]);
    r = multi_launch(ctx, scenes[g], cam, &po[g], mp[g], true);
    if (r != RT_OK) return r;
    return multi_copy_out(ctx, cam, mp[g], out_rgb);
  };
  auto finish = [&](uint32_t g) -> int {
    RtContext* ctx = ctxs[g];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return multi_settle(ctx, mp[g], stats ? &stats[g] : nullptr, total, [&]() -> int {
      int r = multi_launch(ctx, scenes[g], cam, &po[g], mp[g], true);
      return r != RT_OK ? r : multi_copy_out(ctx, cam, mp[g], out_rgb);
    });
  };
  std::vector<int> rcs(n, RT_OK);
  if (locked || n == 1) {
    for (uint32_t g = 0; g < n; g++) rcs[g] = one(g);  // every GPU is launched (and its copy queued) before any is waited for
    for (uint32_t g = 0; g < n; g++)
      if (rcs[g] == RT_OK) rcs[g] = finish(g);
  } else {
    std::vector<std::thread> pool;
    for (uint32_t g = 0; g < n; g++)
      pool.emplace_back([&, g] {
        rcs[g] = one(g);
        if (rcs[g] == RT_OK) rcs[g] = finish(g);
      });
    for (auto& t : pool) t.join();
  }
  for (uint32_t g = 0; g < n; g++)
    if (rcs[g] != RT_OK) {
      (void)hipSetDevice(ctxs[g]->device);
      (void)hipStreamSynchronize(ctxs[g]->stream);
      if (g != 0) ctxs[0]->error = "GPU " + std::to_string(ctxs[g]->device) + ": " + ctxs[g]->error;
      return rcs[g];
    }
  return RT_OK;
}

int rt_render_multi_device(RtContext* const* ctxs, const RtDeviceScene* const* scenes, uint32_t n, const RtCamera* cam,
                           const RtRenderOptions* opts, float* d_out_rgb, RtStats* stats) {
  int rc = multi_check_args(ctxs, scenes, n, cam, d_out_rgb);
  if (rc != RT_OK) return rc;
  auto t0 = std::chrono::steady_clock::now();
  RtContext* c0 = ctxs[0];
  std::vector<MultiPart> mp(n);
  std::vector<RtRenderOptions> po(n);
  PartLayout L{};
  L.n = n;
  L.vsize = cam->vsize;
  L.row_floats = cam->hsize * 3u;
  uint32_t rows_all = 0;
  for (uint32_t g = 0; g < n; g++) {
    po[g] = multi_options(opts, n, g);
    mp[g].part = make_part(cam, &po[g]);
    L.first[g] = rows_all;
    rows_all += mp[g].part.rows;
  }
  L.tile_h = po[0].tile_h;
  // RT_MULTI_FORCE_GATHER=1: one GPU goes through the exchange step as well (the rehearsal the tests use)
  const bool gather = n > 1 || std::getenv("RT_MULTI_FORCE_GATHER") != nullptr;
  // the first context's frame buffer holds all parts one after the other: its own part is rendered in place
  HIP_TRY(c0, hipSetDevice(c0->device));
  if (gather) {
    rc = multi_frame_buffer(c0, (size_t)rows_all * L.row_floats);
    if (rc != RT_OK) return rc;
  }
  for (uint32_t g = 0; g < n; g++) {
    RtContext* ctx = ctxs[g];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (g == 0) {
      mp[g].d_rgb = gather ? static_cast<float*>(c0->d_frame[0]) : d_out_rgb;
    } else {
      rc = multi_frame_buffer(ctx, (size_t)mp[g].part.rows * L.row_floats);
      if (rc != RT_OK) return rc;
      mp[g].d_rgb = static_cast<float*>(ctx->d_frame[0]);
    }
    rc = multi_launch(ctx, scenes[g], cam, &po[g], mp[g], true);
    if (rc != RT_OK) return rc;
  }
  auto exchange = [&]() -> int {
    if (!gather) return RT_OK;
    bool distinct = true;
    for (uint32_t g = 0; g < n; g++)
      for (uint32_t k = 0; k < g; k++) distinct = distinct && ctxs[g]->device != ctxs[k]->device;
    float* staging = static_cast<float*>(c0->d_frame[0]);
    if (distinct) {
      std::vector<int> devices(n);
      for (uint32_t g = 0; g < n; g++) devices[g] = ctxs[g]->device;
      CommSet* cs = nullptr;
      int r = multi_comms(c0, devices, &cs);
      if (r != RT_OK) return r;
      RcclApi* api = rccl_api();
      ncclResult_t nr = api->GroupStart();
      // (one rank: RCCL copies the part onto itself, the rehearsal on one GPU)
      for (uint32_t g = (n == 1 ? 0u : 1u); g < n && nr == ncclSuccess; g++) {
        const size_t count = (size_t)mp[g].part.rows * L.row_floats;
        if (!count) continue;
        nr = api->Recv(staging + (size_t)L.first[g] * L.row_floats, count, ncclFloat, (int)g, cs->comms[0], c0->stream);
        if (nr == ncclSuccess) nr = api->Send(mp[g].d_rgb, count, ncclFloat, 0, cs->comms[g], ctxs[g]->stream);
      }
      const ncclResult_t ne = api->GroupEnd();
      if (nr == ncclSuccess) nr = ne;
      if (nr != ncclSuccess) return fail(c0, RT_E_DEVICE, std::string("RCCL gather: ") + api->GetErrorString(nr));
    } else {
      // several contexts on one GPU (tests on a single-GPU box): RCCL takes a device once per communicator, and the
      // "exchange" is a copy inside that GPU's memory, ordered behind each part's render
      for (uint32_t g = 1; g < n; g++) {
        if (ctxs[g]->device != c0->device) return fail(c0, RT_E_ABI, "rt_render_multi_device: contexts share some GPUs but not all");
        const size_t bytes = (size_t)mp[g].part.rows * L.row_floats * sizeof(float);
        HIP_TRY(c0, hipEventRecord(ctxs[g]->ev_band[0], ctxs[g]->stream));
        HIP_TRY(c0, hipStreamWaitEvent(c0->stream, ctxs[g]->ev_band[0], 0));
        if (bytes)
          HIP_TRY(c0, hipMemcpyAsync(staging + (size_t)L.first[g] * L.row_floats, mp[g].d_rgb, bytes, hipMemcpyDeviceToDevice, c0->stream));
      }
    }
    HIP_TRY(c0, hipSetDevice(c0->device));
    hipLaunchKernelGGL(k_deinterleave, dim3((L.row_floats / 4u + 256u) / 256u, cam->vsize), dim3(256), 0, c0->stream, staging,
                       d_out_rgb, L);
    HIP_TRY(c0, hipGetLastError());
    return RT_OK;
  };
  rc = exchange();
  if (rc != RT_OK) return rc;
  // every part's stream, then the first context's (the gather and the de-interleave are on it)
  bool again = false;
  for (uint32_t g = n; g-- > 0;) {
    RtContext* ctx = ctxs[g];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    rc = multi_settle(ctx, mp[g], stats ? &stats[g] : nullptr, total, [&]() -> int {
      again = true;
      return multi_launch(ctx, scenes[g], cam, &po[g], mp[g], true);
    });
    if (rc != RT_OK) return rc;
  }
  if (again) {  // a part was rendered again (its intersection lists had overflowed): gather once more
    rc = exchange();
    if (rc != RT_OK) return rc;
    for (uint32_t g = 0; g < n; g++) {
      HIP_TRY(ctxs[g], hipSetDevice(ctxs[g]->device));
      HIP_TRY(ctxs[g], hipStreamSynchronize(ctxs[g]->stream));
    }
  }
  return RT_OK;
}

namespace {
// device staging of one rt_color_at batch: rays in, colours and hits out
struct RayBufs {
  float *o = nullptr, *d = nullptr, *rgb = nullptr, *t = nullptr, *uv = nullptr;
  int32_t* id = nullptr;
};
// Copies the rays in, launches, copies the results out and reads the counters of `block` (zeroed here).  Allocates
// nothing: the buffers and the counter block are the caller's.
int color_at_run(RtContext* ctx, const RtDeviceScene* ds, const RtRenderOptions* opts, const float* origins,
                 const float* directions, uint32_t n, uint32_t remaining, float* out_rgb, float* hit_t, int32_t* hit_id,
                 float* hit_uv, RtStats* stats_out, const RayBufs& b, unsigned long long* block,
                 std::chrono::steady_clock::time_point t0, bool may_grow) {
  const size_t n3 = (size_t)n * 3 * sizeof(float);
  HIP_TRY(ctx, hipMemcpyAsync(b.o, origins, n3, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(b.d, directions, n3, hipMemcpyHostToDevice, ctx->stream));
  Scene sc = ds->sc;
  sc.flags = opts ? opts->flags : 0;
  sc.sticky = ctx->d_sticky;
  RenderParams rp = make_params(opts);
  unsigned long long earlier = 0;
trace_again:
  attach_side(ctx, sc);
  HIP_TRY(ctx, hipMemsetAsync(block, 0, RT_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
  {
    int rc_side = side_enter(ctx, ctx->stream);
    if (rc_side != RT_OK) return rc_side;
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  RayOut ro{b.o, b.d, n, b.rgb, b.t, b.id, b.uv};
  if (sc.n_nodes == 0 && !sc.side_cap) {  // (side lists belong to the lanes of the persistent kernel)
    hipLaunchKernelGGL(k_color_at_direct, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, ctx->stream, sc, rp, ro,
                       remaining, block);
  } else {
    const uint32_t blocks = std::min<uint32_t>((n + RT_BLOCK - 1) / RT_BLOCK, ctx->n_cus * RT_BLOCKS_PER_CU);
    const bool cull = (sc.props & RT_SCENE_SHAPE_CULL) != 0;
    auto kern = (sc.flags & RT_RENDER_COUNT_WORK) ? (cull ? k_color_at_jobs<true, true> : k_color_at_jobs<true, false>)
                                                  : (cull ? k_color_at_jobs<false, true> : k_color_at_jobs<false, false>);
    if (sc.side_cap) kern = k_color_at_jobs<true, true, true>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(RT_BLOCK), 0, ctx->stream, sc, rp, ro, remaining, block);
  }
  if ((sc.props & RT_SCENE_HAS_COPIES) && hit_id && n)
    hipLaunchKernelGGL(k_copies_rays, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, sc, ro);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  {
    int rc_side = side_leave(ctx, ctx->stream);
    if (rc_side != RT_OK) return rc_side;
  }
  if (out_rgb) HIP_TRY(ctx, hipMemcpyAsync(out_rgb, b.rgb, n3, hipMemcpyDeviceToHost, ctx->stream));
  if (hit_t) HIP_TRY(ctx, hipMemcpyAsync(hit_t, b.t, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (hit_id) HIP_TRY(ctx, hipMemcpyAsync(hit_id, b.id, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (hit_uv) HIP_TRY(ctx, hipMemcpyAsync(hit_uv, b.uv, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  RtStats local{};
  double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int rc = read_stats(ctx, ctx->stream, &local, total, 1, block);
  if (rc != RT_OK) return rc;
  if (stats_out) *stats_out = local;
  bool retry = false;
  rc = settle_overflow(ctx, ctx->stream, local.overflow, may_grow, &retry, &earlier);
  if (retry) goto trace_again;
  return rc;
}
}  // namespace

int rt_color_at(RtContext* ctx, const RtDeviceScene* ds, const RtRenderOptions* opts, const float* origins,
                const float* directions, uint32_t n, uint32_t remaining, float* out_rgb, float* hit_t,
                int32_t* hit_id, float* hit_uv, RtStats* stats) {
  if (!ctx || !ds || (n && (!origins || !directions))) return RT_E_ABI;
  if (ds->owner != ctx) return fail(ctx, RT_E_ABI, "scene belongs to another context");
  if (remaining + 1 > RT_BIG_RAY_STACK) return fail(ctx, RT_E_SCENE, "remaining exceeds the device ray stack (16 levels)");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n == 0) return RT_OK;
  {
    int rc_big = need_big(ctx, ds->needs_big || remaining + 1 > RT_MAX_RAY_STACK, true);
    if (rc_big != RT_OK) return rc_big;
  }
  RayBufs b;
  auto cleanup = [&]() {
    for (void* p : {(void*)b.o, (void*)b.d, (void*)b.rgb, (void*)b.t, (void*)b.uv, (void*)b.id})
      if (p) (void)hipFree(p);
  };
  const size_t n3 = (size_t)n * 3 * sizeof(float);
  hipError_t e = hipMalloc((void**)&b.o, n3);
  if (e == hipSuccess) e = hipMalloc((void**)&b.d, n3);
  if (e == hipSuccess) e = hipMalloc((void**)&b.rgb, n3);
  if (e == hipSuccess) e = hipMalloc((void**)&b.t, (size_t)n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b.uv, (size_t)n * 2 * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&b.id, (size_t)n * sizeof(int32_t));
  if (e != hipSuccess) {
    cleanup();
    return fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_DEVICE, std::string("rt_color_at buffers: ") + hipGetErrorString(e));
  }
  unsigned slot = 0;
  unsigned long long* block = nullptr;
  int rc = ring_acquire(ctx, ctx->stream, &slot, &block);
  if (rc == RT_OK) {
    rc = color_at_run(ctx, ds, opts, origins, directions, n, remaining, out_rgb, hit_t, hit_id, hit_uv, stats, b, block, t0, true);
    (void)ring_release(ctx, ctx->stream, slot);
  }
  (void)hipStreamSynchronize(ctx->stream);
  cleanup();
  return rc;
}

// rt_color_at without any allocation: the rays are staged through this GPU's static scratch array.
int rt_color_at_static(RtContext* ctx, const RtDeviceScene* ds, const RtRenderOptions* opts, const float* origins,
                       const float* directions, uint32_t n, uint32_t remaining, float* out_rgb, float* hit_t,
                       int32_t* hit_id, float* hit_uv, RtStats* stats) {
  if (!ctx || !ds || (n && (!origins || !directions))) return RT_E_ABI;
  if (ds->owner != ctx) return fail(ctx, RT_E_ABI, "scene belongs to another context");
  if (remaining + 1 > RT_BIG_RAY_STACK) return fail(ctx, RT_E_SCENE, "remaining exceeds the device ray stack (16 levels)");
  {
    int rc_big = need_big(ctx, ds->needs_big || remaining + 1 > RT_MAX_RAY_STACK, false);
    if (rc_big != RT_OK) return rc_big;
  }
  if (n > RT_STATIC_MAX_RAYS)
    return fail(ctx, RT_E_LIMIT, "rt_color_at_static: " + std::to_string(n) + " rays exceed RT_STATIC_MAX_RAYS (" +
                                     std::to_string(RT_STATIC_MAX_RAYS) + "); split the batch or use rt_color_at");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n == 0) return RT_OK;
  float* base = reinterpret_cast<float*>(ctx->static_scratch);
  const size_t N = RT_STATIC_MAX_RAYS;
  RayBufs b;
  b.o = base;
  b.d = base + 3 * N;
  b.rgb = base + 6 * N;
  b.t = base + 9 * N;
  b.uv = base + 10 * N;
  b.id = reinterpret_cast<int32_t*>(base + 12 * N);
  unsigned long long* block = reinterpret_cast<unsigned long long*>(ctx->static_scratch + (size_t)RT_STATIC_WORDS * 4u);
  // the array keeps the last call's contents: the counter block is zeroed by color_at_run, every output word the
  // caller reads (entries 0..n) is written by the kernel
  // (may_grow = false: growing the side lists would allocate)
  const int rc = color_at_run(ctx, ds, opts, origins, directions, n, remaining, out_rgb, hit_t, hit_id, hit_uv, stats, b, block, t0, false);
  // the array is shared by every context on this GPU: after a failure in the middle of the call (a copy or an event
  // record) the kernel or copies may still be in flight on it, so the call only returns once the stream is idle
  if (rc != RT_OK) (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

}  // extern "C"
