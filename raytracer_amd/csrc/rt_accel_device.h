// rt_accel_device.h — the binary SAH tree of rt_accel.h (TreeBuilder::build) built on the device.
//
// Same algorithm, same arithmetic, level by level instead of depth first: every level of the tree is one pass over all
// triangles of the subtree.  A node is a contiguous range [b, e) of a permutation of the triangles; per level
//   k_bounds   box, centroid box, sum of unit normals, largest 1/sin(e1,e2) of every node (segmented reduction over the
//              lanes of a wave, one atomic per run and quantity: min / max on order-preserving integers, the normal sum
//              in 2^-32 fixed point, so that the result does not depend on the order the atomics arrive in)
//   k_prep     per node: bin scales, cone axis; clears its 3 x 16 bins
//   k_bin      16 bins per axis over the centroids (count, box): in LDS when a block's 256 triangles belong to one
//              node, straight to memory otherwise; the smallest cos(normal, cone axis) of the node
//   k_split    per node: leaf, or the cheapest of the 45 splits under rt_accel.h's rule (areas in double, the same
//              order of comparisons, the same limit on the levels that remain); writes its box and its reference into
//              the parent's record
//   k_flags / k_scan_blocks / k_children / k_scatter   stable partition of every splitting node at once: one exclusive
//              prefix sum of "goes left" over all positions, ranges of the children, the permuted triangles
// The tree that comes out has the boxes, splits and leaves of TreeBuilder's (tests/test_gpu_accel_device.py compares
// them one by one); cones agree to rounding (the host sums the normals in double in triangle order).  A node for which
// no split passes takes the object median, as in TreeBuilder (k_median: ranks by counting in LDS, for nodes of up to
// kMedianMax triangles; k_mcount / k_mupdate: the median key by bisection, for larger ones).  Leaves hold a run of the
// final permutation, so a leaf's first triangle is its `b`.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "rt_accel.h"

namespace rtaccel_dev {

constexpr uint32_t kNoW = 0xFFFFFFFFu;
constexpr int kBins = 16;
constexpr int kItemFloats = 13;  // lo[3] hi[3] c[3] nrm[3] kalpha, one array of N each
constexpr uint32_t kLeaf = 0u, kSplit = 1u, kMedian = 2u;
constexpr uint32_t kMedianMax = 4096;  // triangles of a node the device splits at the object median (ranks by counting, in LDS)

__device__ __forceinline__ int f2o(float f) {  // order-preserving float -> int (finite values and infinities)
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }
__device__ __forceinline__ long long d2o(double d) {
  const long long i = __double_as_longlong(d);
  return i >= 0 ? i : i ^ 0x7FFFFFFFFFFFFFFFll;
}
__device__ __forceinline__ double o2d(long long i) { return __longlong_as_double(i >= 0 ? i : i ^ 0x7FFFFFFFFFFFFFFFll); }

struct Glob {
  int blo[3], bhi[3];  // box of all (padded) triangles
  int lmax, amax;      // longest edge, largest |e1||e2|
  uint32_t bad;        // a non-finite triangle: the subtree keeps the reference walk (TreeBuilder::add)
  uint32_t n_nodes;    // binary nodes allocated (0 = header, 1 = top)
  uint32_t n_next;     // nodes of the next level
  uint32_t max_next;   // triangles of the largest node of the next level (k_children)
  uint32_t n_median;   // nodes of this level that take the object-median split, up to kMedianMax triangles (k_median)
  uint32_t n_large;    // ... and larger ones (k_mcount / k_mupdate)
};
struct WNode {  // a node of the level being built
  uint32_t b, e, parent, side;
  int lo[3], hi[3], clo[3], chi[3];
  long long ax[3];
  int kmax;
  uint32_t bad;  // a triangle without a usable normal below this node
  long long cmin;
  float scale[3], lo_c[3];
  uint32_t axis_ok, cone_ok;
  double axd[3], al;
  uint32_t kind, axis, split, me, child0;
  uint32_t large;  // an object-median split of more than kMedianMax triangles: the median key is found by bisection
  unsigned long long mlo, mhi, mmid;  // ... over [mlo, mhi], mmid the key being counted against
  uint32_t mcnt, pad_;                // ... triangles of the node with key <= mmid
};
struct Bin {
  uint32_t cnt;
  int lo[3], hi[3];
};

__device__ __forceinline__ void wnode_init(WNode& w, uint32_t b, uint32_t e, uint32_t parent, uint32_t side) {
  w.b = b, w.e = e, w.parent = parent, w.side = side;
  for (int a = 0; a < 3; a++) {
    w.lo[a] = w.clo[a] = f2o(INFINITY);
    w.hi[a] = w.chi[a] = f2o(-INFINITY);
    w.ax[a] = 0;
  }
  w.kmax = f2o(0.f);
  w.bad = 0;
  w.cmin = d2o(1.0);
  w.kind = kLeaf;
  w.large = 0;
}

__global__ void k_init(uint32_t n, uint32_t* perm, uint32_t* node_of, WNode* wn, Glob* g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = i, node_of[i] = 0;
  if (i == 0) {
    wnode_init(wn[0], 0, n, kNoW, 0);
    for (int a = 0; a < 3; a++) g->blo[a] = f2o(INFINITY), g->bhi[a] = f2o(-INFINITY);
    g->lmax = g->amax = f2o(0.f);
    g->bad = 0, g->n_nodes = 1, g->n_next = 0, g->max_next = 0, g->n_median = 0, g->n_large = 0;
  }
}

// TreeBuilder::add, one triangle per lane: the same float / double expressions in the same order
__global__ void k_items(const RtTriGeom* geom, const uint32_t* tri, uint32_t n, float* f, Glob* g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float lmax = 0.f, amax = 0.f, blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool bad = false;
  if (i < n) {
    const RtTriGeom q = geom[tri[i]];
    float lo[3], hi[3], ext = 0.f, mag = 0.f;
    for (int a = 0; a < 3; a++) {
      const float p1 = q.p1[a], p2 = p1 + q.e1[a], p3 = p1 + q.e2[a];
      if (!isfinite(p1) || !isfinite(p2) || !isfinite(p3)) bad = true;
      lo[a] = fminf(p1, fminf(p2, p3));
      hi[a] = fmaxf(p1, fmaxf(p2, p3));
      ext = fmaxf(ext, hi[a] - lo[a]);
      mag = fmaxf(mag, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
    }
    float l1 = 0.f, l2 = 0.f, l3 = 0.f;
    for (int a = 0; a < 3; a++) {
      const float d = q.e2[a] - q.e1[a];
      l1 += q.e1[a] * q.e1[a], l2 += q.e2[a] * q.e2[a], l3 += d * d;
    }
    float nrm[3], kalpha;
    {
      const double e1[3] = {q.e1[0], q.e1[1], q.e1[2]}, e2[3] = {q.e2[0], q.e2[1], q.e2[2]};
      const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      const double cl = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      const double sina = cl / (sqrt((double)l1) * sqrt((double)l2));
      if (cl > 0 && isfinite(sina) && sina > 1e-5) {
        for (int a = 0; a < 3; a++) nrm[a] = (float)(c[a] / cl);
        kalpha = (float)(1.0 / sina);
      } else {
        nrm[0] = nrm[1] = nrm[2] = 0.f;
        kalpha = INFINITY;
      }
    }
    const float edge = sqrtf(fmaxf(l1, fmaxf(l2, l3)));
    lmax = edge;
    amax = sqrtf(l1) * sqrtf(l2) * 1.000001f;
    const float pad = rtaccel::kPadRel * fmaxf(ext, edge) + rtaccel::kPadAbs * mag + 1e-30f;
    for (int a = 0; a < 3; a++) {
      lo[a] -= pad;
      hi[a] += pad;
      f[(size_t)a * n + i] = lo[a];
      f[(size_t)(3 + a) * n + i] = hi[a];
      f[(size_t)(6 + a) * n + i] = 0.5f * (lo[a] + hi[a]);
      f[(size_t)(9 + a) * n + i] = nrm[a];
      blo[a] = lo[a], bhi[a] = hi[a];
    }
    f[(size_t)12 * n + i] = kalpha;
  }
  for (int d = 32; d > 0; d >>= 1) {
    lmax = fmaxf(lmax, __shfl_down(lmax, d));
    amax = fmaxf(amax, __shfl_down(amax, d));
    for (int a = 0; a < 3; a++) blo[a] = fminf(blo[a], __shfl_down(blo[a], d)), bhi[a] = fmaxf(bhi[a], __shfl_down(bhi[a], d));
  }
  if (__ballot(bad)) {
    if ((threadIdx.x & 63u) == 0u) atomicOr(&g->bad, 1u);
  }
  if ((threadIdx.x & 63u) == 0u && blockIdx.x * blockDim.x + (threadIdx.x & ~63u) < n) {
    atomicMax(&g->lmax, f2o(lmax));
    atomicMax(&g->amax, f2o(amax));
    for (int a = 0; a < 3; a++) atomicMin(&g->blo[a], f2o(blo[a])), atomicMax(&g->bhi[a], f2o(bhi[a]));
  }
}

// Reduction over runs of equal keys among the 64 lanes of a wave (a run is contiguous: positions are consecutive and
// a node is a range of positions).  After the loop the first lane of every run holds the run's result.
template <class T, class Op>
__device__ __forceinline__ T run_reduce(T v, const uint32_t key, Op op) {
  const uint32_t lane = threadIdx.x & 63u;
  for (int d = 1; d < 64; d <<= 1) {
    const T o = __shfl_down(v, d);
    const uint32_t k = (uint32_t)__shfl_down((int)key, d);
    if (lane + d < 64u && k == key) v = op(v, o);
  }
  return v;
}
__device__ __forceinline__ bool run_head(const uint32_t key) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
  return key != kNoW && (lane == 0u || prev != key);
}

// A block takes kBoundsChunks consecutive pieces of 256 positions.  What belongs to the node of the block's first
// position is gathered in LDS and goes to memory once a block (high in the tree every position of a block belongs to
// one node, and one atomic per run of a wave was 218 k atomics on sixteen words: 3.9 of the level's 4 ms); runs of
// other nodes go to memory directly, as before.
constexpr uint32_t kBoundsChunks = 4;
__global__ __launch_bounds__(256) void k_bounds(const uint32_t* perm, const uint32_t* node_of, const float* f, uint32_t n, WNode* wn) {
  __shared__ int Llo[3], Lhi[3], Lclo[3], Lchi[3], Lkmax;
  __shared__ unsigned long long Lax[3];
  __shared__ uint32_t Lbad, Lany, ukey;
  const uint32_t p_first = blockIdx.x * (kBoundsChunks * 256u);
  if (threadIdx.x == 0) {
    ukey = p_first < n ? node_of[p_first] : kNoW;
    Lbad = 0u, Lany = 0u, Lkmax = f2o(-INFINITY);
    for (int a = 0; a < 3; a++) Llo[a] = Lclo[a] = f2o(INFINITY), Lhi[a] = Lchi[a] = f2o(-INFINITY), Lax[a] = 0ull;
  }
  __syncthreads();
  auto fmin_ = [](float x, float y) { return fminf(x, y); };
  auto fmax_ = [](float x, float y) { return fmaxf(x, y); };
  auto add_ = [](long long x, long long y) { return x + y; };
  auto or_ = [](int x, int y) { return x | y; };
  for (uint32_t ch = 0; ch < kBoundsChunks; ch++) {
    const uint32_t p = p_first + ch * 256u + threadIdx.x;
    const uint32_t key = p < n ? node_of[p] : kNoW;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, c[3] = {0.f, 0.f, 0.f};
    long long ax[3] = {0, 0, 0};
    float kmax = 0.f;
    int bad = 0;
    if (key != kNoW) {
      const uint32_t it = perm[p];
      for (int a = 0; a < 3; a++) {
        lo[a] = f[(size_t)a * n + it];
        hi[a] = f[(size_t)(3 + a) * n + it];
        c[a] = f[(size_t)(6 + a) * n + it];
        ax[a] = llrint((double)f[(size_t)(9 + a) * n + it] * 4294967296.0);
      }
      const float ka = f[(size_t)12 * n + it];
      bad = !isfinite(ka);
      kmax = bad ? 0.f : ka;
    }
    float clo[3], chi[3];
    for (int a = 0; a < 3; a++) {
      clo[a] = run_reduce(c[a], key, fmin_);
      chi[a] = run_reduce(c[a], key, fmax_);
      lo[a] = run_reduce(lo[a], key, fmin_);
      hi[a] = run_reduce(hi[a], key, fmax_);
      ax[a] = run_reduce(ax[a], key, add_);
    }
    kmax = run_reduce(kmax, key, fmax_);
    bad = run_reduce(bad, key, or_);
    if (run_head(key)) {
      if (key == ukey) {
        for (int a = 0; a < 3; a++) {
          atomicMin(&Llo[a], f2o(lo[a]));
          atomicMax(&Lhi[a], f2o(hi[a]));
          atomicMin(&Lclo[a], f2o(clo[a]));
          atomicMax(&Lchi[a], f2o(chi[a]));
          atomicAdd(&Lax[a], (unsigned long long)ax[a]);
        }
        atomicMax(&Lkmax, f2o(kmax));
        if (bad) atomicOr(&Lbad, 1u);
        Lany = 1u;
      } else {
        WNode& w = wn[key];
        for (int a = 0; a < 3; a++) {
          atomicMin(&w.lo[a], f2o(lo[a]));
          atomicMax(&w.hi[a], f2o(hi[a]));
          atomicMin(&w.clo[a], f2o(clo[a]));
          atomicMax(&w.chi[a], f2o(chi[a]));
          atomicAdd(reinterpret_cast<unsigned long long*>(&w.ax[a]), (unsigned long long)ax[a]);
        }
        atomicMax(&w.kmax, f2o(kmax));
        if (bad) atomicOr(&w.bad, 1u);
      }
    }
  }
  __syncthreads();
  if (Lany && ukey != kNoW) {
    WNode& w = wn[ukey];
    const uint32_t a = threadIdx.x;
    if (a < 3u) {
      atomicMin(&w.lo[a], Llo[a]);
      atomicMax(&w.hi[a], Lhi[a]);
      atomicMin(&w.clo[a], Lclo[a]);
      atomicMax(&w.chi[a], Lchi[a]);
      atomicAdd(reinterpret_cast<unsigned long long*>(&w.ax[a]), Lax[a]);
    }
    if (a == 3u) {
      atomicMax(&w.kmax, Lkmax);
      if (Lbad) atomicOr(&w.bad, 1u);
    }
  }
}

__global__ void k_prep(WNode* wn, uint32_t m, Bin* bins, Glob* g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0u) g->n_next = 0u, g->n_median = 0u, g->n_large = 0u, g->max_next = 0u;  // this level's counters (the host has read the last level's)
  if (i >= m) return;
  WNode& w = wn[i];
  const uint32_t n = w.e - w.b;
  if (n <= rtaccel::kLeafTris) return;
  uint32_t ok = 0;
  for (int a = 0; a < 3; a++) {
    const float clo = o2f(w.clo[a]), chi = o2f(w.chi[a]);
    const float ext = chi - clo;
    w.lo_c[a] = clo;
    w.scale[a] = 0.f;
    if (ext > 0.f) {
      ok |= 1u << a;
      w.scale[a] = kBins / ext;
    }
  }
  w.axis_ok = ok;
  double al2 = 0;
  for (int a = 0; a < 3; a++) {
    w.axd[a] = (double)w.ax[a] * (1.0 / 4294967296.0);
    al2 += w.axd[a] * w.axd[a];
  }
  w.al = sqrt(al2);
  w.cone_ok = !w.bad && w.al > 1e-6 * n;
  Bin* B = bins + (size_t)i * 3 * kBins;
  for (int k = 0; k < 3 * kBins; k++) {
    B[k].cnt = 0;
    for (int a = 0; a < 3; a++) B[k].lo[a] = f2o(INFINITY), B[k].hi[a] = f2o(-INFINITY);
  }
}

__device__ __forceinline__ int bin_of(float c, float lo_c, float scale) {
  return min(kBins - 1, max(0, (int)((c - lo_c) * scale)));
}

constexpr uint32_t kBinChunks = 1;  // pieces of 256 positions a block takes (4: one flush of LDS bins instead of four high in the tree,
                                    // but four times as many levels whose nodes are narrower than a block and bin straight to memory: 7.4 -> 11.4 ms)
__global__ __launch_bounds__(256) void k_bin(const uint32_t* perm, const uint32_t* node_of, const float* f, uint32_t n, WNode* wn,
                                             Bin* bins) {
  __shared__ Bin L[3 * kBins];
  __shared__ uint32_t ukey;  // the block's one node, or kNoW when its positions belong to several (or to none)
  const uint32_t p_first = blockIdx.x * (kBinChunks * 256u);
  if (threadIdx.x == 0) {
    const uint32_t last = min(n, p_first + kBinChunks * 256u) - 1u;
    const uint32_t k0 = p_first < n ? node_of[p_first] : kNoW;
    ukey = (k0 != kNoW && node_of[last] == k0) ? k0 : kNoW;
  }
  for (uint32_t k = threadIdx.x; k < 3 * kBins; k += blockDim.x) {
    L[k].cnt = 0;
    for (int a = 0; a < 3; a++) L[k].lo[a] = f2o(INFINITY), L[k].hi[a] = f2o(-INFINITY);
  }
  __syncthreads();
  const bool in_lds = ukey != kNoW;
  for (uint32_t ch = 0; ch < kBinChunks; ch++) {
    const uint32_t p = p_first + ch * 256u + threadIdx.x;
    const uint32_t key = p < n ? node_of[p] : kNoW;
    bool active = false;
    double cosv = 1.0;
    if (key != kNoW) {
      const WNode& w = wn[key];
      if (w.e - w.b > rtaccel::kLeafTris) {
        active = true;
        const uint32_t it = perm[p];
        int lo[3], hi[3];
        float c[3], nr[3];
        for (int a = 0; a < 3; a++) {
          lo[a] = f2o(f[(size_t)a * n + it]);
          hi[a] = f2o(f[(size_t)(3 + a) * n + it]);
          c[a] = f[(size_t)(6 + a) * n + it];
          nr[a] = f[(size_t)(9 + a) * n + it];
        }
        for (int axis = 0; axis < 3; axis++) {
          if (!((w.axis_ok >> axis) & 1u)) continue;
          const int k = bin_of(c[axis], w.lo_c[axis], w.scale[axis]);
          if (in_lds) {
            Bin& B = L[axis * kBins + k];
            atomicAdd(&B.cnt, 1u);
            for (int a = 0; a < 3; a++) atomicMin(&B.lo[a], lo[a]), atomicMax(&B.hi[a], hi[a]);
          } else {
            Bin& B = bins[((size_t)key * 3 + axis) * kBins + k];
            atomicAdd(&B.cnt, 1u);
            for (int a = 0; a < 3; a++) atomicMin(&B.lo[a], lo[a]), atomicMax(&B.hi[a], hi[a]);
          }
        }
        if (w.cone_ok) cosv = (w.axd[0] * nr[0] + w.axd[1] * nr[1] + w.axd[2] * nr[2]) / w.al;
      }
    }
    // smallest cosine of the node: runs of the wave, one atomic each
    {
      const uint32_t k2 = active ? key : kNoW;
      const double mn = run_reduce(cosv, k2, [](double x, double y) { return x < y ? x : y; });
      if (run_head(k2) && wn[k2].cone_ok) atomicMin(&wn[k2].cmin, d2o(mn));
    }
  }
  __syncthreads();
  if (in_lds)
    for (uint32_t k = threadIdx.x; k < 3 * kBins; k += blockDim.x)
      if (L[k].cnt) {
        Bin& B = bins[(size_t)ukey * 3 * kBins + k];
        atomicAdd(&B.cnt, L[k].cnt);
        for (int a = 0; a < 3; a++) atomicMin(&B.lo[a], L[k].lo[a]), atomicMax(&B.hi[a], L[k].hi[a]);
      }
}

__device__ __forceinline__ int levels_for(uint32_t n) {
  int l = 0;
  while (n > rtaccel::kLeafTris) {
    n = (n + 1) / 2;
    l++;
  }
  return l;
}

__global__ void k_split(WNode* wn, uint32_t m, const Bin* bins, rtaccel::Node* nodes, Glob* g, int depth, uint32_t* median_list) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  WNode& w = wn[i];
  const uint32_t n = w.e - w.b;
  float* bmin = nullptr;
  float* bmax = nullptr;
  uint32_t *ref = nullptr, *count = nullptr;
  if (w.parent != kNoW) {
    rtaccel::Node& P = nodes[w.parent];
    bmin = w.side ? P.rmin : P.lmin, bmax = w.side ? P.rmax : P.lmax;
    ref = w.side ? &P.right : &P.left, count = w.side ? &P.rcount : &P.lcount;
    for (int a = 0; a < 3; a++) bmin[a] = o2f(w.lo[a]), bmax[a] = o2f(w.hi[a]);
  }
  if (n <= rtaccel::kLeafTris) {
    w.kind = kLeaf;
    if (ref) *ref = w.b, *count = n;
    return;
  }
  const Bin* B = bins + (size_t)i * 3 * kBins;
  double best = INFINITY;
  int best_axis = -1, best_split = 0;
  for (int axis = 0; axis < 3; axis++) {
    if (!((w.axis_ok >> axis) & 1u)) continue;
    const Bin* b = B + axis * kBins;
    double ra[kBins];
    uint32_t rn[kBins];
    float l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t acc = 0;
    auto area = [](const float* l, const float* h) {
      const double dx = (double)h[0] - l[0], dy = (double)h[1] - l[1], dz = (double)h[2] - l[2];
      return dx * dy + dy * dz + dz * dx;
    };
    for (int k = kBins - 1; k > 0; k--) {
      for (int a = 0; a < 3; a++) l3[a] = fminf(l3[a], o2f(b[k].lo[a])), h3[a] = fmaxf(h3[a], o2f(b[k].hi[a]));
      acc += b[k].cnt;
      ra[k] = acc ? area(l3, h3) : 0.0;
      rn[k] = acc;
    }
    for (int a = 0; a < 3; a++) l3[a] = INFINITY, h3[a] = -INFINITY;
    acc = 0;
    for (int k = 0; k + 1 < kBins; k++) {
      for (int a = 0; a < 3; a++) l3[a] = fminf(l3[a], o2f(b[k].lo[a])), h3[a] = fmaxf(h3[a], o2f(b[k].hi[a]));
      acc += b[k].cnt;
      if (acc == 0 || rn[k + 1] == 0) continue;
      if (depth + 1 + levels_for(acc) > rtaccel::kMaxDepth || depth + 1 + levels_for(rn[k + 1]) > rtaccel::kMaxDepth) continue;
      const double cost = area(l3, h3) * acc + ra[k + 1] * rn[k + 1];
      if (cost < best) best = cost, best_axis = axis, best_split = k;
    }
  }
  if (best_axis < 0) {
    // TreeBuilder: the object median on the axis of largest centroid extent (k_median ranks the triangles)
    int axis = 0;
    for (int a = 1; a < 3; a++)
      if (o2f(w.chi[a]) - o2f(w.clo[a]) > o2f(w.chi[axis]) - o2f(w.clo[axis])) axis = a;
    w.kind = kMedian;
    w.axis = (uint32_t)axis;
    w.split = 0;
    if (n > kMedianMax) {
      w.large = 1;
      w.mlo = 0ull, w.mhi = ~0ull, w.mmid = ~0ull >> 1, w.mcnt = 0;
      atomicAdd(&g->n_large, 1u);
    } else {
      median_list[atomicAdd(&g->n_median, 1u)] = i;
    }
  } else {
    w.kind = kSplit;
    w.axis = (uint32_t)best_axis;
    w.split = (uint32_t)best_split;
  }
  const uint32_t me = atomicAdd(&g->n_nodes, 1u);
  w.me = me;
  if (ref) *ref = me, *count = 0;
  rtaccel::Node& N = nodes[me];
  N.cone[0] = N.cone[1] = N.cone[2] = 0.f;
  N.cone[3] = -2.f;
  if (w.cone_ok) {
    const double cmin = o2d(w.cmin);
    const double wd = acos(fmax(-1.0, fmin(1.0, cmin))) + 2e-3;
    if (wd < 1.4) {
      const double inv_k = 1.0 / ((double)o2f(w.kmax) * 1.001);
      for (int a = 0; a < 3; a++) N.cone[a] = (float)(w.axd[a] / w.al * inv_k);
      N.cone[3] = (float)cos(wd);
    }
  }
}

// Object-median split (TreeBuilder's fallback: nth_element over (centroid on the axis, shape index)): a triangle goes
// left when fewer than n / 2 triangles of the node come before it in that order.  Ranks by counting, keys in LDS; one
// block per node of the level's list.  The halves hold the same triangles as the host's; their order inside a half is
// the order they had (nth_element leaves it unspecified), so leaves below such a node hold the same triangles in
// another order.
__global__ __launch_bounds__(256) void k_median(const uint32_t* perm, const float* f, const uint32_t* shape, uint32_t n,
                                                const WNode* wn, const uint32_t* median_list, const Glob* g, uint8_t* mflag) {
  __shared__ float kc[kMedianMax];
  __shared__ uint32_t ks[kMedianMax];
  for (uint32_t q = blockIdx.x; q < g->n_median; q += gridDim.x) {
    const WNode& w = wn[median_list[q]];
    const uint32_t cnt = w.e - w.b;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) {
      const uint32_t it = perm[w.b + k];
      kc[k] = f[(size_t)(6 + w.axis) * n + it];
      ks[k] = shape[it];
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) {
      const float c = kc[k];
      const uint32_t sh = ks[k];
      uint32_t before = 0;
      for (uint32_t j = 0; j < cnt; j++) before += (kc[j] < c || (kc[j] == c && ks[j] < sh)) ? 1u : 0u;
      mflag[w.b + k] = before < cnt / 2u ? 1 : 0;
    }
  }
}

// The same split for nodes too large for that: the order (centroid on the axis, shape index) as one 64-bit key, and the
// key of the triangle that n / 2 - 1 others precede found by bisection -- 64 rounds of "how many keys of the node are <=
// mmid" (k_mcount, all such nodes of the level at once) and "which half of the interval" (k_mupdate).  A triangle goes
// left when its key is <= that key.  (-0 and +0 compare equal in TreeBuilder's order: c + 0 makes them one key.)
__device__ __forceinline__ unsigned long long median_key(float c, uint32_t shape) {
  return ((unsigned long long)((uint32_t)f2o(c + 0.f) ^ 0x80000000u) << 32) | shape;
}
__global__ void k_mcount(const uint32_t* perm, const uint32_t* node_of, const float* f, const uint32_t* shape, uint32_t n,
                         WNode* wn) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t key = p < n ? node_of[p] : kNoW;
  uint32_t below = 0;
  if (key != kNoW) {
    const WNode& w = wn[key];
    if (w.kind == kMedian && w.large) {
      const uint32_t it = perm[p];
      below = median_key(f[(size_t)(6 + w.axis) * n + it], shape[it]) <= w.mmid ? 1u : 0u;
    } else {
      key = kNoW;
    }
  }
  const uint32_t sum = run_reduce(below, key, [](uint32_t x, uint32_t y) { return x + y; });
  if (run_head(key) && sum) atomicAdd(&wn[key].mcnt, sum);
}
__global__ void k_mupdate(WNode* wn, uint32_t m) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  WNode& w = wn[i];
  if (w.kind != kMedian || !w.large || w.mlo == w.mhi) return;
  if (w.mcnt >= (w.e - w.b) / 2u)
    w.mhi = w.mmid;
  else
    w.mlo = w.mmid + 1ull;
  w.mmid = w.mlo + (w.mhi - w.mlo) / 2ull;
  w.mcnt = 0;
}

// flag = this position goes to the left child; exclusive prefix sums inside the block, the block's total on the side
__global__ __launch_bounds__(256) void k_flags(const uint32_t* perm, const uint32_t* node_of, const float* f, const uint32_t* shape,
                                               uint32_t n, const WNode* wn, const uint8_t* mflag, uint32_t* pre, uint8_t* flag,
                                               uint32_t* block_sum) {
  __shared__ uint32_t wave_sum[4];
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t key = p < n ? node_of[p] : kNoW;
  uint32_t fl = 0;
  if (key != kNoW) {
    const WNode& w = wn[key];
    if (w.kind == kSplit)
      fl = bin_of(f[(size_t)(6 + w.axis) * n + perm[p]], w.lo_c[w.axis], w.scale[w.axis]) <= (int)w.split;
    else if (w.kind == kMedian)
      fl = w.large ? (median_key(f[(size_t)(6 + w.axis) * n + perm[p]], shape[perm[p]]) <= w.mlo ? 1u : 0u) : mflag[p];
  }
  const unsigned long long b = __ballot(fl);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t before = __popcll(b & ((1ull << lane) - 1ull));
  if (lane == 0) wave_sum[wave] = __popcll(b);
  __syncthreads();
  uint32_t off = 0;
  for (uint32_t k = 0; k < wave; k++) off += wave_sum[k];
  if (p < n) pre[p] = off + before, flag[p] = (uint8_t)fl;
  if (threadIdx.x == 0) block_sum[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}
// exclusive scan of the block totals, in place (one block)
__global__ __launch_bounds__(1024) void k_scan_blocks(uint32_t* block_sum, uint32_t nb) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (nb + 1023u) / 1024u;
  const uint32_t b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
  uint32_t s = 0;
  for (uint32_t k = b0; k < b1; k++) s += block_sum[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (uint32_t k = 0; k < 1024u; k++) {
      const uint32_t v = part[k];
      part[k] = run;
      run += v;
    }
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (uint32_t k = b0; k < b1; k++) {
    const uint32_t v = block_sum[k];
    block_sum[k] = run;
    run += v;
  }
}
__device__ __forceinline__ uint32_t prefix_at(const uint32_t* pre, const uint32_t* block_sum, uint32_t p) {
  return pre[p] + block_sum[p >> 8];
}
__global__ void k_children(WNode* wn, uint32_t m, WNode* next, uint32_t next_cap, const uint32_t* pre, const uint8_t* flag,
                           const uint32_t* block_sum, Glob* g) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  WNode& w = wn[i];
  if (w.kind == kLeaf) return;
  const uint32_t n_left = prefix_at(pre, block_sum, w.e - 1u) + flag[w.e - 1u] - prefix_at(pre, block_sum, w.b);
  const uint32_t mid = w.b + n_left;
  const uint32_t s = atomicAdd(&g->n_next, 2u);
  if (s + 2u > next_cap) {  // (build_tree sizes `next` for every level there can be: never seen, never trusted)
    atomicOr(&g->bad, 2u);
    return;
  }
  atomicMax(&g->max_next, max(n_left, (w.e - w.b) - n_left));
  w.child0 = s;
  wnode_init(next[s], w.b, mid, w.me, 0);
  wnode_init(next[s + 1], mid, w.e, w.me, 1);
}
__global__ void k_scatter(const uint32_t* perm, const uint32_t* node_of, uint32_t n, const WNode* wn, const uint32_t* pre,
                          const uint8_t* flag, const uint32_t* block_sum, uint32_t* perm2, uint32_t* node_of2) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t key = node_of[p];
  if (key == kNoW || wn[key].kind == kLeaf) {
    perm2[p] = perm[p];
    node_of2[p] = kNoW;
    return;
  }
  const WNode& w = wn[key];
  const uint32_t base = prefix_at(pre, block_sum, w.b);
  const uint32_t n_left = prefix_at(pre, block_sum, w.e - 1u) + flag[w.e - 1u] - base;
  const uint32_t rank_left = prefix_at(pre, block_sum, p) - base;
  const bool left = flag[p] != 0;
  const uint32_t q = left ? w.b + rank_left : w.b + n_left + ((p - w.b) - rank_left);
  perm2[q] = perm[p];
  node_of2[q] = w.child0 + (left ? 0u : 1u);
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {
    if (n <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr, cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)) != hipSuccess) return false;
    cap = n;
    return true;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

// The tree over the triangles of the subtree under reference node R, into `tb` as TreeBuilder::finish leaves it
// (nodes: header, then the binary nodes in pre-order; tris in leaf order).  false: not built here (no device memory, a
// HIP error, a non-finite triangle): the caller uses TreeBuilder itself.
inline bool build_tree(int device, const RtScene* s, uint32_t R, rtaccel::TreeBuilder& tb) {
  std::vector<uint32_t> tri, shape, owner;
  {
    size_t total = 0;
    for (uint32_t i = R; i < s->nodes[R].skip; i++) total += s->nodes[i].n_shapes;
    tri.reserve(total), shape.reserve(total), owner.reserve(total);
  }
  for (uint32_t i = R; i < s->nodes[R].skip; i++) {
    const RtNode& nd = s->nodes[i];
    for (uint32_t k = 0; k < nd.n_shapes; k++) tri.push_back(nd.first_tri + k), shape.push_back(nd.first_shape + k), owner.push_back(i);
  }
  const uint32_t n = (uint32_t)tri.size();
  if (n < rtaccel::kMinTris || n > 0x0FFFFFF0u) return false;
  const bool timing = std::getenv("RT_TIMING") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto stage = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "device tree: %-37s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
    t_prev = now;
  };
  if (hipSetDevice(device) != hipSuccess) return false;
  // the per-level counters come back through page-locked memory (two read-backs a level: into pageable memory each
  // one went through a staging copy); freed after the stream has drained
  struct Pinned {
    Glob* p = nullptr;
    ~Pinned() {
      if (p) (void)hipHostFree(p);
    }
  } pinned;
  if (hipHostMalloc(reinterpret_cast<void**>(&pinned.p), sizeof(Glob), hipHostMallocDefault) != hipSuccess) return false;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
  struct StreamGuard {
    hipStream_t s;
    ~StreamGuard() {  // (every way out: nothing of this call is in flight once its host vectors go)
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  } guard{st};
  // the triangles the subtree uses are a few runs of tri_geom: upload the span that holds them
  uint32_t t_lo = 0xFFFFFFFFu, t_hi = 0;
  for (uint32_t t : tri) t_lo = std::min(t_lo, t), t_hi = std::max(t_hi, t);
  for (uint32_t& t : tri) t -= t_lo;
  const size_t span = (size_t)t_hi - t_lo + 1;
  DevBuf<RtTriGeom> d_geom;
  DevBuf<uint32_t> d_tri, d_shape, d_perm, d_perm2, d_node, d_node2, d_pre, d_bsum, d_mlist;
  DevBuf<uint8_t> d_flag, d_mflag;
  DevBuf<float> d_f;
  DevBuf<Glob> d_g;
  DevBuf<WNode> d_w, d_w2;
  DevBuf<Bin> d_bins;
  DevBuf<rtaccel::Node> d_nodes;
  const uint32_t nb = (n + 255u) / 256u;
  if (!d_geom.ensure(span) || !d_tri.ensure(n) || !d_perm.ensure(n) || !d_perm2.ensure(n) || !d_node.ensure(n) ||
      !d_node2.ensure(n) || !d_pre.ensure(n) || !d_bsum.ensure(nb) || !d_flag.ensure(n) || !d_mflag.ensure(n) || !d_shape.ensure(n) || !d_f.ensure((size_t)kItemFloats * n) ||
      !d_g.ensure(1) || !d_nodes.ensure((size_t)n + 2))
    return false;
#define RT_DEV_TRY(x)                                                                                  \
  if (hipError_t e_ = (x); e_ != hipSuccess) {                                                         \
    if (timing) std::fprintf(stderr, "device tree: %s at line %d\n", hipGetErrorString(e_), __LINE__); \
    return false;                                                                                      \
  }
  stage("item lists, buffers");
  RT_DEV_TRY(hipMemcpyAsync(d_geom.p, s->tri_geom + t_lo, span * sizeof(RtTriGeom), hipMemcpyHostToDevice, st));
  RT_DEV_TRY(hipMemcpyAsync(d_tri.p, tri.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  RT_DEV_TRY(hipMemcpyAsync(d_shape.p, shape.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
  // A node that is split holds more than kLeafTris triangles, so no level has more than 2 n / (kLeafTris + 1) nodes: the
  // per-level buffers are made for that once (growing them level by level was a hipFree -- a device-wide wait -- and a
  // hipMalloc three times a level).
  const size_t w_cap = std::max<size_t>(1024, 2 * ((size_t)n / (rtaccel::kLeafTris + 1)) + 16);
  if (!d_w.ensure(w_cap) || !d_w2.ensure(w_cap) || !d_bins.ensure(w_cap * 3 * kBins) || !d_mlist.ensure(w_cap)) return false;
  hipLaunchKernelGGL(k_init, dim3(nb), dim3(256), 0, st, n, d_perm.p, d_node.p, d_w.p, d_g.p);
  hipLaunchKernelGGL(k_items, dim3(nb), dim3(256), 0, st, d_geom.p, d_tri.p, n, d_f.p, d_g.p);
  Glob& g = *pinned.p;
  g = Glob{};
  uint32_t m = 1, level_max = n;
  int depth = 1, depth_seen = 1;
  while (m > 0) {
    depth_seen = depth;
    const uint32_t mb = (m + 63u) / 64u;
    hipLaunchKernelGGL(k_bounds, dim3((nb + kBoundsChunks - 1u) / kBoundsChunks), dim3(256), 0, st, d_perm.p, d_node.p, d_f.p, n, d_w.p);
    hipLaunchKernelGGL(k_prep, dim3(mb), dim3(64), 0, st, d_w.p, m, d_bins.p, d_g.p);
    hipLaunchKernelGGL(k_bin, dim3((nb + kBinChunks - 1u) / kBinChunks), dim3(256), 0, st, d_perm.p, d_node.p, d_f.p, n, d_w.p, d_bins.p);
    hipLaunchKernelGGL(k_split, dim3(mb), dim3(64), 0, st, d_w.p, m, d_bins.p, d_nodes.p, d_g.p, depth, d_mlist.p);
    // object-median splits of large nodes (deep under a mesh close to what one tree can hold): the host has to know of
    // them, which costs a round trip -- only on levels that still have a node that large
    if (level_max > kMedianMax) {
      RT_DEV_TRY(hipMemcpyAsync(&g, d_g.p, sizeof(g), hipMemcpyDeviceToHost, st));
      RT_DEV_TRY(hipStreamSynchronize(st));
      if (g.n_large)
        for (int round = 0; round < 64; round++) {
          hipLaunchKernelGGL(k_mcount, dim3(nb), dim3(256), 0, st, d_perm.p, d_node.p, d_f.p, d_shape.p, n, d_w.p);
          hipLaunchKernelGGL(k_mupdate, dim3(mb), dim3(64), 0, st, d_w.p, m);
        }
    }
    // (its blocks take the entries of the list the device counted: any grid will do)
    hipLaunchKernelGGL(k_median, dim3(std::min<uint32_t>(m, 2048u)), dim3(256), 0, st, d_perm.p, d_f.p, d_shape.p, n, d_w.p, d_mlist.p,
                       d_g.p, d_mflag.p);
    hipLaunchKernelGGL(k_flags, dim3(nb), dim3(256), 0, st, d_perm.p, d_node.p, d_f.p, d_shape.p, n, d_w.p, d_mflag.p, d_pre.p,
                       d_flag.p, d_bsum.p);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, st, d_bsum.p, nb);
    hipLaunchKernelGGL(k_children, dim3(mb), dim3(64), 0, st, d_w.p, m, d_w2.p, (uint32_t)w_cap, d_pre.p, d_flag.p, d_bsum.p, d_g.p);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, st, d_perm.p, d_node.p, n, d_w.p, d_pre.p, d_flag.p, d_bsum.p, d_perm2.p,
                       d_node2.p);
    RT_DEV_TRY(hipGetLastError());
    RT_DEV_TRY(hipMemcpyAsync(&g, d_g.p, sizeof(g), hipMemcpyDeviceToHost, st));
    RT_DEV_TRY(hipStreamSynchronize(st));
    if (g.bad) {
      if (timing) std::fprintf(stderr, "device tree: declined at depth %d (%s)\n", depth, (g.bad & 2u) ? "more nodes in a level than there can be" : "non-finite triangle");
      return false;
    }
    level_max = g.max_next;
    std::swap(d_perm.p, d_perm2.p);
    std::swap(d_node.p, d_node2.p);
    m = g.n_next;
    if (m == 0) break;
    if (depth >= rtaccel::kMaxDepth) return false;  // cannot happen under k_split's rule
    depth++;
    std::swap(d_w.p, d_w2.p);
    std::swap(d_w.cap, d_w2.cap);
    if ((size_t)m > w_cap) return false;  // (cannot happen: see w_cap)
  }
  stage("levels");
  const uint32_t n_nodes = g.n_nodes;
  if (n_nodes < 2) return false;  // a single leaf: TreeBuilder::finish refuses it too
  std::vector<rtaccel::Node> bfs(n_nodes);
  std::vector<uint32_t> perm(n);
  RT_DEV_TRY(hipMemcpyAsync(bfs.data(), d_nodes.p, (size_t)n_nodes * sizeof(rtaccel::Node), hipMemcpyDeviceToHost, st));
  RT_DEV_TRY(hipMemcpyAsync(perm.data(), d_perm.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  RT_DEV_TRY(hipStreamSynchronize(st));
#undef RT_DEV_TRY
  // the leaf triangles need the permutation only: gathered on threads while this one renumbers the nodes
  tb.tris.resize(n);
  std::vector<std::thread> th;
  {
    auto fill = [&, t_lo](uint32_t p0, uint32_t p1) {
      for (uint32_t p = p0; p < p1; p++) {
        const uint32_t it = perm[p];
        const RtTriGeom& q = s->tri_geom[tri[it] + t_lo];
        rtaccel::Tri t{};
        std::memcpy(t.p1, q.p1, 12);
        std::memcpy(t.e1, q.e1, 12);
        std::memcpy(t.e2, q.e2, 12);
        t.shape = shape[it];
        t.owner = owner[it];
        t.kind_flags = s->shapes[shape[it]].kind_flags;
        tb.tris[p] = t;
      }
    };
    const uint32_t nt = n >= (1u << 16) ? std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1u;
    for (uint32_t k = 0; k < nt; k++) {
      const uint32_t p0 = (uint32_t)((uint64_t)n * k / nt), p1 = (uint32_t)((uint64_t)n * (k + 1) / nt);
      if (nt == 1)
        fill(p0, p1);
      else
        th.emplace_back(fill, p0, p1);
    }
  }
  struct Joiner {
    std::vector<std::thread>& th;
    ~Joiner() {
      for (auto& t : th)
        if (t.joinable()) t.join();
    }
  } joiner{th};
  // pre-order numbering (TreeBuilder's layout: a node, its left subtree, its right subtree)
  std::vector<uint32_t> new_of(n_nodes, 0);
  {
    std::vector<uint32_t> stack{1};
    uint32_t next = 1;
    while (!stack.empty()) {
      const uint32_t q = stack.back();
      stack.pop_back();
      if (q == 0 || q >= n_nodes || new_of[q] != 0 || next >= n_nodes) return false;  // not a tree: never seen, but never trusted
      new_of[q] = next++;
      if (bfs[q].rcount == 0) stack.push_back(bfs[q].right);
      if (bfs[q].lcount == 0) stack.push_back(bfs[q].left);
    }
    if (next != n_nodes) return false;
  }
  tb.nodes.assign(n_nodes, rtaccel::Node{});
  for (uint32_t q = 1; q < n_nodes; q++) {
    rtaccel::Node nd = bfs[q];
    if (nd.lcount == 0) nd.left = new_of[nd.left];
    if (nd.rcount == 0) nd.right = new_of[nd.right];
    tb.nodes[new_of[q]] = nd;
  }
  for (auto& t : th) t.join();
  stage("download, pre-order, leaf triangles");
  tb.depth_seen = depth_seen;
  if (depth_seen > rtaccel::kMaxDepth) return false;
  // the header (TreeBuilder::finish)
  auto h2f = [](int i) {
    const int b = i >= 0 ? i : i ^ 0x7FFFFFFF;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
  };
  rtaccel::Node& h = tb.nodes[0];
  std::memset(&h, 0, sizeof(h));
  float r2 = 0.f;
  for (int a = 0; a < 3; a++) {
    const float blo = h2f(g.blo[a]), bhi = h2f(g.bhi[a]);
    h.lmin[a] = 0.5f * (blo + bhi);
    const float half = 0.5f * (bhi - blo);
    r2 += half * half;
  }
  const float radius = std::sqrt(r2) * 1.0001f, lmax = h2f(g.lmax), amax = h2f(g.amax);
  std::memcpy(&h.left, &radius, 4);
  h.lmax[0] = lmax;
  h.lmax[1] = amax;
  return std::isfinite(radius) && std::isfinite(lmax) && std::isfinite(amax);
}

}  // namespace rtaccel_dev
