// Shared device helpers for the gfx950 hot-path kernels (fp64 / complex128).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "ssspy_amd.h"

namespace ssspy {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double2 c128;  // x = re, y = im; layout-compatible with numpy.complex128

// ---------------------------------------------------------------- error plumbing
extern thread_local char g_last_error[512];

inline int fail(int code, const char *msg) {
  std::snprintf(g_last_error, sizeof(g_last_error), "%s", msg);
  return code;
}

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return SSSPY_ERR_HIP;
  }
  return SSSPY_OK;
}

#define SSSPY_REQUIRE(cond, msg) \
  do {                           \
    if (!(cond)) return ::ssspy::fail(SSSPY_ERR_BADARG, msg); \
  } while (0)


// switch on a runtime n_sources to a compile-time NN (1..8)
// (the same for kernels that only exist up to 4 sources: above, a bin is spread over lanes)
#define DISPATCH_N4(N_, CALL)                                                           \
  switch (N_) {                                                                         \
    case 1: { constexpr int NN = 1; CALL; } break;                                      \
    case 2: { constexpr int NN = 2; CALL; } break;                                      \
    case 3: { constexpr int NN = 3; CALL; } break;                                      \
    case 4: { constexpr int NN = 4; CALL; } break;                                      \
    default: return ::ssspy::fail(SSSPY_ERR_UNSUPPORTED, "n_sources must be in [1, 4]"); \
  }

// (kernels that exist up to 6 x 6: from 7 x 7 on a matrix is spread over 8 lanes, herm_rows8.hpp)
#define DISPATCH_N6(N_, CALL)                                                           \
  switch (N_) {                                                                         \
    case 1: { constexpr int NN = 1; CALL; } break;                                      \
    case 2: { constexpr int NN = 2; CALL; } break;                                      \
    case 3: { constexpr int NN = 3; CALL; } break;                                      \
    case 4: { constexpr int NN = 4; CALL; } break;                                      \
    case 5: { constexpr int NN = 5; CALL; } break;                                      \
    case 6: { constexpr int NN = 6; CALL; } break;                                      \
    default: return ::ssspy::fail(SSSPY_ERR_UNSUPPORTED, "size must be in [1, 6] here"); \
  }

#define DISPATCH_N(N_, CALL)                                                            \
  switch (N_) {                                                                         \
    case 1: { constexpr int NN = 1; CALL; } break;                                      \
    case 2: { constexpr int NN = 2; CALL; } break;                                      \
    case 3: { constexpr int NN = 3; CALL; } break;                                      \
    case 4: { constexpr int NN = 4; CALL; } break;                                      \
    case 5: { constexpr int NN = 5; CALL; } break;                                      \
    case 6: { constexpr int NN = 6; CALL; } break;                                      \
    case 7: { constexpr int NN = 7; CALL; } break;                                      \
    case 8: { constexpr int NN = 8; CALL; } break;                                      \
    default: return ::ssspy::fail(SSSPY_ERR_UNSUPPORTED, "n_sources must be in [1, 8]"); \
  }

// ---------------------------------------------------------------- complex arithmetic
__device__ __forceinline__ c128 cmake(double re, double im) { return make_double2(re, im); }

// Workgroup id -> work item such that consecutive items run on the same XCD.  The dispatcher is
// observed to place workgroup id on XCD id % 8 (MI355X_MICROARCH.md, "Workgroup dispatch"; a speed
// assumption only): ids {x, x + 8, x + 16, ...} get the contiguous item range of XCD x, so the
// operand every item of a mixture shares (its activation / basis tile, 0.26 - 0.5 MB) is pulled
// into one L2 instead of eight.  Bijective on [0, n) for any n.
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
  const int per = n >> 3, rem = n & 7;
  const int xcd = id & 7, slot = id >> 3;
  return xcd * per + min(xcd, rem) + slot;
}

// the same for a 3-D grid (x fastest): the XCD-contiguous item as (x, y, z)
struct GridItem {
  int x, y, z;
};
__device__ __forceinline__ GridItem xcd_contiguous_grid() {
  const int gx = gridDim.x, gy = gridDim.y;
  const int item = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z),
                                  gx * gy * gridDim.z);
  GridItem g;
  g.x = item % gx;
  g.y = (item / gx) % gy;
  g.z = item / (gx * gy);
  return g;
}

// Two consecutive doubles of a real row, as one 16-byte load that only needs 8-byte alignment (rows
// of an odd length start on odd multiples of 8 bytes; gfx950 global loads are dword-aligned).  The
// second element is read only when it belongs to the row: (p[0], p[1]) if j + 1 < len,
// (p[0], 0) if j + 1 == len, zeros beyond.
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ double2 load_pair_in_row(const double *__restrict__ row, int j, int len) {
  double2 val = make_double2(0.0, 0.0);
  if (j + 1 < len) {
    const double2_a8 v = *reinterpret_cast<const double2_a8 *>(row + j);
    val = make_double2(v.x, v.y);
  } else if (j < len) {
    val.x = row[j];
  }
  return val;
}
__device__ __forceinline__ c128 cadd(c128 a, c128 b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c128 csub(c128 a, c128 b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c128 cconj(c128 a) { return cmake(a.x, -a.y); }
__device__ __forceinline__ c128 cmul(c128 a, c128 b) {
  return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ c128 cmulc(c128 a, c128 b) {
  return cmake(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// acc += a * b
__device__ __forceinline__ void cfma(c128 &acc, c128 a, c128 b) {
  acc.x = fma(a.x, b.x, acc.x);
  acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y);
  acc.y = fma(a.y, b.x, acc.y);
}
// acc -= a * b
__device__ __forceinline__ void cfms(c128 &acc, c128 a, c128 b) {
  acc.x = fma(-a.x, b.x, acc.x);
  acc.x = fma(a.y, b.y, acc.x);
  acc.y = fma(-a.x, b.y, acc.y);
  acc.y = fma(-a.y, b.x, acc.y);
}
__device__ __forceinline__ c128 cscale(c128 a, double s) { return cmake(a.x * s, a.y * s); }
__device__ __forceinline__ double cabs2(c128 a) { return fma(a.x, a.x, a.y * a.y); }
__device__ __forceinline__ double cabs1(c128 a) { return fabs(a.x) + fabs(a.y); }
__device__ __forceinline__ c128 crecip(c128 a) {
  // Smith's algorithm: no overflow for well-scaled inputs, ~1 ulp
  if (fabs(a.x) >= fabs(a.y)) {
    double r = a.y / a.x;
    double d = 1.0 / fma(a.y, r, a.x);
    return cmake(d, -r * d);
  } else {
    double r = a.x / a.y;
    double d = 1.0 / fma(a.x, r, a.y);
    return cmake(r * d, -d);
  }
}
__device__ __forceinline__ c128 cdiv(c128 a, c128 b) { return cmul(a, crecip(b)); }
// 1 / a by v_rcp_f64 + two Newton steps on |a|^2 (~1 ulp) for the pivots of the row-distributed
// solvers (k_ip1_rows, k_ip2_rows): those kernels are chains of dependent pivots, and Smith's form
// above is two IEEE divides (~12 dependent instructions each) and a branch per reciprocal.  For
// well-scaled arguments only: |a|^2 must neither overflow nor underflow (statistics of spectrograms;
// the latency form k_ip1_small has used the same since round 3).
__device__ __forceinline__ c128 crecip_fast(c128 a) {
  const double m2 = fma(a.x, a.x, a.y * a.y);
  double r = __builtin_amdgcn_rcp(m2);
  double e = fma(-m2, r, 1.0);
  r = fma(r, e, r);
  e = fma(-m2, r, 1.0);
  r = fma(r, e, r);
  return cmake(a.x * r, -a.y * r);
}

__device__ __forceinline__ double apply_floor(double x, int kind, double eps) {
  // numpy.maximum propagates NaN; fmax would swallow it, so spell the compare out
  if (kind == SSSPY_FLOOR_MAX) return (x < eps) ? eps : x;
  if (kind == SSSPY_FLOOR_ADD) return x + eps;
  return x;
}

// x^e with the exponents the MM updates use; domain == 2 takes the pow-free path
__device__ __forceinline__ double pow_fast(double x, double e) { return pow(x, e); }

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum over the block; result valid in thread 0.  scratch: >= blockDim.x/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double total = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) total += scratch[w];
  }
  return total;
}


// ---------------------------------------------------------------- deterministic cross-block sums
// Sums that cross workgroups without fp64 atomics (whose order, and so whose low bits, change from
// run to run): every producer block stores its partial sums to its own slab, and k_fold_slabs adds
// the slabs of a reduction set IN SLAB ORDER.  A set can have hundreds of slabs (a single-mixture
// ISS sweep: 513 blocks), so the fold has two levels inside one launch: block (x, g) adds the <= 32
// slabs of group g for 256 outputs, and the block that draws the last ticket of column x adds the
// groups' results -- whoever is last, the order (and the bits) are the same.  (Tried first: the
// ticket and the fold inside the producing kernel -- one agent-scope release per producer block cost
// the fused ISS sweep 2x at 128 mixtures; in the fold kernel it is one per 32 slabs.)
// Visibility follows the release / acquire recipe of cdna_hip_programming.md (split-K reduction):
// plain stores -> s_waitcnt vmcnt(0) -> barrier -> lane 0: agent-scope release, s_waitcnt vmcnt(0)
// (restated in asm: hipcc may drop it), relaxed agent-scope ticket; the last block: agent-scope
// acquire -> barrier -> plain loads.  The counters are zeroed by the host before every launch.
constexpr int FOLD_GROUP = 32;

// true (in every thread) in the block that drew the last of `members` tickets.  `flag`: one LDS word.
// Every thread of the block must call it.
__device__ __forceinline__ bool last_arriver(unsigned *counter, unsigned members, int *flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket =
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == members - 1u;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *flag = last ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}

// sum of count values src[k * stride], k in order, eight loads in flight per round trip.
// V: double (one output per thread) or double2 (two; 16-byte loads, stride in units of V)
template <typename V>
__device__ __forceinline__ V ordered_sum(const V *src, long long stride, int count) {
  V s = V{};
  for (int k0 = 0; k0 < count; k0 += 8) {
    V v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)min(k0 + u, count - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < count) s += v[u];  // count is uniform: a scalar branch, not a select per lane
  }
  return s;
}

__host__ __device__ inline int fold_groups(int nslabs) {
  return (nslabs + FOLD_GROUP - 1) / FOLD_GROUP;
}

// out[e] = sum_k slabs[k * total + e] (k < nslabs, in order), e < total (in units of V).
// grid: (ceil(total / 256), fold_groups(nslabs)); gslabs: groups * total V of scratch;
// counters: gridDim.x zeroed words.
// accumulate: out[e] += the sum (one writer per element: still the same bits on every run)
template <typename V>
static __global__ __launch_bounds__(256) void k_fold_slabs(const V *__restrict__ slabs, V *gslabs,
                                                           unsigned *counters, V *out,
                                                           long long total, int nslabs,
                                                           int accumulate) {
  __shared__ int flag;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = e < total;
  const int ng = gridDim.y, g = blockIdx.y;
  const int members = min(FOLD_GROUP, nslabs - g * FOLD_GROUP);
  const V s = live ? ordered_sum(slabs + (long long)g * FOLD_GROUP * total + e, total, members) : V{};
  if (ng == 1) {
    if (live) {
      V r = s;
      if (accumulate) r += out[e];
      out[e] = r;
    }
    return;
  }
  if (live) gslabs[(long long)g * total + e] = s;
  if (!last_arriver(counters + blockIdx.x, (unsigned)ng, &flag)) return;
  if (live) {
    V r = ordered_sum(gslabs + e, total, ng);
    if (accumulate) r += out[e];
    out[e] = r;
  }
}

// host side: scratch behind the slabs and the launch.  `scratch` holds fold_scratch_bytes(...).
static inline size_t fold_scratch_bytes(long long total, int nslabs) {
  const int ng = fold_groups(nslabs);
  if (ng <= 1) return 0;
  return (size_t)ng * total * sizeof(double) + (size_t)((total + 255) / 256) * sizeof(unsigned);
}
static inline int launch_fold_slabs(const double *slabs, void *scratch, double *out, long long total,
                             int nslabs, hipStream_t st, int accumulate = 0) {
  const int ng = fold_groups(nslabs);
  double *gslabs = (double *)scratch;
  unsigned *counters = ng > 1 ? (unsigned *)(gslabs + (size_t)ng * total) : nullptr;
  // two outputs per thread through 16-byte loads when the layout allows (same sums, same order)
  const bool pairs = total % 2 == 0 &&
                     (((uintptr_t)slabs | (uintptr_t)gslabs | (uintptr_t)out) & 15u) == 0;
  const long long units = pairs ? total / 2 : total;
  const unsigned gx = (unsigned)((units + 255) / 256);
  if (ng > 1) {
    hipError_t e = hipMemsetAsync(counters, 0, (size_t)gx * sizeof(unsigned), st);
    if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  }
  if (pairs)
    hipLaunchKernelGGL(k_fold_slabs<double2>, dim3(gx, ng), dim3(256), 0, st, (const double2 *)slabs,
                       (double2 *)gslabs, counters, (double2 *)out, units, nslabs, accumulate);
  else
    hipLaunchKernelGGL(k_fold_slabs<double>, dim3(gx, ng), dim3(256), 0, st, slabs, gslabs, counters,
                       out, units, nslabs, accumulate);
  return check_launch("k_fold_slabs");
}

// ---- per-mixture scalars (loss terms) without atomics: contributor `slot` of mixture b stores its
// share to slots[slot * B + b] (every contributor its own slot; slots nobody writes stay zero), the
// launcher zeroes the slots first and folds them in slot order afterwards.
static inline size_t scalar_slots_bytes(int B, int nslots) {
  const size_t a = ((size_t)nslots * B * sizeof(double) + 255) & ~(size_t)255;
  return a + ((fold_scratch_bytes(B, nslots) + 255) & ~(size_t)255);
}
static inline int scalar_slots_begin(void *ws, int B, int nslots, hipStream_t st) {
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)nslots * B * sizeof(double), st);
  return e == hipSuccess ? SSSPY_OK : fail(SSSPY_ERR_HIP, hipGetErrorString(e));
}
// out[b] (+)= sum over slots
static inline int scalar_slots_fold(void *ws, int B, int nslots, double *out, int accumulate,
                                    hipStream_t st) {
  const size_t a = ((size_t)nslots * B * sizeof(double) + 255) & ~(size_t)255;
  return launch_fold_slabs((const double *)ws, (char *)ws + a, out, B, nslots, st, accumulate);
}

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  // v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
  // C/D: col = lane&15, row = (lane>>4) + 4*reg
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------- buffer (SRD) addressing
// A row of a tensor addressed as (wave-uniform base in SGPRs) + (32-bit per-lane byte offset): the
// only per-lane address state is the offset, where flat global loads keep a 64-bit address per
// access alive in VGPRs (the register-resident ISS slab kernel has none to spare).  Build the
// descriptor from wave-uniform values only.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ c128 buffer_load_c128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return cmake(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
}
__device__ __forceinline__ double buffer_load_f64(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
  return __hiloint2double((int)v[1], (int)v[0]);
}
__device__ __forceinline__ void buffer_store_f64(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double x) {
  u32x2_t v;
  v[0] = (unsigned)__double2loint(x);
  v[1] = (unsigned)__double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, 0);
}
__device__ __forceinline__ void buffer_store_c128(__amdgpu_buffer_rsrc_t r, unsigned byte_off, c128 z) {
  u32x4_t v;
  v[0] = (unsigned)__double2loint(z.x);
  v[1] = (unsigned)__double2hiint(z.x);
  v[2] = (unsigned)__double2loint(z.y);
  v[3] = (unsigned)__double2hiint(z.y);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0);
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace ssspy
