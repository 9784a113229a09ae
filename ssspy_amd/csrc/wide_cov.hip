// Weighted covariance of 5..8-channel mixtures on the f64 matrix cores.
//
//   U[b, i, s] = (1/T) sum_j phi[s, (i,) j] x_ij x_ij^H        x_ij in C^N, N <= 8
//
// (ssspy/bss/_update_spatial_model.py:60-83 and the U of every IP update above 4 channels.)
// A per-lane N x N accumulator (cov_core.hpp) needs N (N + 1) registers per weight set: 72 doubles at
// N = 8, two sets at most, and the generic kernel spends its time spilling (2.1 ms per pass for 16
// mixtures of N = 8, F = 1025, T = 512 -- 11 % of the fp64 rate).  Here the outer product is a real
// 16 x 16 x 4 MFMA: with x~ = [Re x_0..7 ; Im x_0..7] (rows past N are zero),
//
//   D = sum_j phi_j x~_j x~_j^T = [ RR  RI ; IR  II ],   U = (RR + II) + i (IR - RI),
//
// so a (bin, weight set) is one 16 x 16 accumulator tile (8 VGPRs per lane), a wave owns one bin and
// all weight sets (64 VGPRs at S = 8) and walks the frames eight at a time: one 16-byte load per lane
// brings 8 channels x 8 frames (whole 128-byte lines), a DPP rotate by 8 inside each row of 16 lanes
// pairs the real part of one frame with the imaginary part of the same frame held by the partner
// lane, and 2 S MFMAs consume the slab.  No barrier, no cross-wave reduction; the weights pass
// through a wave-private LDS patch (see ZRING / WRING).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

namespace ssspy {

namespace {

// value of lane (l ^ 8) within each row of 16 lanes (row_ror:8)
__device__ __forceinline__ double partner8(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Loads in flight per wave.  Rounds 4-5 kept two slabs (x and the weights of all sets as 16-byte
// loads that every lane of a 16-lane row repeated: 36 registers a slab) -- 16 KB in flight per CU,
// 2 TB/s by Little's law, and 8 channels x 8 mixtures of 513 x 256 took 113 us for 201 MB.  Round 6:
// the x ring is ZRING slabs deep (4 registers each); the weights of TWO slabs and all sets come as ONE
// 16-byte load per lane (lane = (set, frame pair): 1 KB without repeats), WRING pairs deep, and reach
// the lanes that multiply with them through a wave-private 1 KB LDS patch.
constexpr int ZRING = 8, WRING = 4;
static_assert(ZRING == 2 * WRING, "a weight load covers two slabs");

// grid: ceil(B F / 4); wave w of a block owns (mixture, bin) item 4 blockIdx.x + w.
// weight: FRAME (B, S, T), BIN_FRAME (B, S, F, T), UNIT none.  U: (B, F, S, N, N).
template <int NS, int MODE>
__global__ __launch_bounds__(256, 2) void k_wide_cov(const c128 *__restrict__ A,
                                                  const double *__restrict__ weight,
                                                  c128 *__restrict__ U, int N, int F, int T,
                                                  long long nitems) {
  constexpr bool WEIGHTED = MODE != SSSPY_WEIGHT_UNIT;
  __shared__ __attribute__((aligned(16))) double2 wpatch[WEIGHTED ? 4 : 1][2][64];  // [set * 8 + pair]
  // (readfirstlane: the compiler cannot see that threadIdx.x >> 6 is wave-uniform, and descriptors
  // built from "divergent" values put every buffer load inside a waterfall loop)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long item = (long long)blockIdx.x * 4 + wave;
  if (item >= nitems) return;
  const int b = (int)(item / F), bin = (int)(item - (long long)b * F);
  const int a = lane & 7, h = (lane >> 3) & 1, k = lane >> 4;
  // one descriptor over the mixture: channels past N and frames past the tensor read as zero
  const __amdgpu_buffer_rsrc_t xr =
      make_rsrc(A + (long long)b * N * F * T, (unsigned)N * (unsigned)F * (unsigned)T * 16u);
  const unsigned xbase = (((unsigned)a * (unsigned)F + (unsigned)bin) * (unsigned)T + 2u * k + h) * 16u;
  const __amdgpu_buffer_rsrc_t wr =
      MODE == SSSPY_WEIGHT_FRAME
          ? make_rsrc(weight + (long long)b * NS * T, (unsigned)NS * (unsigned)T * 8u)
          : make_rsrc(weight + (long long)b * NS * F * T,
                      (unsigned)NS * (unsigned)F * (unsigned)T * 8u);
  const unsigned wrow = MODE == SSSPY_WEIGHT_FRAME ? (unsigned)T : (unsigned)F * (unsigned)T;
  // lane (set = lane & 7, pair = lane >> 3) of a weight load: frames 16 sp + 2 pair, + 1 of its set;
  // sets past NS lie behind the descriptor's range (zeros, never read back)
  const unsigned wbase = ((MODE == SSSPY_WEIGHT_FRAME ? 0u : (unsigned)bin * (unsigned)T) +
                          2u * (unsigned)(lane >> 3) + (unsigned)(lane & 7) * wrow) * 8u;
  const int nslabs = (T + 7) >> 3, npairs = (nslabs + 1) >> 1;
  auto loadz = [&](u32x4_t &z, const int s) __attribute__((always_inline)) {
    z = __builtin_amdgcn_raw_buffer_load_b128(xr, xbase + 128u * min(s, nslabs - 1), 0, 0);
  };
  auto loadw = [&](u32x4_t &w, const int sp) __attribute__((always_inline)) {
    if (WEIGHTED) w = __builtin_amdgcn_raw_buffer_load_b128(wr, wbase + 128u * min(sp, npairs - 1), 0, 0);
  };
  double4_t acc[NS];
#pragma unroll
  for (int n = 0; n < NS; ++n) acc[n] = double4_t{0.0, 0.0, 0.0, 0.0};
  // one slab into the accumulators; TAIL masks the frames past T (the last slab only)
  auto consume = [&](const u32x4_t &zq, const int s, auto tail) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail)::value;
    const c128 z = cmake(__hiloint2double((int)zq[1], (int)zq[0]), __hiloint2double((int)zq[3], (int)zq[2]));
    // lane (a, h, k) holds frame t1 + h of channel a; the tile rows are [Re ; Im] of ONE frame
    const double sent = h ? z.x : z.y;
    const double got = partner8(sent);
    const int t1 = 8 * s + 2 * k;
    const bool ok1 = !TAIL || t1 < T, ok2 = !TAIL || t1 + 1 < T;
    // frames past T were loaded from whatever follows the row: zero the samples themselves, not
    // only their weights (0 * Inf would put a NaN of a neighbouring row into this bin)
    const double v1 = ok1 ? (h ? got : z.x) : 0.0;  // row a + 8 h of frame t1 = 8 s + 2 k
    const double v2 = ok2 ? (h ? z.y : got) : 0.0;  // ... of frame t1 + 1
    double2 w[NS];
    if (WEIGHTED) {
      const double2 *wp = &wpatch[wave][(s >> 1) & 1][4 * (s & 1) + k];
#pragma unroll
      for (int n = 0; n < NS; ++n) w[n] = wp[8 * n];
    }
    // (the two updates of one accumulator are NS MFMAs apart)
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const double p1 = ok1 ? (WEIGHTED ? w[n].x : 1.0) : 0.0;
      acc[n] = mfma_f64(p1 * v1, v1, acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const double p2 = ok2 ? (WEIGHTED ? w[n].y : 1.0) : 0.0;
      acc[n] = mfma_f64(p2 * v2, v2, acc[n]);
    }
  };
  u32x4_t zr[ZRING], wq[WRING];
#pragma unroll
  for (int u = 0; u < ZRING; ++u) loadz(zr[u], u);
#pragma unroll
  for (int u = 0; u < WRING; ++u) loadw(wq[u], u);
  for (int s0 = 0; s0 < nslabs; s0 += ZRING) {
#pragma unroll
    for (int pp = 0; pp < WRING; ++pp) {
      // park the pair's weights (the refills are unconditional, clamped to the last slab / pair:
      // loads inside branches make the compiler drain the whole queue at every join)
      const int sp = (s0 >> 1) + pp;
      if (WEIGHTED) {
        wave_lds_fence();
        wpatch[wave][pp & 1][(lane & 7) * 8 + (lane >> 3)] =
            make_double2(__hiloint2double((int)wq[pp][1], (int)wq[pp][0]),
                         __hiloint2double((int)wq[pp][3], (int)wq[pp][2]));
        wave_lds_fence();
      }
      loadw(wq[pp], sp + WRING);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = s0 + 2 * pp + u;
        if (8 * s + 8 <= T) consume(zr[2 * pp + u], s, std::false_type{});
        else if (s < nslabs) consume(zr[2 * pp + u], s, std::true_type{});
        loadz(zr[2 * pp + u], s + ZRING);
      }
    }
  }
  // D[row = k + 4 r][col = lane & 15]: rows / cols 0..7 real parts, 8..15 imaginary parts
  const double scale = 1.0 / (double)T;
  const int col = lane & 15;
#pragma unroll
  for (int n = 0; n < NS; ++n) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double re = acc[n][r] + partner8(acc[n][r + 2]);  // RR + II
      const int row = k + 4 * r;
      // IR - RI; the diagonal is real (its two sums differ by rounding only)
      const double im = row == col ? 0.0 : acc[n][r + 2] - partner8(acc[n][r]);
      if (col < N && row < N)
        U[(((long long)b * F + bin) * NS + n) * (N * N) + row * N + col] =
            cmake(re * scale, im * scale);
    }
  }
}

template <int NS>
int launch(const c128 *A, const double *weight, int kind, c128 *U, int B, int N, int F, int T,
           hipStream_t st) {
  const long long nitems = (long long)B * F;
  dim3 grid((unsigned)((nitems + 3) / 4)), block(256);
  switch (kind) {
    case SSSPY_WEIGHT_FRAME:
      hipLaunchKernelGGL((k_wide_cov<NS, SSSPY_WEIGHT_FRAME>), grid, block, 0, st, A, weight, U, N,
                         F, T, nitems);
      break;
    case SSSPY_WEIGHT_BIN_FRAME:
      hipLaunchKernelGGL((k_wide_cov<NS, SSSPY_WEIGHT_BIN_FRAME>), grid, block, 0, st, A, weight, U,
                         N, F, T, nitems);
      break;
    default:
      return fail(SSSPY_ERR_BADARG, "wide_weighted_cov: unknown weight_kind");
  }
  return check_launch("k_wide_cov");
}

}  // namespace

// whether wide_weighted_cov() takes the shape (else the caller keeps the generic kernel)
bool wide_weighted_cov_ok(int N, int S, int F, int T, int kind) {
  static const bool disabled = std::getenv("SSSPY_AMD_NO_FAST") != nullptr;
  // (5 channels pay for the rows the 16 x 16 tile pads: level with the per-lane kernel, which stays)
  if (disabled || N < 6 || N > 8 || S < 1 || S > 8) return false;
  if ((long long)N * F * T * 16 >= (1ll << 32)) return false;
  if (kind == SSSPY_WEIGHT_UNIT) return S == 1;
  return kind == SSSPY_WEIGHT_FRAME || kind == SSSPY_WEIGHT_BIN_FRAME;
}

int wide_weighted_cov(const void *A, const double *weight, int kind, void *U, int B, int N, int S,
                      int F, int T, hipStream_t st) {
  const c128 *a = (const c128 *)A;
  c128 *u = (c128 *)U;
  if (kind == SSSPY_WEIGHT_UNIT) {
    const long long nitems = (long long)B * F;
    hipLaunchKernelGGL((k_wide_cov<1, SSSPY_WEIGHT_UNIT>), dim3((unsigned)((nitems + 3) / 4)),
                       dim3(256), 0, st, a, weight, u, N, F, T, nitems);
    return check_launch("k_wide_cov");
  }
  switch (S) {
    case 1: return launch<1>(a, weight, kind, u, B, N, F, T, st);
    case 2: return launch<2>(a, weight, kind, u, B, N, F, T, st);
    case 3: return launch<3>(a, weight, kind, u, B, N, F, T, st);
    case 4: return launch<4>(a, weight, kind, u, B, N, F, T, st);
    case 5: return launch<5>(a, weight, kind, u, B, N, F, T, st);
    case 6: return launch<6>(a, weight, kind, u, B, N, F, T, st);
    case 7: return launch<7>(a, weight, kind, u, B, N, F, T, st);
    case 8: return launch<8>(a, weight, kind, u, B, N, F, T, st);
  }
  return fail(SSSPY_ERR_UNSUPPORTED, "wide_weighted_cov: more than 8 weight sets");
}

}  // namespace ssspy
