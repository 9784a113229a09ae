// Latency kernels of the ILRMA-IP1 iteration for a HANDFUL of mixtures (the reference's own use:
// one mixture per call; BASELINE configs[1] literally).
//
// With one mixture the iteration is a chain of short dependent kernels on a chip that holds the
// whole problem at once, and what each costs is (i) ~5 us of launch / drain whatever it does,
// (ii) the time to stream X once (33.6 MB: every wave's LAST load returns after ~7 us, whatever the
// grid shape -- measured with per-wave phase stamps, profiles/r03_single_mixture.md) and (iii) the
// length of its dependent instruction chain; throughput rules do not apply.  Measured per kernel
// (rocprofv3, one mixture of the configs[1] shape) against the throughput kernels of ilrma_fast.hip:
//   * activation pass: one 16 x 16 MFMA tile per WAVE (four waves = four bin tiles of the same 16
//     frames, every operand requested before anything else, demixing matrices / basis rows parked in
//     a wave-private LDS patch, |y|^2 of all sources formed first so the x tile is dead before the
//     GEMMs, accumulators folded through LDS: 16 partial sums per frame tile instead of 16 per wave):
//     25.8 + 7.1 us (pass + fold) -> 21.6 + 5.4 us;
//   * covariance fold + IP1 + output power in ONE kernel: the 16 lanes of a bin sum its partial
//     records into LDS, then four lanes per bin run the N sequential projections out of LDS with
//     the LU solve row-distributed and reciprocals by v_rcp_f64 + Newton: 5.5 + 16.9 us -> 12.0 us
//     (one lane per bin out of LDS: 44 us -- the chain of ~5000 dependent fp64 instructions is what
//     costs, not the loads);
//   * normalisation with every load issued before the reduction: 14.8 -> 6.0 us.
// The basis and covariance passes keep the throughput kernels: the same one-tile-per-wave form
// measured 27 and 40 us against their 21 + 7 and 17 us (65 bin tiles x 8 chunks is 8 blocks more
// than the chip holds at <= 256 registers, and the covariance accumulators do not fit that budget).
// Same math and tile layouts as ilrma_fast.hip; the summation order differs, so results agree with
// the throughput kernels to rounding (1e-13).
// Compiled once per N (-DSSSPY_N=2..4).
#include <cstdlib>

#include "common.hpp"
#include "cov_core.hpp"
#include "fast_model.hpp"
#include "fast_tiles.hpp"
#include "smallmat.hpp"

#ifndef SSSPY_N
#error "compile with -DSSSPY_N=<n_sources>"
#endif
#if SSSPY_N > 4
#error "small-batch path is built for n_sources <= 4"
#endif

#define SSSPY_CAT_(a, b) a##b
#define SSSPY_CAT(a, b) SSSPY_CAT_(a, b)
#define LAUNCHER(name) SSSPY_CAT(SSSPY_CAT(name, _n), SSSPY_N)

namespace ssspy {
namespace SSSPY_CAT(ilrma_small_n, SSSPY_N) {

constexpr int N = SSSPY_N;
using fast::FastModel;
using fast::FM_GAUSS;
using fast::FM_GAUSS1;
using fast::FM_GAUSSP;
using fast::FM_GGD;
using fast::FM_T;
using fast::mm_num_factor;
using fast::rcp_nr;
typedef fast::XTile<N> XTile;

// (num / den)^expo of the multiplicative updates.  The square root (Gauss, t at domain 2) and the
// plain ratio (ME) stay inline; the general power is a call: an inlined fp64 pow is ~2 KB of code
// per site, and these kernels run every instruction ONCE per wave from a cold instruction cache.
__device__ __attribute__((noinline)) double pow_call(double x, double e) { return pow(x, e); }
__device__ __forceinline__ double ratio_pow(double ratio, double expo) {
  if (expo == 0.5) return sqrt(ratio);
  if (expo == 1.0) return ratio;
  return pow_call(ratio, expo);
}

constexpr int WSTRIDE = N * N + 1;  // 16-byte slots per staged demixing matrix (odd: no bank clash)
constexpr int NV = 8 * N;           // accumulator doubles per lane: (num, den) x 4 registers x N

// (num, den) accumulators of the four waves -> LDS -> every thread sums four (source, register,
// lane) pairs over the waves.  `fold` holds 4 * NV * 64 doubles.
__device__ __forceinline__ void fold_store(double *fold, int wave, int lane, const double4_t (&num)[N],
                                           const double4_t (&den)[N]) {
  double *mine = fold + (size_t)wave * NV * 64 + lane;
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[((n * 4 + r) * 2) * 64] = num[n][r];
      mine[((n * 4 + r) * 2 + 1) * 64] = den[n][r];
    }
}
__device__ __forceinline__ void fold_sum(const double *fold, int nr, int ln, double &sn, double &sd) {
  sn = 0.0;
  sd = 0.0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    sn += fold[((size_t)w * NV + nr * 2) * 64 + ln];
    sd += fold[((size_t)w * NV + nr * 2 + 1) * 64 + ln];
  }
}

// The exchange through a wave-private LDS patch is between lanes of ONE wave: no barrier, but the
// per-thread memory model does not order it, so the wave's outstanding LDS reads are drained before
// a patch is rewritten and its writes before the patch is read (cf. fast_tiles.hpp, xtile_transpose).
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
}

// Pins a value to this program point: the arithmetic that produces it cannot be sunk towards its
// use (hipcc otherwise moves the whole |y|^2 phase below the reads of EVERY demixing row, keeps 256
// registers of rows alive and spills them).
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }

// ---- development aid (-DSSSPY_SMALL_TRACE, N = 4 only): per-wave phase stamps of the shader clock
#if defined(SSSPY_SMALL_TRACE) && SSSPY_N == 4
#define SMALL_TRACE 1
__device__ long long g_small_trace[8 * 4096];
__device__ __forceinline__ void stamp(long long (&ts)[8], int k, bool drain) {
  if (drain) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the phase has really ended
  ts[k] = __builtin_readcyclecounter();
}
#define STAMP(k, drain) stamp(ts_, k, drain)
#define STAMP_DECL long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP_FLUSH()                                                                         \
  do {                                                                                        \
    const int wid_ = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 +   \
                     (threadIdx.x >> 6);                                                      \
    if ((threadIdx.x & 63) == 0 && wid_ < 4096)                                               \
      for (int k_ = 0; k_ < 8; ++k_) g_small_trace[wid_ * 8 + k_] = ts_[k_];                  \
  } while (0)
#else
#define STAMP(k, drain)
#define STAMP_DECL
#define STAMP_FLUSH()
#endif

constexpr int VT_ROW = 17;                 // doubles per staged row (16 + 1 pad)
constexpr int VT_SIZE = N * 16 * VT_ROW;   // doubles per staged (n, k, frame) or (n, bin, k) tile

// ========================================================================= activation (pass 2)
// grid: (frame tiles, chunks, B), 256 threads.  Wave w of chunk ch walks the bin tiles
// ch * tpc + w, + 4, ... below (ch + 1) * tpc.  nchunks == 1: update in place; otherwise the sums go
// to part[(b * nchunks + ch)][n][num | den][k][frame] (the layout k_ilrma_activation_finalize folds).
template <bool HAS_W, int MODEL>
__global__ __launch_bounds__(256, 2) void k_activation_small(const c128 *__restrict__ X,
                                                             const c128 *__restrict__ W,
                                                             const double *__restrict__ basis,
                                                             double *act, int F, int T, int K,
                                                             int floor_kind, double eps, int tpc,
                                                             int nchunks, double *__restrict__ part,
                                                             FastModel fm) {
  // one buffer: per wave the staged demixing matrices and basis rows of its current bin tile during
  // the walk, the accumulators of the four waves afterwards
  __shared__ __attribute__((aligned(16))) double smem[4 * NV * 64];
  STAMP_DECL;
  STAMP(0, false);
  double *fold = smem;
  constexpr int WAVE_STAGE = 2 * 16 * WSTRIDE + N * 16 * 17;  // doubles: W patch + T rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const int j0 = blockIdx.x * 16;
  const int jf = j0 + c;
  const bool fvalid = jf < T;
  const int jc = fvalid ? jf : T - 1;
  const fast::XSrc<N> xsrc = fast::make_xsrc<N>(X + (long long)b * N * F * T, F, T);
  const double *basis_b = basis + (long long)b * N * F * K;
  const c128 *W_b = HAS_W ? W + (long long)b * F * N * N : nullptr;
  c128 *wmine = reinterpret_cast<c128 *>(smem + (size_t)wave * WAVE_STAGE);  // [bin 16][WSTRIDE]
  double *tmine = smem + (size_t)wave * WAVE_STAGE + 2 * 16 * WSTRIDE;       // [n][bin 16][17]
  const int ntiles = (F + 15) >> 4;
  // (the last chunk takes the remainder: 65 bin tiles = 15 chunks of 4 and one of 5)
  const int t_end = chunk == nchunks - 1 ? ntiles : (chunk + 1) * tpc;
  const int ksteps = (K + 3) >> 2;
  double *vt = smem + 4 * (size_t)WAVE_STAGE;
  double4_t numv[N], denv[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    numv[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    denv[n] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
  // every wave makes the same number of rounds (the staging barrier sits in the first one); a round
  // beyond the wave's last tile only fetches (a clamped tile) and is skipped
  const int rounds = (t_end - chunk * tpc + 3) >> 2;
  for (int u = 0; u < rounds; ++u) {
    const int it = chunk * tpc + wave + 4 * u;
    const bool live = it < t_end;  // wave-uniform
    const int i0 = min(it, t_end - 1) * 16;
    // the tile's operands, requested before anything else (the longest round trip): x (16 loads of
    // 16 bytes per lane), its demixing matrices and basis rows (coalesced; parked in the wave's LDS
    // patch below and read from there in the two MFMA operand layouts)
    XTile cur;
    // (four scalars, not an array: hipcc promoted the array to 16 KB of LDS, and the store into it
    // waited for the x loads issued before -- a second serial round trip)
    c128 wv0 = cmake(0.0, 0.0), wv1 = wv0, wv2 = wv0, wv3 = wv0;
    double tv[(N * 256 + 63) / 64];
    fast::xtile_load_framemajor<N>(cur, xsrc, T, i0, jc, q);
    constexpr int NWV = (16 * N * N + 63) / 64;
    auto wload = [&](const int v) __attribute__((always_inline)) -> c128 {
      const int e = lane + 64 * v;
      const int bl = e / (N * N), rem = e % (N * N);
      return W_b[(long long)min(i0 + min(bl, 15), F - 1) * (N * N) + rem];  // (parked if in range)
    };
    if (HAS_W) {
      wv0 = wload(0);
      if (NWV > 1) wv1 = wload(1);
      if (NWV > 2) wv2 = wload(2);
      if (NWV > 3) wv3 = wload(3);
    }
#pragma unroll
    for (int v = 0; v < (N * 256 + 63) / 64; ++v) {
      const int e = lane + 64 * v;  // (n, bin, k)
      const int k = e & 15, bl = (e >> 4) & 15, n = e >> 8;
      const double tval =
          basis_b[((long long)min(n, N - 1) * F + min(i0 + bl, F - 1)) * K + min(k, K - 1)];
      tv[v] = (n < N && k < K && i0 + bl < F) ? tval : 0.0;
    }
    if (u == 0) {
      // GEMM1 B operand V[n, 4 ks + q, frame j0 + c]: the activation tile of the workgroup's 16
      // frames, staged once behind the waves' patches
      for (int e = threadIdx.x; e < N * 256; e += 256) {
        const int f = e & 15, k = (e >> 4) & 15, n = e >> 8;
        const double vval =
            act[(((long long)b * N + n) * K + min(k, K - 1)) * T + min(j0 + f, T - 1)];
        vt[(e >> 4) * VT_ROW + f] = (k < K && j0 + f < T) ? vval : 0.0;
      }
      __syncthreads();
      STAMP(1, true);
    }
    if (!live) continue;
    wave_lds_fence();
    if (HAS_W) {
      auto wpark = [&](const int v, const c128 val) __attribute__((always_inline)) {
        const int e = lane + 64 * v;
        if (e < 16 * N * N) wmine[(e / (N * N)) * WSTRIDE + e % (N * N)] = val;
      };
      wpark(0, wv0);
      if (NWV > 1) wpark(1, wv1);
      if (NWV > 2) wpark(2, wv2);
      if (NWV > 3) wpark(3, wv3);
    }
#pragma unroll
    for (int v = 0; v < (N * 256 + 63) / 64; ++v) {
      const int e = lane + 64 * v;
      if (e < N * 256) tmine[(e >> 4) * 17 + (e & 15)] = tv[v];  // row = n * 16 + bin
    }
    wave_lds_fence();
    STAMP(2, true);
    double pw[N][4];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c128 y = cur.x[n][r];
        if (HAS_W) {
          const c128 *wr = wmine + (q + 4 * r) * WSTRIDE + n * N;
          y = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < N; ++m) cfma(y, wr[m], cur.x[m][r]);
        }
        pw[n][r] = cabs2(y);
        if (HAS_W) pin(pw[n][r]);
      }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3, true);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *tn = tmine + n * 16 * 17;
      // GEMM1: A[row = c -> bin i0 + c][kk = q] = T[n, i0 + c, 4 ks + q]
      double4_t R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < ksteps)
          R = mfma_f64(tn[c * 17 + 4 * ks + q], vt[(n * 16 + 4 * ks + q) * VT_ROW + c], R);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = fvalid && (i0 + q + 4 * r < F);
        const double rinv = rcp_nr(R[r]);
        const double bb = valid ? rinv : 0.0;
        const double aa = valid ? mm_num_factor<MODEL>(pw[n][r], R[r], rinv, fm) : 0.0;
        // GEMM2: A[row = c -> basis index c][kk = q] = T[n, bin i0 + q + 4 r, c]
        const double ta = tn[(q + 4 * r) * 17 + c];
        numv[n] = mfma_f64(ta, aa, numv[n]);
        denv[n] = mfma_f64(ta, bb, denv[n]);
      }
    }
  }
#ifdef SMALL_TRACE
  asm volatile("" : "+v"(numv[N - 1]), "+v"(denv[N - 1]));  // the last MFMAs have retired
#endif
  STAMP(4, true);
  __syncthreads();  // every wave is done with its patch: the buffer becomes the fold area
  STAMP(5, true);
  fold_store(fold, wave, lane, numv, denv);
  __syncthreads();
  // D: col = frame c, row = q + 4 r -> basis index q + 4 r
#pragma unroll
  for (int u = 0; u < (N * 4 * 64 + 255) / 256; ++u) {
    const int pidx = threadIdx.x + 256 * u;
    const int ln = pidx & 63, nr = pidx >> 6;
    if (nr >= N * 4) break;
    const int n = nr >> 2, r = nr & 3;
    const int k = (ln >> 4) + 4 * r, j = j0 + (ln & 15);
    double sn, sd;
    fold_sum(fold, nr, ln, sn, sd);
    if (k < K && j < T) {
      if (nchunks == 1) {
        double *dst = act + (((long long)b * N + n) * K + k) * T + j;
        *dst = apply_floor(ratio_pow(sn / sd, fm.expo) * (*dst), floor_kind, eps);
      } else {
        const long long base = ((((long long)b * nchunks + chunk) * N + n) * 2) * K;
        part[(base + k) * T + j] = sn;
        part[(base + K + k) * T + j] = sd;
      }
    }
  }
  STAMP(6, true);
  STAMP_FLUSH();
}

// act <- floor(act * (sum_chunks num / sum_chunks den)^expo); grid: (ceil(K T / 256), N, B)
__global__ __launch_bounds__(256) void k_activation_small_finalize(double *act,
                                                                   const double *__restrict__ part,
                                                                   int K, int T, int nchunks,
                                                                   int floor_kind, double eps,
                                                                   double expo) {
  const int b = blockIdx.z, n = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over K * T
  if (e >= (long long)K * T) return;
  double *dst = act + ((long long)b * N + n) * K * T + e;
  const double vold = *dst;
  double sn = 0.0, sd = 0.0;
  for (int ch0 = 0; ch0 < nchunks; ch0 += 8) {  // eight chunks per round trip
    double vn[8], vd[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long base =
          ((((long long)b * nchunks + min(ch0 + u, nchunks - 1)) * N + n) * 2) * K * T;
      vn[u] = part[base + e];
      vd[u] = part[base + (long long)K * T + e];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      sn += ch0 + u < nchunks ? vn[u] : 0.0;
      sd += ch0 + u < nchunks ? vd[u] : 0.0;
    }
  }
  *dst = apply_floor(ratio_pow(sn / sd, expo) * vold, floor_kind, eps);
}

// ============================================================ fold + IP1 + output power (pass 4)
// grid: (bin tiles, B), 256 threads = 16 bins x 16 lanes.  The lanes of a bin sum the N^3 entries of
// its covariance records over the chunks into LDS (coalesced 256-byte rows) and stage its demixing
// matrix and static covariance beside them; then wave 0 alone runs the N sequential projections
// (ssspy/bss/_update_spatial_model.py:63-76) and the output power w C w^H, FOUR LANES PER BIN: lane
// r owns row r of the filter and of A = W U_n, the LU solve is row-distributed as in k_ip1_rows
// (spatial_kernels.hip: pivot = largest |re| + |im| among the unused rows, lowest row on ties --
// LAPACK's choice; the pivot lane broadcasts its row; back substitution broadcasts one unknown per
// step).  What a single mixture pays for is the LENGTH of the dependent instruction chain: one lane
// per bin executes ~5000 dependent fp64 instructions (32 us measured), a quarter of the row
// arithmetic plus reciprocals by v_rcp_f64 + 2 Newton steps (~1 ulp, fast_tiles.hpp) instead of
// IEEE divides leave ~1/4 of that.
__device__ __forceinline__ c128 crecip_nr(c128 a) {
  const double inv = rcp_nr(fma(a.x, a.x, a.y * a.y));
  return cmake(a.x * inv, -a.y * inv);
}

// Input: the finished covariance U (B, F, N, N, N) (nchunks == 0), or the partial records of the
// split items of k_wcov_fast, upart[(b * groups + group) * nchunks + ch][rbins][N^3] (TailPlan with
// no unsplit item: a handful of mixtures), which are summed here instead of by k_wcov_fold.
// rec: c128 between consecutive records (rbins N^3 for k_wcov_fast; FastMNMF's records share a
// wider scratch slot, mnmf_tail_doubles()).
// logdet (optional): logdet[tile * logdet_stride + b] <- sum over the tile's bins of log|det W| of
// the filters AS THEY COME IN (the state the loss of the previous iteration describes,
// ssspy/bss/ilrma.py:1910-1967): wave 1, idle once the operands are staged, takes a lane per bin
// while wave 0 runs the projections -- the dedicated one-block kernel was 13 us + a launch gap per
// iteration of a one-mixture run with record_loss=True.
__global__ __launch_bounds__(256) void k_ip1_small(c128 *W, const c128 *__restrict__ Usrc,
                                                   const c128 *__restrict__ C, double *qbuf, int F,
                                                   int nchunks, int rbins, long long rec,
                                                   int floor_kind, double eps, int *info,
                                                   double *logdet, long long logdet_stride) {
  constexpr int NN = N * N, G = 4;
  __shared__ __attribute__((aligned(16))) c128 Us[16][N][NN + 1];
  __shared__ __attribute__((aligned(16))) c128 Ws[16][NN + 1];
  __shared__ __attribute__((aligned(16))) c128 Cs[16][NN + 1];
  const int tile = blockIdx.x, b = blockIdx.y;  // gridDim.x = bin tiles
  {
    const int bl = threadIdx.x >> 4, e = threadIdx.x & 15;
    const int bin = tile * 16 + bl;
    if (bin < F && e < NN) {
      const long long gb = (long long)b * F + bin;
      const c128 wv = W[gb * NN + e];
      const c128 cv = C ? C[gb * NN + e] : cmake(0.0, 0.0);
      double re[N], im[N];
      if (nchunks == 0) {
        const c128 *src = Usrc + gb * (N * NN) + e;
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const c128 v = src[n * NN];
          re[n] = v.x;
          im[n] = v.y;
        }
      } else {
        const int groups = (F + rbins - 1) / rbins, group = bin / rbins;
        const c128 *src = Usrc + ((long long)b * groups + group) * nchunks * rec +
                          (long long)(bin - group * rbins) * (N * NN) + e;
#pragma unroll
        for (int n = 0; n < N; ++n) re[n] = im[n] = 0.0;
        for (int ch0 = 0; ch0 < nchunks; ch0 += 8) {  // eight chunks per round trip
          c128 v[8][N];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < N; ++n) v[u][n] = src[(long long)min(ch0 + u, nchunks - 1) * rec + n * NN];
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < N; ++n) {
              re[n] += ch0 + u < nchunks ? v[u][n].x : 0.0;
              im[n] += ch0 + u < nchunks ? v[u][n].y : 0.0;
            }
        }
      }
      Ws[bl][e] = wv;
      Cs[bl][e] = cv;
#pragma unroll
      for (int n = 0; n < N; ++n) Us[bl][n][e] = cmake(re[n], im[n]);
    }
  }
  __syncthreads();
  if (logdet && threadIdx.x >= 64 && threadIdx.x < 128) {
    const int l = threadIdx.x - 64;
    double ld = 0.0;
    if (l < 16 && tile * 16 + l < F) {
      Mat<N> A;
#pragma unroll
      for (int rr = 0; rr < N; ++rr)
#pragma unroll
        for (int c = 0; c < N; ++c) A.a[rr][c] = Ws[l][rr * N + c];
      ld = logabsdet<N>(A);
    }
    // the 16 shares in lane order on lane 0 (fixed order: the loss list is the same on every run)
    double total = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) total += __shfl(ld, q, 64);
    if (l == 0) logdet[(long long)tile * logdet_stride + b] = total;
  }
  if (threadIdx.x >= 64) return;
  const int bl = threadIdx.x >> 2, r = threadIdx.x & 3;  // 16 bins x 4 lanes
  const int bin = tile * 16 + bl;
  const bool live = bin < F;
  const int bs = live ? bl : 0;  // idle groups shadow the tile's first bin, never store
  const bool row = r < N;
  const int rr = row ? r : N - 1;
  c128 Wr[N];
#pragma unroll
  for (int c = 0; c < N; ++c) Wr[c] = Ws[bs][rr * N + c];
  bool ok = true;
#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    const c128 *Un = &Us[bs][n][0];
    c128 a[N];
#pragma unroll
    for (int c = 0; c < N; ++c) a[c] = cmake(0.0, 0.0);
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int c = 0; c < N; ++c) cfma(a[c], Wr[m], Un[m * N + c]);
    c128 rhs = cmake(r == n ? 1.0 : 0.0, 0.0);
    int order = row ? -1 : N;  // elimination step at which my row became the pivot row
    int plane[N];              // lane of the group that owns pivot k (uniform within the group)
    c128 pinv[N];              // 1 / pivot k, reused by the back substitution
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const bool cand = order < 0;
      double bv = cand ? cabs1(a[k]) : -1.0;
      int bln = r;
#pragma unroll
      for (int m = 1; m < G; m <<= 1) {
        const double ov = __shfl_xor(bv, m, G);
        const int ol = __shfl_xor(bln, m, G);
        const bool take = ov > bv || (ov == bv && ol < bln);
        bv = take ? ov : bv;
        bln = take ? ol : bln;
      }
      plane[k] = bln;
      if (r == bln) order = k;
      c128 prow[N];
#pragma unroll
      for (int c = k; c < N; ++c)
        prow[c] = cmake(__shfl(a[c].x, bln, G), __shfl(a[c].y, bln, G));
      const c128 prhs = cmake(__shfl(rhs.x, bln, G), __shfl(rhs.y, bln, G));
      const c128 piv = prow[k];
      ok = ok && (piv.x != 0.0 || piv.y != 0.0);
      pinv[k] = crecip_nr(piv);
      if (order < 0) {  // still unused: eliminate column k
        const c128 f = cmul(a[k], pinv[k]);
#pragma unroll
        for (int c = k + 1; c < N; ++c) cfms(a[c], f, prow[c]);
        cfms(rhs, f, prhs);
      }
    }
    c128 w[N];
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
      // the owner of pivot k has folded the unknowns above k into its right-hand side already
      const c128 mine = cmul(rhs, pinv[k]);
      w[k] = cmake(__shfl(mine.x, plane[k], G), __shfl(mine.y, plane[k], G));
      if (order < k) cfms(rhs, a[k], w[k]);
    }
    // Re(w^H U_n w): lane r takes row r, summed in row order on every lane (as quad_form)
    c128 t = cmake(0.0, 0.0);
#pragma unroll
    for (int b2 = 0; b2 < N; ++b2) cfma(t, Un[rr * N + b2], w[b2]);
    double qv = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c)
      if (c == rr) qv = w[c].x * t.x + w[c].y * t.y;
    qv = row ? qv : 0.0;
    double qf = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c) qf += __shfl(qv, c, G);
    qf = qf < 0.0 ? 0.0 : qf;  // np.maximum(., 0): NaN propagates
    const double d = apply_floor(sqrt(qf), floor_kind, eps);
    if (r == n) {
#pragma unroll
      for (int c = 0; c < N; ++c) Wr[c] = cmake(w[c].x / d, -w[c].y / d);
    }
  }
  if (live && row) {
    const long long gb = (long long)b * F + bin;
#pragma unroll
    for (int c = 0; c < N; ++c) W[gb * NN + r * N + c] = Wr[c];
    if (!ok && info && r == 0) atomicAdd(info, 1);
    if (C && qbuf) {
      // y_r = sum_m W[r][m] x_m  =>  E|y_r|^2 = conj(v)^H C conj(v) with v = row r of W
      const c128 *Cm = &Cs[bl][0];
      c128 v[N];
#pragma unroll
      for (int m = 0; m < N; ++m) v[m] = cconj(Wr[m]);
      double qo = 0.0;
#pragma unroll
      for (int a2 = 0; a2 < N; ++a2) {
        c128 t = cmake(0.0, 0.0);
#pragma unroll
        for (int b2 = 0; b2 < N; ++b2) cfma(t, Cm[a2 * N + b2], v[b2]);
        qo += v[a2].x * t.x + v[a2].y * t.y;
      }
      qbuf[gb * N + r] = qo;
    }
  }
}

// =========================================================== power normalisation (pass 5)
// psi_n^2 = (1/F) sum_i q[i][n]; rows of W divided by psi_n, basis rows by psi_n^p
// (ssspy/bss/ilrma.py:412-444).  grid: (bin tiles, B), 256 threads.  Every block folds q over all
// bins in the fixed order of k_norm_scale (ilrma_api.hip: same psi, bit for bit), but asks for its
// 16 demixing matrices and basis rows BEFORE the reduction: one round trip instead of three.
__global__ __launch_bounds__(256) void k_norm_small(c128 *W, double *basis,
                                                    const double *__restrict__ qbuf, int F, int K,
                                                    double p, int floor_kind, double eps) {
  __shared__ double wsum[4][N];
  __shared__ double psi[N];
  const int tile = blockIdx.x, b = blockIdx.y;
  const int i0 = tile * 16;
  const int nb = min(16, F - i0);
  // own rows, requested first
  c128 *Wb = W + ((long long)b * F + i0) * N * N;
  const bool wlive = (int)threadIdx.x < nb * N * N;
  c128 wv = cmake(0.0, 0.0);
  if (wlive) wv = Wb[threadIdx.x];
  constexpr int TPT = (N * 16 * 16 + 255) / 256;  // basis entries per thread at K = 16
  double tv[TPT];
  const int per_n = nb * K;
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    const int idx = threadIdx.x + 256 * u;  // (n, local bin, k) over N * nb * K
    const int n = idx / per_n, rem = idx - n * per_n;
    tv[u] = (n < N) ? basis[(((long long)b * N + n) * F + i0) * K + rem] : 0.0;
  }
  const double *qb = qbuf + (long long)b * F * N;
  {
    const int n = threadIdx.x % N;
    const int stride = (256 / N) * N;
    // (same per-thread order as k_norm_scale, eight loads in flight per round trip)
    double local = 0.0;
    if ((int)threadIdx.x < stride) {
      const int total = F * N;
      for (int e0 = threadIdx.x; e0 < total; e0 += 8 * stride) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = qb[min(e0 + u * stride, total - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) local += e0 + u * stride < total ? v[u] : 0.0;
      }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n2 = 0; n2 < N; ++n2) {
      const double mine = (n == n2) ? local : 0.0;
      const double tot = wave_sum(mine);
      if (lane == 0) wsum[wave][n2] = tot;
    }
    __syncthreads();
    if ((int)threadIdx.x < N) {
      double v = 0.0;
      for (int wv2 = 0; wv2 < 4; ++wv2) v += wsum[wv2][threadIdx.x];
      v = v / (double)F;
      v = v < 0.0 ? 0.0 : v;
      psi[threadIdx.x] = apply_floor(sqrt(v), floor_kind, eps);
    }
  }
  __syncthreads();
  if (wlive) {
    const int n = (threadIdx.x / N) % N;
    Wb[threadIdx.x] = cmake(wv.x / psi[n], wv.y / psi[n]);
  }
#pragma unroll
  for (int u = 0; u < TPT; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int n = idx / per_n, rem = idx - n * per_n;
    if (n < N) {
      const double pp = (p == 2.0) ? psi[n] * psi[n] : pow(psi[n], p);
      basis[(((long long)b * N + n) * F + i0) * K + rem] = tv[u] / pp;
    }
  }
}

}  // namespace ilrma_small_n<N>
using namespace SSSPY_CAT(ilrma_small_n, SSSPY_N);
using fast::make_fast_model;

// Work split of the activation pass: tpc bin tiles per chunk (the last chunk takes the remainder),
// chosen so that the pass has at most the 512 workgroups the chip holds at once (2 per CU: 64 KB of
// LDS, <= 256 registers) -- one more is a second round, i.e. twice the time.
struct SmallPlan {
  int tiles_f, tiles_t;  // 16-bin tiles, 16-frame tiles
  int tpc, nchunks_a;
};
constexpr int SMALL_SLOTS = 512;
static inline SmallPlan make_small_plan(int B, int F, int T) {
  SmallPlan p;
  p.tiles_f = (F + 15) / 16;
  p.tiles_t = (T + 15) / 16;
  for (int t = 1;; ++t) {
    p.tpc = 4 * t;
    p.nchunks_a = p.tiles_f / p.tpc > 0 ? p.tiles_f / p.tpc : 1;
    if ((long long)B * p.tiles_t * p.nchunks_a <= SMALL_SLOTS || p.nchunks_a == 1) break;
  }
  return p;
}

// scratch (bytes) of the activation partials
size_t LAUNCHER(ilrma_small_scratch)(int B, int F, int T, int K) {
  const SmallPlan p = make_small_plan(B, F, T);
  return (size_t)B * p.nchunks_a * N * 2 * K * T * sizeof(double);
}

#define SSSPY_SMALL_MODEL(kernel, HW, ...)                                                        \
  switch (fmodel) {                                                                               \
    case FM_T: hipLaunchKernelGGL((kernel<HW, FM_T>), grid, block, 0, st, __VA_ARGS__); break;     \
    case FM_GGD: hipLaunchKernelGGL((kernel<HW, FM_GGD>), grid, block, 0, st, __VA_ARGS__); break; \
    case FM_GAUSS1:                                                                               \
      hipLaunchKernelGGL((kernel<HW, FM_GAUSS1>), grid, block, 0, st, __VA_ARGS__);               \
      break;                                                                                      \
    case FM_GAUSSP:                                                                               \
      hipLaunchKernelGGL((kernel<HW, FM_GAUSSP>), grid, block, 0, st, __VA_ARGS__);               \
      break;                                                                                      \
    default: hipLaunchKernelGGL((kernel<HW, FM_GAUSS>), grid, block, 0, st, __VA_ARGS__); break;   \
  }

int LAUNCHER(ilrma_small_activation)(const void *X, const void *W, const double *basis, double *act,
                                     int B, int F, int T, int K, int floor_kind, double eps,
                                     double *part, int fmodel, double mparam, int me,
                                     hipStream_t st) {
  const SmallPlan p = make_small_plan(B, F, T);
  const FastModel fm = make_fast_model(fmodel, mparam, me);
  dim3 grid(p.tiles_t, p.nchunks_a, B), block(256);
  if (W) {
    SSSPY_SMALL_MODEL(k_activation_small, true, (const c128 *)X, (const c128 *)W, basis, act, F, T,
                      K, floor_kind, eps, p.tpc, p.nchunks_a, part, fm)
  } else {
    SSSPY_SMALL_MODEL(k_activation_small, false, (const c128 *)X, (const c128 *)W, basis, act, F,
                      T, K, floor_kind, eps, p.tpc, p.nchunks_a, part, fm)
  }
  int rc = check_launch("k_activation_small");
  if (rc || p.nchunks_a == 1) return rc;
  hipLaunchKernelGGL(k_activation_small_finalize,
                     dim3((unsigned)(((long long)K * T + 255) / 256), N, B), block, 0, st, act,
                     (const double *)part, K, T, p.nchunks_a, floor_kind, eps, fm.expo);
  return check_launch("k_activation_small_finalize");
}

// IP1 (+ output power q[bin][n] = Re(w_n C w_n^H) when C and qbuf are given) from the finished
// covariance (nchunks == 0: `Usrc` is U) or from the partial records k_wcov_fast left for its fold
// (nchunks of them per group of `rbins` bins).
// rec_stride: c128 between records, 0 = packed (rbins N^3).
int LAUNCHER(ilrma_small_ip1)(const void *Usrc, int nchunks, int rbins, long long rec_stride,
                              const void *C, void *W, int B, int F, int floor_kind, double eps,
                              double *qbuf, int *info, hipStream_t st) {
  const long long rec = rec_stride > 0 ? rec_stride : (long long)rbins * (N * N * N);
  hipLaunchKernelGGL(k_ip1_small, dim3((F + 15) / 16, B), dim3(256), 0, st, (c128 *)W,
                     (const c128 *)Usrc, (const c128 *)C, C ? qbuf : (double *)nullptr, F, nchunks,
                     rbins, rec, floor_kind, eps, info, (double *)nullptr, 0ll);
  return check_launch("k_ip1_small");
}

// The same, leaving the log-determinants of the incoming filters as one share per tile of 16 bins:
// logdet[tile * logdet_stride + b], tile < ceil(F / 16) (see k_ip1_small).
int LAUNCHER(ilrma_small_ip1_logdet)(const void *Usrc, int nchunks, int rbins, long long rec_stride,
                                     const void *C, void *W, int B, int F, int floor_kind,
                                     double eps, double *qbuf, int *info, double *logdet,
                                     long long logdet_stride, hipStream_t st) {
  const long long rec = rec_stride > 0 ? rec_stride : (long long)rbins * (N * N * N);
  hipLaunchKernelGGL(k_ip1_small, dim3((F + 15) / 16, B), dim3(256), 0, st, (c128 *)W,
                     (const c128 *)Usrc, (const c128 *)C, C ? qbuf : (double *)nullptr, F, nchunks,
                     rbins, rec, floor_kind, eps, info, logdet, logdet_stride);
  return check_launch("k_ip1_small");
}

// power normalisation from q (ssspy/bss/ilrma.py:412-444)
int LAUNCHER(ilrma_small_norm)(void *W, double *basis, const double *qbuf, int B, int F, int K,
                               double domain, int floor_kind, double eps, hipStream_t st) {
  hipLaunchKernelGGL(k_norm_small, dim3((F + 15) / 16, B), dim3(256), 0, st, (c128 *)W, basis, qbuf,
                     F, K, domain, floor_kind, eps);
  return check_launch("k_norm_small");
}

#undef SSSPY_SMALL_MODEL

#ifdef SMALL_TRACE
extern "C" int ssspy_debug_small_trace(long long *host_out, int count) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_small_trace),
                                  (size_t)count * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif

}  // namespace ssspy
