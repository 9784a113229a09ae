// FastGaussMNMF, general shapes: any n_channels M in [2, 8] and n_sources N in [1, 8].
//
// The MFMA-tile kernels of mnmf_kernels.hip keep the diagonaliser rows, the spatial rows and the
// GEMM1 output of every source in registers and are compiled for N, M <= 4.  Beyond that the unit of
// work here is "one lane = one (bin, frame) point" (the GaussMNMF design, gmnmf_kernels.hip): the
// lane forms lambda_n, |q_m^H x|^2 and R~_m = sum_n lambda_n d_nm and leaves
//   traces   A_n = sum_m d_nm |q_m^H x|^2 / R~_m^2,  Bt_n = sum_m d_nm / R~_m      (basis, activation)
//   weights  1 / R~_m                                                           (diagonaliser: IP1)
//   loss     sum_m |q_m^H x|^2 / R~_m + log R~_m
// in HBM; small kernels contract them with the activation (over frames) or the basis (over bins), and
// the spatial update folds its N x M sums per bin through LDS.  Per iteration that is five passes over
// X instead of four plus the (N, F, T) traces.  Round 6: the passes walk runs of bins with the
// activation column of a frame held in registers (k_walk) instead of re-reading it per bin.
//
// replaces: ssspy/bss/mnmf.py:1278-1303 (update_once), :1305-1417, :1449-1514, :1635-1675, :632-678,
//           :1219-1261 (loss), :1174-1217 (Wiener filter) for shapes outside N, M <= 4.
#include "common.hpp"
#include "hermitian.hpp"
#include "smallmat.hpp"
#include "ssspy_amd.h"

namespace ssspy {

int ip1_with_power(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N,
                   int floor_kind, double floor_eps, int *info, hipStream_t st);
int row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st);

namespace fmg {

constexpr int NMAX = SSSPY_MAX_SOURCES;
enum { MODE_TRACES = 0, MODE_WEIGHTS = 1, MODE_LOSS = 2 };

// ---- Round 6.  Rounds 3-5 ran one workgroup per (bin, frame block): every point recomputed
// lambda_n = sum_k t_nik v_nkj from activation rows that all F bins of a mixture re-read -- 64 loads
// per point, 4.3 GB through the L2 per pass at 8 mixtures of 8 channels (24 TB/s: the passes ran at
// the L2's rate, 1.5 TB/s of HBM traffic) -- and the two contractions read the traces once per
// basis index.  Now a workgroup keeps the activation tile of ITS 64 frames in LDS (n_basis <= 8:
// 32 KB) and each of its waves walks a run of bins with it, the bin's Q, D and basis rows staged in
// a wave-private LDS patch; wider bases take the column from memory per bin as before.  (The column
// in 128 registers per lane was tried first: the Wiener-filter mode then spilled 500 of them.)
constexpr int KT = 8;   // activation rows per source a lane holds
constexpr int WB = 4;   // waves per workgroup, each walking its own run of bins

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Q, D and the basis rows of one bin, private to a wave
template <int M>
struct WaveBin {
  c128 q[M * M];
  double d[NMAX * M];   // d[n * M + m], zero beyond N
  double t[NMAX * KT];  // t[n * KT + k], zero beyond (N, K); unused when K > KT
};

// Q, D and the basis rows of bin i in flight to a lane's registers (one element of each per lane:
// M M, NMAX M and NMAX KT are all <= 64) together with the lane's point of X, and their parking in
// the wave's patch.  The walk fetches bin i + 1 before it works on bin i: with about two waves per
// SIMD the bins of a wave were a chain of load latencies (Q -> LDS -> X -> arithmetic -> stores),
// 78 us for a pass whose arithmetic is 27.
template <int M>
struct BinFetch {
  c128 x[M];
  c128 q;
  double d, t;
};

template <int M>
__device__ __forceinline__ void fetch_bin(BinFetch<M> &f, const c128 *__restrict__ Xb,
                                          const c128 *__restrict__ Q, const double *__restrict__ Dsp,
                                          const double *__restrict__ basis, int b, int N, int F, int T,
                                          int K, int i, int j, int lane) {
  static_assert(M * M <= 64 && NMAX * M <= 64 && NMAX * KT <= 64, "one element per lane");
  const long long bin = (long long)b * F + i;
  f.q = lane < M * M ? Q[bin * (M * M) + lane] : cmake(0.0, 0.0);
  f.d = lane < N * M ? Dsp[bin * (N * M) + lane] : 0.0;
  const int n = lane / KT, k = lane % KT;
  f.t = (K <= KT && n < N && k < K) ? basis[(((long long)b * N + n) * F + i) * K + k] : 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m) f.x[m] = Xb[((long long)m * F + i) * T + j];
}

template <int M>
__device__ __forceinline__ void park_bin(WaveBin<M> &s, const BinFetch<M> &f, int lane) {
  wave_lds_sync();  // (the previous bin's reads are done)
  if (lane < M * M) s.q[lane] = f.q;
  if (lane < NMAX * M) s.d[lane] = f.d;
  s.t[lane] = f.t;
  wave_lds_sync();
}

// the workgroup's activation tile (n_basis <= KT): vt[n * KT + k][lane] = act[b, n, k, j0 + lane],
// shared by its four waves (they walk different bins of the same 64 frames); zero beyond (N, K)
__device__ __forceinline__ void load_tile(double (*vt)[64], const double *__restrict__ act_b, int N,
                                          int T, int K, int j0) {
  for (int e = threadIdx.x >> 6; e < NMAX * KT; e += WB) {
    const int n = e / KT, k = e % KT, lane = threadIdx.x & 63;
    const int j = min(j0 + lane, T - 1);
    vt[e][lane] = (n < N && k < K) ? act_b[((long long)n * K + k) * T + j] : 0.0;
  }
  __syncthreads();
}

// lam[n] = sum_k t_nik v_nkj of bin i at the lane's frame
template <int M>
__device__ __forceinline__ void lambda_terms(double (&lam)[NMAX], const double (*vt)[64],
                                             const WaveBin<M> &s, const double *__restrict__ act_b,
                                             const double *__restrict__ basis, int b, int N, int F,
                                             int T, int K, int i, int j) {
  if (K <= KT) {
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      double l = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) l = fma(s.t[n * KT + k], vt[n * KT + k][threadIdx.x & 63], l);
      lam[n] = l;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      double l = 0.0;
      if (n < N) {
        const double *tr = basis + (((long long)b * N + n) * F + i) * K;
        for (int k = 0; k < K; ++k) l = fma(tr[k], act_b[((long long)n * K + k) * T + j], l);
      }
      lam[n] = l;
    }
  }
  // (scheduling fences between the phases of a point: left alone, hipcc hoists the LDS reads of all
  //  of them to the top and spills the activation column)
  __builtin_amdgcn_sched_barrier(0);
}

// per-point terms of bin i: lam[n], qx2[m] = |(Q x)_m|^2, rc[m] = R~_m
template <int M>
__device__ __forceinline__ void point_terms(double (&lam)[NMAX], double (&qx2)[M], double (&rc)[M],
                                            const double (*vt)[64], const WaveBin<M> &s,
                                            const c128 (&x)[M],
                                            const double *__restrict__ act_b,
                                            const double *__restrict__ basis, int b, int N, int F,
                                            int T, int K, int i, int j) {
  lambda_terms<M>(lam, vt, s, act_b, basis, b, N, F, T, K, i, j);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    c128 acc = cmake(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < M; ++a) cfma(acc, s.q[m * M + a], x[a]);
    qx2[m] = cabs2(acc);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    double r = 0.0;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) r = fma(lam[n], s.d[n * M + m], r);
    rc[m] = r;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// bins per wave so that the launch has about 512 workgroups (two rounds of the chip's 256 CUs at
// one workgroup each, or one at two), at most 16 (the reuse of the activation column)
struct WalkPlan {
  int gx, gy, bpw;  // frame tiles, bin groups, bins per wave
};
static inline WalkPlan walk_plan(int B, int F, int T) {
  WalkPlan p;
  p.gx = (T + 63) / 64;
  const int most = (F + WB - 1) / WB;  // groups at one bin per wave
  int gy = (512 + p.gx * B - 1) / (p.gx * B);
  gy = gy < 1 ? 1 : (gy > most ? most : gy);
  p.bpw = (F + WB * gy - 1) / (WB * gy);
  if (p.bpw > 16) p.bpw = 16;
  p.gy = (F + WB * p.bpw - 1) / (WB * p.bpw);
  return p;
}

enum { MODE_SEPARATE = 3, MODE_SPATIAL = 4 };
// a point's row in the wave's LDS patch (MODE_SPATIAL): lam[0..8), h[8..16), g[16..24) -- the operands
// of a 16 x 16 x 4 matrix product over the frames (rows 8..15 of the left operand are zero) -- padded
// against bank conflicts; the patch holds 32 points, the wave's 64 go through it in two halves (with
// all 64 the kernel held 110 KB of LDS: one workgroup per CU)
constexpr int PROW = 26;

// grid: (frame tiles of 64, bin groups, B), 256 threads: wave w walks bins
// [(blockIdx.y * WB + w) * bpw, + bpw) for the block's 64 frames (lane = frame).
//   MODE_TRACES   out0, out1 (B,N,F,T): A_n, Bt_n
//   MODE_WEIGHTS  out0 (B,M,F,T): 1 / R~_m
//   MODE_LOSS     out0: one slot per (workgroup, wave), [slot][B]
//   MODE_SPATIAL  out0: per frame tile the (num, den) sums of the spatial update,
//                 [tile][b][i][n * M + m][2]
//   MODE_SEPARATE Yout (B,N,F,T): the Wiener filter's closed form R^-1 = Q^H diag(1 / rc) Q
//                 (ref: ssspy/bss/mnmf.py:1174-1217) where the eigenvalue floor is provably idle
//                 (lambda_min(R) >= min rc / ||Q||_F^2 > eps); bins with a point where it may act
//                 are flagged in `redo` for the repair launch (k_separate<M, true>)
template <int M, int MODE>
__global__ __launch_bounds__(64 * WB, 2) void k_walk(const c128 *__restrict__ X,
                                                  const c128 *__restrict__ Q,
                                                  const c128 *__restrict__ Qinv,
                                                  const double *__restrict__ Dsp,
                                                  const double *__restrict__ basis,
                                                  const double *__restrict__ act,
                                                  double *__restrict__ out0,
                                                  double *__restrict__ out1, c128 *__restrict__ Yout,
                                                  int N, int F, int T, int K, int bpw, int ref,
                                                  int floor_kind, double eps, int *redo) {
  __shared__ WaveBin<M> bins[WB];
  __shared__ double pts[MODE == MODE_SPATIAL ? WB * 32 * PROW : 1];
  __shared__ c128 qref[MODE == MODE_SEPARATE ? WB * M : 1];
  __shared__ c128 smem_s[MODE == MODE_SEPARATE ? WB * M * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z;
  const int j_raw = blockIdx.x * 64 + lane;
  const bool valid = j_raw < T;
  const int j = valid ? j_raw : T - 1;
  const c128 *Xb = X + (long long)b * M * F * T;
  const double *act_b = act + (long long)b * N * K * T;
  WaveBin<M> &s = bins[wave];
  __shared__ double vt[NMAX * KT][64];
  if (K <= KT) load_tile(vt, act_b, N, T, K, blockIdx.x * 64);
  const int i0 = (blockIdx.y * WB + wave) * bpw;
  double loss = 0.0;
  const int i1 = min(i0 + bpw, F);
  BinFetch<M> nxt;
  if (i0 < i1) fetch_bin<M>(nxt, Xb, Q, Dsp, basis, b, N, F, T, K, i0, j, lane);
#pragma unroll 1
  for (int i = i0; i < i1; ++i) {
    park_bin<M>(s, nxt, lane);
    c128 x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = nxt.x[m];
    fetch_bin<M>(nxt, Xb, Q, Dsp, basis, b, N, F, T, K, min(i + 1, i1 - 1), j, lane);
    __builtin_amdgcn_sched_barrier(0);
    double lam[NMAX], qx2[M], rc[M];
    if (MODE != MODE_SEPARATE) point_terms<M>(lam, qx2, rc, vt, s, x, act_b, basis, b, N, F, T, K, i, j);
    if (MODE == MODE_TRACES) {
      double g[M], h[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        g[m] = 1.0 / rc[m];
        h[m] = qx2[m] * g[m] * g[m];
      }
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < N && valid) {
          double sa = 0.0, sb = 0.0;
#pragma unroll
          for (int m = 0; m < M; ++m) {
            sa = fma(s.d[n * M + m], h[m], sa);
            sb = fma(s.d[n * M + m], g[m], sb);
          }
          const long long o = (((long long)b * N + n) * F + i) * T + j;
          out0[o] = sa;
          out1[o] = sb;
        }
      }
    } else if (MODE == MODE_WEIGHTS) {
      if (valid) {
#pragma unroll
        for (int m = 0; m < M; ++m) out0[(((long long)b * M + m) * F + i) * T + j] = 1.0 / rc[m];
      }
    } else if (MODE == MODE_LOSS) {
      double term = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) term += qx2[m] / rc[m] + log(rc[m]);
      loss += valid ? term : 0.0;
    } else if (MODE == MODE_SPATIAL) {
      // d_inm <- d_inm sqrt(sum_j lam_n h_m / sum_j lam_n g_m): the sums over the wave's 64 frames
      // are a (16 x 64) x (64 x 16) product -- rows lam_0..7 (and eight zero rows), columns h_0..7,
      // g_0..7 -- on the matrix core: the points go through the wave's LDS patch in operand layout,
      // 32 at a time, 16 v_mfma_f64_16x16x4.  (Rounds 3-5: lane (n, m) walked the 64 points, 192 LDS
      // reads.)
      double *pw = pts + wave * 32 * PROW;
      double *mine = pw + (lane & 31) * PROW;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
      const int kq = lane >> 4, ic = lane & 15;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        wave_lds_sync();
        if ((lane >> 5) == half) {
#pragma unroll
          for (int n = 0; n < NMAX; ++n) mine[n] = valid ? lam[n] : 0.0;  // frames beyond T add nothing
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const double g = m < M ? 1.0 / rc[m < M ? m : 0] : 0.0;
            mine[8 + m] = m < M ? qx2[m < M ? m : 0] * g * g : 0.0;
            mine[16 + m] = g;
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const double *row = pw + (4 * t + kq) * PROW;
          const double lamv = row[ic & 7];
          acc = mfma_f64(ic < 8 ? lamv : 0.0, row[8 + ic], acc);
        }
      }
      // D: column ic, rows kq + 4 reg -> n = kq (reg 0), kq + 4 (reg 1); rows 8..15 are the zero rows
      const int m = ic & 7;
      if (m < M) {
        double *dst = out0 + (((long long)blockIdx.x * gridDim.z + b) * F + i) * (N * M) * 2 + (ic >> 3);
        if (kq < N) dst[(kq * M + m) * 2] = acc[0];
        if (kq + 4 < N) dst[((kq + 4) * M + m) * 2] = acc[1];
      }
    } else {  // MODE_SEPARATE
      // s_m = q~[ref][m] (Q x)_m / rc_m one channel at a time in a rolled loop, parked in the wave's
      // LDS patch; then Y_n = lam_n sum_m d_nm s_m.  (Unrolled, hipcc issues the 64 16-byte reads of
      // Q ahead of the x loads they wait for: 256 registers, 500 spilled at 8 channels.)
      const long long bin = (long long)b * F + i;
      double qf2 = lane < M * M ? cabs2(s.q[lane]) : 0.0;
      qf2 = wave_sum(qf2);
      wave_lds_sync();
      if (lane < M) qref[wave * M + lane] = Qinv[bin * (M * M) + ref * M + lane];
      wave_lds_sync();
      lambda_terms<M>(lam, vt, s, act_b, basis, b, N, F, T, K, i, j);
      c128 *smw = smem_s + (wave * M) * 64 + lane;  // [m][lane]
      double rcmin = 0.0;
#pragma unroll 1
      for (int m = 0; m < M; ++m) {
        double r = 0.0;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) r = fma(lam[n], s.d[n * M + m], r);
        rcmin = m == 0 ? r : (r < rcmin ? r : rcmin);
        c128 acc = cmake(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < M; ++a) cfma(acc, s.q[m * M + a], x[a]);
        const double g = 1.0 / r;
        smw[m * 64] = cmul(qref[wave * M + m], cmake(acc.x * g, acc.y * g));
      }
      const bool closed = floor_kind != SSSPY_FLOOR_ADD && rcmin > eps * qf2 * 1.0000001;
      if (valid && !closed) redo[bin] = 1;  // (every writer stores the same value)
      if (valid && closed) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
          if (n < N) {
            c128 o = cmake(0.0, 0.0);
#pragma unroll
            for (int m = 0; m < M; ++m) {
              const double dv = s.d[n * M + m];
              const c128 sv = smw[m * 64];
              o.x = fma(dv, sv.x, o.x);
              o.y = fma(dv, sv.y, o.y);
            }
            Yout[(((long long)b * N + n) * F + i) * T + j] = cmake(lam[n] * o.x, lam[n] * o.y);
          }
      }
    }
  }
  if (MODE == MODE_LOSS) {
    loss = wave_sum(loss);
    // one slot per (frame tile, bin group, wave) of the mixture, [slot][B]
    if (lane == 0)
      out0[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * WB + wave) * gridDim.z + b] =
          loss / (double)T;
  }
}

// d_inm <- d_inm sqrt(sum_tiles num / sum_tiles den) (no floor; ssspy/bss/mnmf.py:1650-1675)
__global__ __launch_bounds__(256) void k_spatial_fold(double *Dsp, const double *__restrict__ part,
                                                      long long count, int tiles) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const double2 sums = ordered_sum(reinterpret_cast<const double2 *>(part) + e, count, tiles);
  Dsp[e] = sqrt(sums.x / sums.y) * Dsp[e];
}

// basis[b,n,i,k] <- floor(basis * sqrt(sum_j V A / sum_j V Bt)).  A workgroup takes a run of `nb` bins
// of ONE source and a tile of 8 basis indices, parks the tile's activation rows in LDS (up to
// BASIS_TMAX frames; beyond, they are read from memory per bin) and each of its four waves walks
// every fourth bin of the run: the traces are read once, coalesced (lane = frame), the loads of the
// next 256 frames in flight while the current ones are contracted; the activation rows are read
// once per workgroup instead of once per bin (through the L2 that was 5x the traces' bytes).  The
// 16 sums of a (bin, tile) meet by a halving exchange -- at offset 32 a lane keeps eight of its
// values and trades the other eight, at 16 four, ... -- 19 shuffles instead of the 96 of sixteen
// wave-wide sums.  grid: (ceil(F / nb), ceil(K / 8), B N), 256 threads
constexpr int BASIS_TMAX = 1024;
// one step of the halving exchange: the lanes with bit OFF set keep val[C..2C), the others val[0..C),
// each adds what its partner at distance OFF held of the same values.  (A template per step: written
// as one loop over (C, OFF), hipcc left the loop rolled and indexed val[] through 884 v_cndmask --
// 85 of the kernel's 125 us.)
template <int C, int OFF>
__device__ __forceinline__ void halve(double (&val)[2 * KT], int lane) {
  const bool upper = (lane & OFF) != 0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
    const double send = upper ? val[q] : val[q + C];
    const double keep = upper ? val[q + C] : val[q];
    val[q] = keep + __shfl_xor(send, OFF, 64);
  }
}
__global__ __launch_bounds__(256) void k_basis(double *basis, const double *__restrict__ act,
                                               const double *__restrict__ A,
                                               const double *__restrict__ Bt, int N, int F, int T,
                                               int K, int nb, int floor_kind, double eps) {
  extern __shared__ __attribute__((aligned(16))) double vtile[];  // [KT][T] when T <= BASIS_TMAX
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long bn = blockIdx.z;
  const int k0 = blockIdx.y * KT;
  const bool tiled = T <= BASIS_TMAX;
  if (tiled) {
    for (int k = 0; k < KT; ++k)
      for (int j = threadIdx.x; j < T; j += blockDim.x)
        vtile[k * T + j] = k0 + k < K ? act[(bn * K + k0 + k) * T + j] : 0.0;
    __syncthreads();
  }
  const int i_begin = blockIdx.x * nb + wave, i_end = min((int)(blockIdx.x + 1) * nb, F);
  if (i_begin >= i_end) return;
  const int chunks = (T + 255) / 256;  // of four runs of 64 frames
  const double *abase = A + bn * F * T, *btbase = Bt + bn * F * T;
  double av[4], bv[4];
  auto fetch = [&](int i, int c, double(&a4)[4], double(&b4)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = c * 256 + 64 * u + lane;
      const bool in = j < T && i < i_end;
      a4[u] = in ? abase[(long long)i * T + j] : 0.0;
      b4[u] = in ? btbase[(long long)i * T + j] : 0.0;
    }
  };
  fetch(i_begin, 0, av, bv);
  for (int i = i_begin; i < i_end; i += 4) {
    double val[2 * KT];  // sn[0..8), sd[0..8)
#pragma unroll
    for (int k = 0; k < 2 * KT; ++k) val[k] = 0.0;
    for (int c = 0; c < chunks; ++c) {
      double na[4], nbv[4];
      const bool last = c + 1 == chunks;
      fetch(last ? i + 4 : i, last ? 0 : c + 1, na, nbv);  // (zeros past the run's end)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(c * 256 + 64 * u + lane, T - 1);  // (beyond T the traces above are zero)
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const double vv = tiled ? vtile[k * T + j]
                                  : (k0 + k < K ? act[(bn * K + k0 + k) * T + j] : 0.0);
          val[k] = fma(vv, av[u], val[k]);
          val[KT + k] = fma(vv, bv[u], val[KT + k]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = na[u];
        bv[u] = nbv[u];
      }
    }
    halve<8, 32>(val, lane);
    halve<4, 16>(val, lane);
    halve<2, 8>(val, lane);
    halve<1, 4>(val, lane);
    double tot = val[0];  // of value (lane >> 2) & 15, summed over the lanes that differ in bits 5..2
    tot += __shfl_xor(tot, 2, 64);
    tot += __shfl_xor(tot, 1, 64);
    const double den = __shfl(tot, (lane + 32) & 63, 64);  // sd[k] sits 32 lanes above sn[k]
    const int k = lane >> 2;
    if ((lane & 3) == 0 && k < KT && k0 + k < K) {
      const long long o = (bn * F + i) * K + k0 + k;
      basis[o] = apply_floor(basis[o] * sqrt(tot / den), floor_kind, eps);
    }
  }
}

// act[b,n,k,j] <- floor(act * sqrt(sum_i T A / sum_i T Bt)): lanes along 64 frames, the 8 waves of a
// workgroup take every eighth bin, their sums meet in LDS in wave order -- deterministic, no atomics,
// the traces read once per tile of 8 basis indices.  grid: (ceil(T/64), ceil(K/8), N B), 512 threads
constexpr int AW = 8;
static_assert(AW == KT, "k_activation finishes basis index `wave` of the tile");
__global__ __launch_bounds__(64 * AW) void k_activation(const double *__restrict__ basis, double *act,
                                                        const double *__restrict__ A,
                                                        const double *__restrict__ Bt, int N, int F,
                                                        int T, int K, int floor_kind, double eps) {
  __shared__ double part[AW][2 * KT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j_raw = blockIdx.x * 64 + lane;
  const int j = j_raw < T ? j_raw : T - 1;
  const int k0 = blockIdx.y * KT;
  const long long bn = blockIdx.z;
  const double *tb = basis + bn * F * K + k0;
  const long long base = bn * F * T + j;
  double sn[KT], sd[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) sn[k] = sd[k] = 0.0;
  for (int i = wave; i < F; i += AW) {
    const double av = A[base + (long long)i * T], bv = Bt[base + (long long)i * T];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const double t = k0 + k < K ? tb[(long long)i * K + k] : 0.0;
      sn[k] = fma(t, av, sn[k]);
      sd[k] = fma(t, bv, sd[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    part[wave][k][lane] = sn[k];
    part[wave][KT + k][lane] = sd[k];
  }
  __syncthreads();
  // thread (k = wave, frame = lane) finishes one output
  if (k0 + wave < K && j_raw < T) {
    double tn = 0.0, td = 0.0;
#pragma unroll
    for (int w = 0; w < AW; ++w) {
      tn += part[w][wave][lane];
      td += part[w][KT + wave][lane];
    }
    double *dst = act + (bn * K + k0 + wave) * T + j;
    *dst = apply_floor((*dst) * sqrt(tn / td), floor_kind, eps);
  }
}

// psi_m = floor(sqrt(mean_i q[i][m])); Q[:,m,:] /= psi_m; D[:,:,m] /= psi_m^2.  grid (ceil(F/64), B)
__global__ __launch_bounds__(256) void k_norm_scale(c128 *Q, double *Dsp,
                                                    const double *__restrict__ qbuf, int N, int M,
                                                    int F, int floor_kind, double eps) {
  __shared__ double part[256];
  __shared__ double psi[NMAX];
  const int b = blockIdx.y;
  const double *qb = qbuf + (long long)b * F * M;
  // one pass over the (F, M) powers: thread t = r M + m adds rows r, r + 256 / M, ... of channel m,
  // thread m then adds the partial sums of its channel in order (rounds 3-5: M block-wide sums in
  // turn, each over a stride-M walk -- 21 us of latency for 33 KB)
  const int rows = 256 / M, r = threadIdx.x / M, mch = threadIdx.x % M;
  double local = 0.0;
  if (r < rows)
    for (int i = r; i < F; i += rows) local += qb[(long long)i * M + mch];
  part[threadIdx.x] = local;
  __syncthreads();
  if (threadIdx.x < M) {
    double total = 0.0;
    for (int q = 0; q < rows; ++q) total += part[q * M + threadIdx.x];
    double v = total / (double)F;
    v = v < 0.0 ? 0.0 : v;
    psi[threadIdx.x] = apply_floor(sqrt(v), floor_kind, eps);
  }
  __syncthreads();
  const int i0 = blockIdx.x * 64;
  const int nb = min(64, F - i0);
  c128 *Qb = Q + ((long long)b * F + i0) * M * M;
  for (int e = threadIdx.x; e < nb * M * M; e += blockDim.x) {
    const int m = (e / M) % M;
    const c128 v = Qb[e];
    Qb[e] = cmake(v.x / psi[m], v.y / psi[m]);
  }
  double *Db = Dsp + ((long long)b * F + i0) * N * M;
  for (int e = threadIdx.x; e < nb * N * M; e += blockDim.x) {
    const int m = e % M;
    Db[e] = Db[e] / (psi[m] * psi[m]);
  }
}

template <int M>
__global__ __launch_bounds__(64) void k_qinv(const c128 *__restrict__ Q, c128 *Qinv,
                                             long long nbins, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<M> A, Inv;
  load_mat<M>(A, Q + idx * (M * M));
  const bool ok = invert<M>(A, Inv);
  store_mat<M>(Inv, Qinv + idx * (M * M));
  if (!ok && info) atomicAdd(info, 1);
}

// Multichannel Wiener filter (ref: ssspy/bss/mnmf.py:1174-1217): grid (F, B), lanes along frames.
// R = Q~ diag(rc) Q~^H, Q~ = Q^-1; closed form R^-1 = Q^H diag(1/rc) Q when the eigenvalue floor is
// provably idle (lambda_min(R) >= min rc / ||Q||_F^2 > eps), else the Jacobi eigen-floor.
// Round 5: two launches of it.  REPAIR == false holds the closed form alone (no M x M working set: 5-8
// channels no longer spill 530-2400 registers per lane on the path every call takes) and flags the
// bins that have a point where the floor may act; REPAIR == true is the old kernel restricted to the
// flagged bins (both forms per point, as before).
// EIG (an arbitrary flooring callable, evaluated on the host -- round 5): 1 = every point takes the
// literal route up to its eigen-decomposition and leaves the eigenvalues (ascending, as numpy.linalg.
// eigh hands them to the callable: lam_io (B, F, T, M)) and eigenvectors (P_io (B, F, T, M, M)) in HBM;
// 2 = the same walk from the floored eigenvalues and the stored eigenvectors to Y.
template <int M, bool REPAIR, int EIG = 0>
__global__ __launch_bounds__(128) void k_separate(const c128 *__restrict__ X,
                                                  const c128 *__restrict__ Q,
                                                  const c128 *__restrict__ Qinv,
                                                  const double *__restrict__ Dsp,
                                                  const double *__restrict__ basis,
                                                  const double *__restrict__ act, c128 *Y, int N,
                                                  int F, int T, int K, int ref, int floor_kind,
                                                  double eps, int *redo, double *lam_io = nullptr,
                                                  c128 *P_io = nullptr) {
  const int i = blockIdx.x, b = blockIdx.y;
  if (REPAIR && EIG == 0 && !redo[(long long)b * F + i]) return;
  __shared__ c128 qt[M * M];
  __shared__ c128 qsrc[M * M];
  __shared__ double dd[NMAX * M];
  if (threadIdx.x < M * M) {
    qt[threadIdx.x] = Qinv[((long long)b * F + i) * (M * M) + threadIdx.x];
    qsrc[threadIdx.x] = Q[((long long)b * F + i) * (M * M) + threadIdx.x];
  }
  for (int e = threadIdx.x; e < N * M; e += blockDim.x)
    dd[e] = Dsp[((long long)b * F + i) * (N * M) + e];
  __syncthreads();
  double qf2 = 0.0;
#pragma unroll
  for (int e = 0; e < M * M; ++e) qf2 += cabs2(qsrc[e]);
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    double lam[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      double r = 0.0;
      if (n < N) {
        const double *tr = basis + (((long long)b * N + n) * F + i) * K;
        const double *Vn = act + ((long long)b * N + n) * K * T;
        for (int k = 0; k < K; ++k) r = fma(tr[k], Vn[(long long)k * T + j], r);
      }
      lam[n] = r;
    }
    double rc[M];
    double rcmin = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double r = 0.0;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) r = fma(lam[n], dd[n * M + m], r);
      rc[m] = r;
      rcmin = m == 0 ? r : (r < rcmin ? r : rcmin);
    }
    c128 x[M], z[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + i) * T + j];
    c128 sm[M];  // s_m = (Q~^H R^-1 x)_m, then scaled by q~[ref][m]
    const bool closed = EIG == 0 && floor_kind != SSSPY_FLOOR_ADD && rcmin > eps * qf2 * 1.0000001;
    if (!REPAIR && !closed) {
      redo[(long long)b * F + i] = 1;  // (every writer stores the same value)
      continue;
    }
    if (closed) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < M; ++a) cfma(y, qsrc[m * M + a], x[a]);
        const double g = 1.0 / rc[m];
        sm[m] = cmake(y.x * g, y.y * g);
      }
    } else if constexpr (REPAIR) {
      // (A and P live in the lane's scratch memory: jacobi_eigh_rolled indexes them at run time.
      //  Unrolled on registers this branch spilled 530-2 400 VGPRs at 5-8 channels.)
      c128 A[M][M], P[M][M];
#pragma unroll 1
      for (int a = 0; a < M; ++a)
#pragma unroll 1
        for (int c2 = a; c2 < M; ++c2) {
          c128 s = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const c128 zz = cmulc(qt[a * M + m], qt[c2 * M + m]);
            s.x = fma(rc[m], zz.x, s.x);
            s.y = fma(rc[m], zz.y, s.y);
          }
          if (a == c2) s.y = 0.0;
          A[a][c2] = s;
          A[c2][a] = cconj(s);
        }
      double evs[M];
      const long long point = ((long long)b * F + i) * T + j;
      if constexpr (EIG == 2) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
          evs[k] = lam_io[point * M + k];
#pragma unroll
          for (int a = 0; a < M; ++a) P[a][k] = P_io[(point * M + a) * M + k];
        }
      } else {
        jacobi_eigh_rolled<M>(A, P);
#pragma unroll
        for (int k = 0; k < M; ++k) evs[k] = apply_floor(A[k][k].x, floor_kind, eps);
      }
      if constexpr (EIG == 1) {
        // ascending order (rank of every eigenvalue, ties by index)
#pragma unroll 1
        for (int k = 0; k < M; ++k) {
          int rank = 0;
#pragma unroll 1
          for (int l = 0; l < M; ++l)
            rank += (A[l][l].x < A[k][k].x || (A[l][l].x == A[k][k].x && l < k)) ? 1 : 0;
          lam_io[point * M + rank] = A[k][k].x;
#pragma unroll 1
          for (int a = 0; a < M; ++a) P_io[(point * M + a) * M + rank] = P[a][k];
        }
        continue;
      }
#pragma unroll
      for (int a = 0; a < M; ++a) z[a] = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) {
        c128 proj = cmake(0.0, 0.0);  // p_k^H x
#pragma unroll
        for (int a = 0; a < M; ++a) {
          const c128 pk = P[a][k];
          proj.x += pk.x * x[a].x + pk.y * x[a].y;
          proj.y += pk.x * x[a].y - pk.y * x[a].x;
        }
        const double ev = evs[k];
        proj = cmake(proj.x / ev, proj.y / ev);
#pragma unroll
        for (int a = 0; a < M; ++a) cfma(z[a], P[a][k], proj);
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        c128 s = cmake(0.0, 0.0);
#pragma unroll
        for (int c2 = 0; c2 < M; ++c2) {
          const c128 qv = qt[c2 * M + m];
          s.x += qv.x * z[c2].x + qv.y * z[c2].y;
          s.y += qv.x * z[c2].y - qv.y * z[c2].x;
        }
        sm[m] = s;
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) sm[m] = cmul(qt[ref * M + m], sm[m]);
    for (int n = 0; n < N; ++n) {
      c128 y = cmake(0.0, 0.0);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const double wgt = lam[n] * dd[n * M + m];
        y.x = fma(wgt, sm[m].x, y.x);
        y.y = fma(wgt, sm[m].y, y.y);
      }
      Y[(((long long)b * N + n) * F + i) * T + j] = y;
    }
  }
}

#define FMG_DISPATCH_M(M_, CALL)                                                           \
  switch (M_) {                                                                            \
    case 2: { constexpr int MM = 2; CALL; } break;                                         \
    case 3: { constexpr int MM = 3; CALL; } break;                                         \
    case 4: { constexpr int MM = 4; CALL; } break;                                         \
    case 5: { constexpr int MM = 5; CALL; } break;                                         \
    case 6: { constexpr int MM = 6; CALL; } break;                                         \
    case 7: { constexpr int MM = 7; CALL; } break;                                         \
    case 8: { constexpr int MM = 8; CALL; } break;                                         \
    default: return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_channels must be in [2, 8]"); \
  }

template <int MODE>
static int launch_walk(const void *X, const void *Q, const void *Qinv, const double *D,
                       const double *basis, const double *act, double *out0, double *out1, void *Y,
                       int B, int N, int M, int F, int T, int K, int ref, int floor_kind, double eps,
                       int *redo, hipStream_t st) {
  // point_terms keeps lam[NMAX]: more sources would silently drop out of R~
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  const WalkPlan p = walk_plan(B, F, T);
  dim3 grid(p.gx, p.gy, B), block(64 * WB);
  FMG_DISPATCH_M(M, hipLaunchKernelGGL((k_walk<MM, MODE>), grid, block, 0, st, (const c128 *)X,
                                       (const c128 *)Q, (const c128 *)Qinv, D, basis, act, out0,
                                       out1, (c128 *)Y, N, F, T, K, p.bpw, ref, floor_kind, eps,
                                       redo));
  return check_launch("fmnmf_generic walk");
}

}  // namespace fmg

// ---- entry points used by mnmf_api.hip for shapes outside the MFMA-tile kernels
size_t fmnmf_generic_workspace_doubles(int B, int N, int M, int F, int T) {
  // A, Bt (B,N,F,T) each -- or, for the spatial update, (num, den) of every (bin, n, m) per frame
  // tile; the (B,M,F,T) weights behind them
  const size_t pts = (size_t)B * F * T;
  const size_t traces = pts * (2 * (size_t)N);
  const size_t spatial = (size_t)((T + 63) / 64) * B * F * N * M * 2;
  return (traces > spatial ? traces : spatial) + pts * (size_t)M;
}

int fmnmf_generic_update(const void *X, const void *C, void *Q, double *D, double *basis,
                         double *activation, int B, int N, int M, int F, int T, int K, int steps,
                         int floor_kind, double floor_eps, double *gws, void *U, double *qbuf,
                         int *info, hipStream_t st) {
  using namespace fmg;
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  if (M < 2 || M > 8) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_channels must be in [2, 8]");
  const size_t pts = (size_t)B * F * T;
  const size_t traces = pts * (2 * (size_t)N);
  const size_t spatial = (size_t)((T + 63) / 64) * B * F * N * M * 2;
  double *A = gws, *Bt = gws + pts * N, *Wt = gws + (traces > spatial ? traces : spatial);
  int rc = SSSPY_OK;
  auto traces_pass = [&]() {
    return launch_walk<MODE_TRACES>(X, Q, nullptr, D, basis, activation, A, Bt, nullptr, B, N, M, F,
                                    T, K, 0, 0, 0.0, nullptr, st);
  };
  if (steps & SSSPY_MNMF_BASIS) {
    rc = traces_pass();
    if (rc) return rc;
    const size_t tile = T <= BASIS_TMAX ? (size_t)KT * T * sizeof(double) : 0;
    if (tile > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)k_basis,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile);
      if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
    }
    // runs of bins sized for about 768 workgroups (measured: 43 us against 48 at 1536, 58 at 4096),
    // at least one bin per wave
    const int ktiles = (K + KT - 1) / KT;
    int gx = (768 + ktiles * N * B - 1) / (ktiles * N * B);
    gx = gx > (F + 3) / 4 ? (F + 3) / 4 : gx;
    const int nb = (F + gx - 1) / gx;
    hipLaunchKernelGGL(k_basis, dim3((F + nb - 1) / nb, ktiles, N * B), dim3(256), tile, st, basis,
                       (const double *)activation, (const double *)A, (const double *)Bt, N, F, T, K,
                       nb, floor_kind, floor_eps);
    rc = check_launch("fmnmf_generic basis");
    if (rc) return rc;
  }
  if (steps & SSSPY_MNMF_ACTIVATION) {
    rc = traces_pass();
    if (rc) return rc;
    hipLaunchKernelGGL(k_activation, dim3((T + 63) / 64, (K + KT - 1) / KT, N * B), dim3(64 * AW), 0,
                       st, (const double *)basis, activation, (const double *)A, (const double *)Bt,
                       N, F, T, K, floor_kind, floor_eps);
    rc = check_launch("fmnmf_generic activation");
    if (rc) return rc;
  }
  bool have_q = false;
  if (steps & SSSPY_MNMF_DIAGONALIZER) {
    rc = launch_walk<MODE_WEIGHTS>(X, Q, nullptr, D, basis, activation, Wt, nullptr, nullptr, B, N,
                                   M, F, T, K, 0, 0, 0.0, nullptr, st);
    if (rc) return rc;
    rc = ssspy_weighted_covariance(X, Wt, SSSPY_WEIGHT_BIN_FRAME, U, B, M, M, F, T, (void *)st);
    if (rc) return rc;
    rc = ip1_with_power(Q, U, C, C ? qbuf : nullptr, B, F, M, floor_kind, floor_eps, info, st);
    if (rc) return rc;
    have_q = C != nullptr;
  }
  if (steps & SSSPY_MNMF_SPATIAL) {
    rc = launch_walk<MODE_SPATIAL>(X, Q, nullptr, D, basis, activation, gws, nullptr, nullptr, B, N,
                                   M, F, T, K, 0, 0, 0.0, nullptr, st);
    if (rc) return rc;
    const long long count = (long long)B * F * N * M;
    hipLaunchKernelGGL(k_spatial_fold, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, D,
                       (const double *)gws, count, (T + 63) / 64);
    rc = check_launch("fmnmf_generic spatial fold");
    if (rc) return rc;
  }
  if (steps & SSSPY_MNMF_NORMALIZE) {
    if (!have_q) {
      rc = row_power(Q, C, qbuf, B, F, M, st);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_norm_scale, dim3((F + 63) / 64, B), dim3(256), 0, st, (c128 *)Q, D,
                       (const double *)qbuf, N, M, F, floor_kind, floor_eps);
    rc = check_launch("fmnmf_generic norm_scale");
  }
  return rc;
}

int fmnmf_generic_weights(const void *X, const void *Q, const double *D, const double *basis,
                          const double *act, double *Wt, int B, int N, int M, int F, int T, int K,
                          hipStream_t st) {
  return fmg::launch_walk<fmg::MODE_WEIGHTS>(X, Q, nullptr, D, basis, act, Wt, nullptr, nullptr, B, N,
                                             M, F, T, K, 0, 0, 0.0, nullptr, st);
}

static inline int fmg_loss_slots(int B, int F, int T) {
  const fmg::WalkPlan p = fmg::walk_plan(B, F, T);
  return p.gx * p.gy * fmg::WB;
}
size_t fmnmf_generic_loss_ws_bytes(int B, int F, int T) {
  return scalar_slots_bytes(B, fmg_loss_slots(B, F, T));
}
// out[b] = the data term; loss_ws: fmnmf_generic_loss_ws_bytes() (every wave writes its slot)
int fmnmf_generic_loss(const void *X, const void *Q, const double *D, const double *basis,
                       const double *act, double *out, void *loss_ws, int B, int N, int M, int F,
                       int T, int K, hipStream_t st) {
  const int rc = fmg::launch_walk<fmg::MODE_LOSS>(X, Q, nullptr, D, basis, act, (double *)loss_ws,
                                                  nullptr, nullptr, B, N, M, F, T, K, 0, 0, 0.0,
                                                  nullptr, st);
  return rc ? rc : scalar_slots_fold(loss_ws, B, fmg_loss_slots(B, F, T), out, 0, st);
}

int fmnmf_generic_separate(const void *X, const void *Q, void *Qinv, const double *D,
                           const double *basis, const double *act, void *Y, int B, int N, int M,
                           int F, int T, int K, int ref, int floor_kind, double eps, int *info,
                           int *redo, hipStream_t st) {
  // redo: B F ints of scratch (bins the closed-form launch hands to the general one)
  using namespace fmg;
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  const long long nbins = (long long)B * F;
  hipError_t e = hipMemsetAsync(redo, 0, (size_t)nbins * sizeof(int), st);
  if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  FMG_DISPATCH_M(M, hipLaunchKernelGGL((k_qinv<MM>), dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0,
                                       st, (const c128 *)Q, (c128 *)Qinv, nbins, info));
  int rc = check_launch("fmnmf_generic qinv");
  if (rc) return rc;
  rc = launch_walk<MODE_SEPARATE>(X, Q, Qinv, D, basis, act, nullptr, nullptr, Y, B, N, M, F, T, K,
                                  ref, floor_kind, eps, redo, st);
  if (rc) return rc;
  FMG_DISPATCH_M(M, hipLaunchKernelGGL((k_separate<MM, true>), dim3(F, B), dim3(128), 0, st,
                                       (const c128 *)X, (const c128 *)Q, (const c128 *)Qinv, D, basis,
                                       act, (c128 *)Y, N, F, T, K, ref, floor_kind, eps, redo));
  return check_launch("fmnmf_generic separate");
}

// The Wiener filter split at the eigenvalue floor of to_psd (an arbitrary flooring callable):
// stage 1 leaves ascending eigenvalues lam (B,F,T,M) and eigenvectors P (B,F,T,M,M); the host floors
// lam; stage 2 finishes.  Any N, M <= 8.
int fmnmf_generic_separate_eig(const void *X, const void *Q, void *Qinv, const double *D,
                               const double *basis, const double *act, void *Y, int B, int N,
                               int M, int F, int T, int K, int ref, int stage, double *lam,
                               void *P, int *info, hipStream_t st) {
  using namespace fmg;
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  const long long nbins = (long long)B * F;
  FMG_DISPATCH_M(M, {
    if (stage == 1) {
      hipLaunchKernelGGL((k_qinv<MM>), dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                         (const c128 *)Q, (c128 *)Qinv, nbins, info);
      hipLaunchKernelGGL((k_separate<MM, true, 1>), dim3(F, B), dim3(128), 0, st, (const c128 *)X,
                         (const c128 *)Q, (const c128 *)Qinv, D, basis, act, (c128 *)Y, N, F, T, K,
                         ref, SSSPY_FLOOR_NONE, 0.0, (int *)nullptr, lam, (c128 *)P);
    } else {
      hipLaunchKernelGGL((k_separate<MM, true, 2>), dim3(F, B), dim3(128), 0, st, (const c128 *)X,
                         (const c128 *)Q, (const c128 *)Qinv, D, basis, act, (c128 *)Y, N, F, T, K,
                         ref, SSSPY_FLOOR_NONE, 0.0, (int *)nullptr, lam, (c128 *)P);
    }
  });
  return check_launch("fmnmf_generic separate (eigen stages)");
}

}  // namespace ssspy
