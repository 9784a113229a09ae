// Shared per-bin operators: separate, (weighted / cross) covariance, IP1, ISS1 transform,
// projection back, log-determinant.  See include/ssspy_amd.h for the contract of each entry.
#include <cstdlib>

#include "common.hpp"
#include "cov_core.hpp"
#include "smallmat.hpp"
#include "wide_n.hpp"

namespace ssspy {
// ilrma_fast.hip (one translation unit per N)
int ilrma_fast_wcov_frame_n2(const void *, const double *, void *, int, int, int, hipStream_t);
int ilrma_fast_wcov_frame_n3(const void *, const double *, void *, int, int, int, hipStream_t);
int ilrma_fast_wcov_frame_n4(const void *, const double *, void *, int, int, int, hipStream_t);

// wide_cov.hip: 5..8 channels on the matrix cores
bool wide_weighted_cov_ok(int N, int S, int F, int T, int kind);
int wide_weighted_cov(const void *A, const double *weight, int kind, void *U, int B, int N, int S,
                      int F, int T, hipStream_t st);

thread_local char g_last_error[512] = "";

// ---------------------------------------------------------------------------------- separate
// One block per (bin, mixture); lanes run along frames so every channel row is read and
// every source row written as contiguous 16-byte elements.  W_i is block-uniform: its rows are read
// through scalar loads (the index depends on blockIdx and loop counters only) and enter the FMAs as
// SGPR operands -- an LDS copy costs one ds_read per complex product, which bound the kernel at
// 8 sources (0.59 ms for 16 mixtures of N = 8, F = 1025, T = 512: 0.9 TB/s).
// POWER: write |y|^2 as (B, N, F, T) f64 into `Y` instead of y (the grouped NMF passes of a wide
// mixture read nothing else of y: half the bytes written here and read there).
template <int N, bool POWER = false>
__global__ __launch_bounds__(256) void k_separate(const c128 *__restrict__ X,
                                                  const c128 *__restrict__ W, c128 *Y, int F,
                                                  int T) {
  const int i = blockIdx.x, b = blockIdx.y;
  const c128 *__restrict__ w = W + ((long long)b * F + i) * (N * N);
  const long long row0 = ((long long)b * N) * F + i;
  // two frames per thread and pass; the loop over sources stays rolled so that one row of W (2 N
  // scalars) is live at a time instead of the whole matrix spilling out of the SGPR file
  for (int j0 = 0; j0 < T; j0 += 2 * blockDim.x) {
    const int ja = j0 + threadIdx.x, jb = ja + blockDim.x;
    const bool va = ja < T, vb = jb < T;
    c128 xa[N], xb[N];
#pragma unroll
    for (int m = 0; m < N; ++m) {
      xa[m] = va ? X[(row0 + (long long)m * F) * T + ja] : cmake(0.0, 0.0);
      xb[m] = vb ? X[(row0 + (long long)m * F) * T + jb] : cmake(0.0, 0.0);
    }
#pragma unroll 1
    for (int n = 0; n < N; ++n) {
      c128 ya = cmake(0.0, 0.0), yb = cmake(0.0, 0.0);
#pragma unroll
      for (int m = 0; m < N; ++m) {
        const c128 wv = w[n * N + m];
        cfma(ya, wv, xa[m]);
        cfma(yb, wv, xb[m]);
      }
      if constexpr (POWER) {
        double *P = reinterpret_cast<double *>(Y);
        if (va) P[(row0 + (long long)n * F) * T + ja] = cabs2(ya);
        if (vb) P[(row0 + (long long)n * F) * T + jb] = cabs2(yb);
      } else {
        if (va) Y[(row0 + (long long)n * F) * T + ja] = ya;
        if (vb) Y[(row0 + (long long)n * F) * T + jb] = yb;
      }
    }
  }
}

// ------------------------------------------------------------------------ covariance congruence
// Cout_i = G_i C_i G_i^H per bin: the covariance of y' = G y from the covariance of y without a pass
// over the spectrogram.  One thread per output element (any n_sources; N^2 complex products each).
__global__ __launch_bounds__(256) void k_covariance_congruence(const c128 *__restrict__ C,
                                                               const c128 *__restrict__ G,
                                                               c128 *__restrict__ Cout,
                                                               long long nbins, int sets, int N) {
  // C, Cout: (nbins, sets, N, N); G: (nbins, N, N), shared by the sets of its bin
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nbins * sets * N * N) return;
  const long long mat = e / (N * N);
  const int rc = (int)(e - mat * (N * N)), r = rc / N, c = rc - r * N;
  const c128 *Cb = C + mat * (long long)(N * N), *Gb = G + (mat / sets) * (long long)(N * N);
  c128 acc = cmake(0.0, 0.0);
  for (int k = 0; k < N; ++k) {
    c128 t = cmake(0.0, 0.0);  // (C G^H)[k][c]
    for (int l = 0; l < N; ++l) cfma(t, Cb[k * N + l], cconj(Gb[c * N + l]));
    cfma(acc, Gb[r * N + k], t);
  }
  Cout[e] = acc;
}

// The same with one lane per matrix and everything in registers (N <= 4: 16 + 16 loads per lane
// instead of 24 per output element; 32 mixtures of configs[1]: 62 -> see DESIGN 4 item 44).
//
// AMP: also how far the product can round from the sum over the samples it replaces.  The diagonal of
// the result, (G C G^H)_rr = mean phi |y_r|^2, is a sum of positive terms when taken over the
// samples; as a product its terms cancel, and eps * S_rr with S_rr = sum_kl |g_rk| |c_kl| |g_rl|
// bounds the rounding error (off the diagonal: <= eps * sqrt(S_rr S_cc)).  kappa_r = S_rr / (.)_rr
// is therefore the factor by which row r's statistics -- and through the update the row of the
// output they steer -- can lose relative accuracy.  A silent source in a loud bin has a large
// kappa by construction (the filter is there to cancel the loud ones) and no weight in the
// result, so the launch reports the power-weighted mean square: per mixture
//   amp[b] = { sum kappa_n^2 p_n, sum p_n }  over bins and sets n (sets == N),
// kappa_n = max_r kappa_r of set n's matrix -- the statistics under source n's weights, which
// steer output row n only -- and p_n = g_n P g_n^H the power of that row (P: the unweighted
// covariance of the data G applies to); a non-positive diagonal counts as infinite.  eps * sqrt(amp[b][0] / amp[b][1]) estimates the relative Frobenius
// error the route adds to mixture b's spectrogram per iteration.  amp is one half of a ring of
// two: the launch accumulates into its half (zero on entry) and clears the other for the next.
template <int N, bool AMP>
__global__ __launch_bounds__(256) void k_covariance_congruence_n(
    const c128 *__restrict__ C, const c128 *__restrict__ G, c128 *__restrict__ Cout,
    long long nmats, int sets, const c128 *__restrict__ P, double *__restrict__ amp,
    double *__restrict__ amp_next, long long mats_per_mixture, int n_mixtures) {
  long long mat = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (AMP && blockIdx.x == 0)
    for (int e = threadIdx.x; e < 2 * n_mixtures; e += blockDim.x) amp_next[e] = 0.0;
  const bool live = mat < nmats;
  if (!AMP && !live) return;
  if (!live) mat = nmats - 1;  // (AMP: the whole wave reaches the shuffles; the copy stores nothing)
  const c128 *Cb = C + mat * (N * N), *Gb = G + (mat / sets) * (N * N);
  c128 g[N][N], c[N][N], t[N][N];
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int k = 0; k < N; ++k) {
      g[r][k] = Gb[r * N + k];
      c[r][k] = Cb[r * N + k];
    }
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int cc = 0; cc < N; ++cc) {  // t = C G^H
      c128 a = cmake(0.0, 0.0);
#pragma unroll
      for (int l = 0; l < N; ++l) cfma(a, c[k][l], cconj(g[cc][l]));
      t[k][cc] = a;
    }
  double diag[N];
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int cc = 0; cc < N; ++cc) {
      c128 a = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) cfma(a, g[r][k], t[k][cc]);
      if (live) Cout[mat * (N * N) + r * N + cc] = a;
      if (AMP && r == cc) diag[r] = a.x;
    }
  if (AMP) {
    // (sets == N: matrix n of a bin holds the statistics under source n's weights, and every entry
    //  of it steers output row n and no other: y_n <- y_n - (V_n[n,r] / V_n[r,r]) y_r)
    const c128 *Pb = P + (mat / sets) * (N * N);
    const int own = (int)(mat % sets);
    double kmax = 0.0, pown = 0.0;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      double ag[N], s = 0.0;
      c128 q = cmake(0.0, 0.0);  // g_r P g_r^H
#pragma unroll
      for (int k = 0; k < N; ++k) ag[k] = sqrt(cabs2(g[r][k]));
#pragma unroll
      for (int k = 0; k < N; ++k) {
        double u = 0.0;
        c128 pg = cmake(0.0, 0.0);
#pragma unroll
        for (int l = 0; l < N; ++l) {
          u = fma(sqrt(cabs2(c[k][l])), ag[l], u);
          cfma(pg, Pb[k * N + l], cconj(g[r][l]));
        }
        s = fma(ag[k], u, s);
        cfma(q, g[r][k], pg);
      }
      const double d = diag[r];
      kmax = fmax(kmax, d > 0.0 ? s / d : __builtin_huge_val());
      if (r == own) pown = fmax(q.x, 0.0);
    }
    double e2 = kmax * kmax * pown, ptot = pown;
    if (!live) e2 = ptot = 0.0;
    const long long b = mat / mats_per_mixture;
    const long long b0 = __shfl(b, 0);
    if (__all(b == b0)) {  // (a wave inside one mixture: one pair of atomics)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        e2 += __shfl_xor(e2, o);
        ptot += __shfl_xor(ptot, o);
      }
      if ((threadIdx.x & 63) == 0) {
        atomicAdd(amp + 2 * b, e2);
        atomicAdd(amp + 2 * b + 1, ptot);
      }
    } else if (live) {
      atomicAdd(amp + 2 * b, e2);
      atomicAdd(amp + 2 * b + 1, ptot);
    }
  }
}

// out_i = G_i W_i per bin (the demixing filters an output-side update y <- G y implies)
__global__ __launch_bounds__(256) void k_compose_filters(const c128 *__restrict__ G,
                                                         const c128 *__restrict__ W,
                                                         c128 *__restrict__ out, long long nbins,
                                                         int N) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nbins * N * N) return;
  const long long bin = e / (N * N);
  const int rc = (int)(e - bin * (N * N)), r = rc / N, c = rc - r * N;
  const c128 *Gb = G + bin * (long long)(N * N), *Wb = W + bin * (long long)(N * N);
  c128 acc = cmake(0.0, 0.0);
  for (int k = 0; k < N; ++k) cfma(acc, Gb[r * N + k], Wb[k * N + c]);
  out[e] = acc;
}

// ------------------------------------------------------------------------- weighted covariance
template <int N, int SG, int MODE>
__global__ __launch_bounds__(256) void k_weighted_cov(const c128 *__restrict__ A,
                                                      const double *__restrict__ weight,
                                                      c128 *__restrict__ U, int S_total, int F,
                                                      int T) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, s0 = blockIdx.z * SG;
  const int i0 = blockIdx.x * 16;
  const int bin = min(i0 + c, F - 1);
  CovAcc<N, SG> acc;
  acc.clear();
  const int ntiles = (T + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j = jt * 16 + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j + r;
      const bool valid = jj < T;
      const int jc = valid ? jj : T - 1;
      c128 x[N];
#pragma unroll
      for (int m = 0; m < N; ++m) x[m] = A[(((long long)b * N + m) * F + bin) * T + jc];
      double phi[SG];
#pragma unroll
      for (int s = 0; s < SG; ++s) {
        const int sg = s0 + s;
        double p = 0.0;
        if (valid && sg < S_total) {
          if (MODE == SSSPY_WEIGHT_UNIT) p = 1.0;
          if (MODE == SSSPY_WEIGHT_FRAME) p = weight[((long long)b * S_total + sg) * T + jc];
          if (MODE == SSSPY_WEIGHT_BIN_FRAME)
            p = weight[(((long long)b * S_total + sg) * F + bin) * T + jc];
        }
        phi[s] = p;
      }
      acc.add(x, phi);
    }
  }
  acc.fold_q();
  cov_reduce_store<N, SG>(acc, lds, U, (long long)b * F, i0, F, S_total, s0,
                          min(SG, S_total - s0), 1.0 / (double)T);
}

template <int N>
constexpr int cov_group() {
  return N <= 4 ? N : (N == 5 ? 3 : 2);
}

template <int N, int MODE>
static int launch_weighted_cov(const c128 *A, const double *weight, c128 *U, int B, int S, int F,
                               int T, hipStream_t st) {
  constexpr int SG = (MODE == SSSPY_WEIGHT_UNIT) ? 1 : cov_group<N>();
  const int groups = (S + SG - 1) / SG;
  dim3 grid((F + 15) / 16, B, groups), block(256);
  const size_t lds = 4 * cov_lds_doubles_per_wave<N, SG>() * sizeof(double);
  hipLaunchKernelGGL((k_weighted_cov<N, SG, MODE>), grid, block, lds, st, A, weight, U, S, F, T);
  return check_launch("k_weighted_cov");
}

template <int N>
static int dispatch_weighted_cov(const c128 *A, const double *weight, int kind, c128 *U, int B,
                                 int S, int F, int T, hipStream_t st) {
  switch (kind) {
    case SSSPY_WEIGHT_UNIT:
      return launch_weighted_cov<N, SSSPY_WEIGHT_UNIT>(A, weight, U, B, S, F, T, st);
    case SSSPY_WEIGHT_FRAME:
      return launch_weighted_cov<N, SSSPY_WEIGHT_FRAME>(A, weight, U, B, S, F, T, st);
    case SSSPY_WEIGHT_BIN_FRAME:
      return launch_weighted_cov<N, SSSPY_WEIGHT_BIN_FRAME>(A, weight, U, B, S, F, T, st);
  }
  return fail(SSSPY_ERR_BADARG, "weighted_covariance: unknown weight_kind");
}

// ---------------------------------------------------------------------------- cross covariance
// C[a][c] = (1/T) sum_j A_a conj(B_c); one block per (bin, mixture), lanes along frames,
// block-wide tree fold per output (small outputs: Na*Nb <= 64).
template <int NA, int NB>
__global__ __launch_bounds__(256) void k_cross_cov(const c128 *__restrict__ A,
                                                   const c128 *__restrict__ Bm,
                                                   c128 *__restrict__ C, int F, int T) {
  __shared__ double scratch[2 * 4];
  const int i = blockIdx.x, b = blockIdx.y;
  c128 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int c = 0; c < NB; ++c) acc[a][c] = cmake(0.0, 0.0);
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    c128 xa[NA], xb[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) xa[a] = A[(((long long)b * NA + a) * F + i) * T + j];
#pragma unroll
    for (int c = 0; c < NB; ++c) xb[c] = Bm[(((long long)b * NB + c) * F + i) * T + j];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const c128 z = cmulc(xa[a], xb[c]);
        acc[a][c] = cadd(acc[a][c], z);
      }
  }
  const double scale = 1.0 / (double)T;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const double re = block_sum(acc[a][c].x, scratch);
      const double im = block_sum(acc[a][c].y, scratch + 4);
      if (threadIdx.x == 0)
        C[(((long long)b * F + i) * NA + a) * NB + c] = cmake(re * scale, im * scale);
    }
}

// ---------------------------------------------------------------------------------------- IP1
// One lane per (mixture, bin): N sequential projections, each an N x N complex solve.
// When C (static covariance) and qbuf are given, also writes the output power of the updated
// rows, qbuf[bin][n] = Re(w_n C w_n^H), which the power normalisation needs.
template <int N>
__global__ __launch_bounds__(64) void k_ip1(c128 *W, const c128 *__restrict__ U, long long nbins,
                                            int floor_kind, double eps, int *info,
                                            const c128 *__restrict__ C, double *qbuf) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Wm;
  load_mat<N>(Wm, W + idx * (N * N));
  bool ok = true;
#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    Mat<N> Un, A;
    load_mat<N>(Un, U + (idx * N + n) * (N * N));
    matmul<N>(A, Wm, Un);
    c128 w[N];
    ok = solve_unit<N>(A, n, w) && ok;
    double qf = quad_form<N>(w, Un);
    qf = qf < 0.0 ? 0.0 : qf;  // np.maximum(., 0): NaN propagates
    const double d = apply_floor(sqrt(qf), floor_kind, eps);
    // row n of W <- conj(w) / d  (static row select keeps W in registers)
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c)
        if (r == n) Wm.a[r][c] = cmake(w[c].x / d, -w[c].y / d);
  }
  store_mat<N>(Wm, W + idx * (N * N));
  if (!ok && info) atomicAdd(info, 1);
  if (C && qbuf) {
    Mat<N> Cm;
    load_mat<N>(Cm, C + idx * (N * N));
#pragma unroll
    for (int n = 0; n < N; ++n) {
      // y_n = sum_m W[n][m] x_m  =>  E|y_n|^2 = conj(v)^H C conj(v) with v = row n of W
      c128 v[N];
#pragma unroll
      for (int m = 0; m < N; ++m) v[m] = cconj(Wm.a[n][m]);
      qbuf[idx * N + n] = quad_form<N>(v, Cm);
    }
  }
}

// ---- IP1 one source at a time, for flooring functions that cannot run in a kernel: the solve of
// source n writes the UNNORMALISED row conj(w) and d = sqrt(max(Re(w^H U_n w), 0)) per bin, the host
// applies its callable to d (B x F doubles), k_scale_filter_row divides the row.
// ref: ssspy/bss/_update_spatial_model.py:63-76.
template <int N>
__global__ __launch_bounds__(64) void k_ip1_source_solve(c128 *W, const c128 *__restrict__ U,
                                                         double *__restrict__ denom,
                                                         long long nbins, int n, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Wm, Un, A;
  load_mat<N>(Wm, W + idx * (N * N));
  load_mat<N>(Un, U + (idx * N + n) * (N * N));
  matmul<N>(A, Wm, Un);
  c128 w[N];
  const bool ok = solve_unit<N>(A, n, w);
  double qf = quad_form<N>(w, Un);
  qf = qf < 0.0 ? 0.0 : qf;  // np.maximum(., 0): NaN propagates
  denom[idx] = sqrt(qf);
#pragma unroll
  for (int c = 0; c < N; ++c) W[idx * (N * N) + n * N + c] = cconj(w[c]);
  if (!ok && info) atomicAdd(info, 1);
}

__global__ __launch_bounds__(256) void k_scale_filter_row(c128 *W, const double *__restrict__ denom,
                                                          long long nbins, int N, int n) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (bin, column)
  if (e >= nbins * N) return;
  const long long idx = e / N;
  const int c = (int)(e - idx * N);
  const double d = denom[idx];
  c128 *p = W + idx * (N * N) + n * N + c;
  *p = cmake(p->x / d, p->y / d);
}

// The same update with a bin spread over G lanes (G = 8 for 5..8 sources): lane r owns row r of the
// filter and of the product A = W U_n.  The LU solve of A w = e_n is row-distributed -- the pivot
// (largest |re| + |im| among the rows not yet used, the lowest row on ties: LAPACK's choice) is found
// by three exchange steps, its lane broadcasts the row, the other unused rows eliminate; the back
// substitution broadcasts one unknown per step -- so every lane ends with the whole w, the arithmetic
// of a row is the one of lu_forward / lu_backward, and no lane ever holds an N x N matrix (the
// one-lane-per-bin kernel above spills 900 registers at N = 8, and a version with the filter in LDS
// and the covariance streamed still 325: 1.1 resp. 0.56 ms for 16 x 1025 bins; this one 0.26 ms).  The covariance entries are read where they are used: the G lanes of a bin
// ask for the same address (one request).  (Four lanes per bin for N <= 4 at one mixture was tried
// for latency: 115.5 -> 114.3 us per iteration, not kept.)
// (pivots inverted once, by crecip_fast, and kept for the back substitution; one divide for the
//  normalisation: 80 IEEE divides per bin at 8 sources before round 6)
template <int N, int G>
__global__ __launch_bounds__(256) void k_ip1_rows(c128 *W, const c128 *__restrict__ U,
                                                  long long nbins, int floor_kind, double eps,
                                                  int *info, const c128 *__restrict__ C,
                                                  double *qbuf) {
  static_assert(G == 4 || G == 8, "group of 4 or 8 lanes");
  static_assert(N <= G, "one lane per row");
  const int r = threadIdx.x % G;  // my row
  const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const bool live = idx < nbins;
  const long long id = live ? idx : nbins - 1;  // idle groups shadow the last bin, never store
  const bool row = r < N;
  const int rr = row ? r : N - 1;
  c128 Wr[N];
#pragma unroll
  for (int c = 0; c < N; ++c) Wr[c] = W[id * (N * N) + rr * N + c];
  bool ok = true;
#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    const c128 *Un = U + (id * N + n) * (N * N);
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
      int bl = r;
#pragma unroll
      for (int m = 1; m < G; m <<= 1) {
        const double ov = __shfl_xor(bv, m, G);
        const int ol = __shfl_xor(bl, m, G);
        const bool take = ov > bv || (ov == bv && ol < bl);
        bv = take ? ov : bv;
        bl = take ? ol : bl;
      }
      plane[k] = bl;
      if (r == bl) order = k;
      c128 prow[N];
#pragma unroll
      for (int c = k; c < N; ++c)
        prow[c] = cmake(__shfl(a[c].x, bl, G), __shfl(a[c].y, bl, G));
      const c128 prhs = cmake(__shfl(rhs.x, bl, G), __shfl(rhs.y, bl, G));
      const c128 piv = prow[k];
      ok = ok && (piv.x != 0.0 || piv.y != 0.0);
      pinv[k] = crecip_fast(piv);
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
    // Re(w^H U_n w): lane r takes row r
    c128 t = cmake(0.0, 0.0);
#pragma unroll
    for (int b2 = 0; b2 < N; ++b2) cfma(t, Un[rr * N + b2], w[b2]);
    double q = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c)
      if (c == rr) q = w[c].x * t.x + w[c].y * t.y;
    q = row ? q : 0.0;
    // (summed in row order on every lane, like quad_form: a tree would round differently)
    double qf = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c) qf += __shfl(q, c, G);
    qf = qf < 0.0 ? 0.0 : qf;  // np.maximum(., 0): NaN propagates
    const double d = apply_floor(sqrt(qf), floor_kind, eps);
    if (r == n) {
      const double dinv = 1.0 / d;
#pragma unroll
      for (int c = 0; c < N; ++c) Wr[c] = cmake(w[c].x * dinv, -w[c].y * dinv);
    }
  }
  if (live && row) {
#pragma unroll
    for (int c = 0; c < N; ++c) W[idx * (N * N) + r * N + c] = Wr[c];
    if (!ok && info && r == 0) atomicAdd(info, 1);
    if (C && qbuf) {
      // y_r = sum_m W[r][m] x_m  =>  E|y_r|^2 = conj(v)^H C conj(v) with v = row r of W
      const c128 *Cm = C + idx * (N * N);
      c128 v[N];
#pragma unroll
      for (int m = 0; m < N; ++m) v[m] = cconj(Wr[m]);
      double q = 0.0;
#pragma unroll
      for (int a2 = 0; a2 < N; ++a2) {
        c128 t = cmake(0.0, 0.0);
#pragma unroll
        for (int b2 = 0; b2 < N; ++b2) cfma(t, Cm[a2 * N + b2], v[b2]);
        q += v[a2].x * t.x + v[a2].y * t.y;
      }
      qbuf[idx * N + r] = q;
    }
  }
}

// qbuf[bin][n] = Re(w_n C w_n^H) for the current W (stand-alone power statistic)
template <int N>
__global__ __launch_bounds__(64) void k_row_power(const c128 *__restrict__ W,
                                                  const c128 *__restrict__ C, double *qbuf,
                                                  long long nbins) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Wm, Cm;
  load_mat<N>(Wm, W + idx * (N * N));
  load_mat<N>(Cm, C + idx * (N * N));
#pragma unroll
  for (int n = 0; n < N; ++n) {
    c128 v[N];
#pragma unroll
    for (int m = 0; m < N; ++m) v[m] = cconj(Wm.a[n][m]);
    qbuf[idx * N + n] = quad_form<N>(v, Cm);
  }
}

// -------------------------------------------------------------------------------- ISS1 transform
// Runs the N rank-1 steering steps on the per-bin statistics.  With G the transform
// accumulated so far, the current covariance of weight set s is G V0_s G^H; only two of
// its entries are needed per (step n, set s): [s, n] and [n, n].
template <int N>
__global__ __launch_bounds__(64) void k_iss1_transform(const c128 *__restrict__ Vc, c128 *G,
                                                       long long nbins, int floor_kind,
                                                       double eps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Gm;
  set_identity<N>(Gm);
  const c128 *V0 = Vc + idx * (long long)(N * N * N);
#pragma unroll 1
  for (int n = 0; n < N; ++n) {
    c128 gn[N];
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c)
        if (r == n) gn[c] = Gm.a[r][c];
    c128 v[N];
    double den_n = 1.0;
#pragma unroll 1
    for (int s = 0; s < N; ++s) {
      // t = V0_s gn^H
      c128 t[N];
#pragma unroll
      for (int a = 0; a < N; ++a) {
        c128 acc = cmake(0.0, 0.0);
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const c128 u = V0[(s * N + a) * N + d];
          // acc += u * conj(gn[d])
          acc.x = fma(u.x, gn[d].x, acc.x);
          acc.x = fma(u.y, gn[d].y, acc.x);
          acc.y = fma(u.y, gn[d].x, acc.y);
          acc.y = fma(-u.x, gn[d].y, acc.y);
        }
        t[a] = acc;
      }
      c128 num = cmake(0.0, 0.0), dn = cmake(0.0, 0.0);
#pragma unroll
      for (int r = 0; r < N; ++r)
#pragma unroll
        for (int a = 0; a < N; ++a)
          if (r == s) cfma(num, Gm.a[r][a], t[a]);
#pragma unroll
      for (int a = 0; a < N; ++a) cfma(dn, gn[a], t[a]);
      const double den = apply_floor(dn.x, floor_kind, eps);
      const double inv = 1.0 / den;
      c128 vs = cmake(num.x * inv, num.y * inv);
      if (s == n) {
        den_n = den;
        vs = cmake(1.0 - 1.0 / sqrt(den), 0.0);
      }
#pragma unroll
      for (int r = 0; r < N; ++r)
        if (r == s) v[r] = vs;
    }
    (void)den_n;
    // G <- (I - v e_n^T) G : every row r loses v_r * (old row n)
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) cfms(Gm.a[r][c], v[r], gn[c]);
  }
  store_mat<N>(Gm, G + idx * (N * N));
}

// ------------------------------------------------------------------------------ projection back
template <int N>
__global__ __launch_bounds__(64) void k_pb_filter(c128 *W, c128 *G, long long nbins, int ref,
                                                  int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Wm, A, Inv;
  load_mat<N>(Wm, W + idx * (N * N));
  A = Wm;
  const bool ok = invert<N>(A, Inv);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    c128 s = cmake(0.0, 0.0);
#pragma unroll
    for (int r = 0; r < N; ++r)
      if (r == ref) s = Inv.a[r][n];
#pragma unroll
    for (int c = 0; c < N; ++c) Wm.a[n][c] = cmul(Wm.a[n][c], s);
    if (G) {
#pragma unroll
      for (int c = 0; c < N; ++c) G[idx * (N * N) + n * N + c] = (c == n) ? s : cmake(0.0, 0.0);
    }
  }
  store_mat<N>(Wm, W + idx * (N * N));
  if (!ok && info) atomicAdd(info, 1);
}

// G = diag(conj(z_n)), z_n = (sum_j y_n conj(x_ref)) / (sum_j |y_n|^2), from the covariances
// YX[n][m] = mean_j y_n conj(x_m) and YY[n][n] = mean_j |y_n|^2.
// ref: ssspy/algorithm/minimal_distortion_principle.py:6-43 (reference_id given).
__global__ __launch_bounds__(64) void k_mdp_scale(const c128 *__restrict__ YX,
                                                  const c128 *__restrict__ YY, c128 *G,
                                                  long long nbins, int N, int ref) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  for (int n = 0; n < N; ++n) {
    const c128 num = YX[idx * (N * N) + n * N + ref];
    const double den = YY[idx * (N * N) + n * N + n].x;
    for (int c = 0; c < N; ++c)
      G[idx * (N * N) + n * N + c] = (c == n) ? cmake(num.x / den, -num.y / den) : cmake(0.0, 0.0);
  }
}

// basis[b,n,i,:] *= |G[b,i,n,n]|^p  (ref: ssspy/bss/ilrma.py:518-522)
__global__ __launch_bounds__(256) void k_scale_basis_by_diag(double *basis,
                                                             const c128 *__restrict__ G, int N,
                                                             int F, int K, double p) {
  const int i = blockIdx.x, n = blockIdx.y, b = blockIdx.z;
  const c128 g = G[(((long long)b * F + i) * N + n) * N + n];
  const double a2 = cabs2(g);
  const double sc = (p == 2.0) ? a2 : pow(a2, 0.5 * p);
  double *row = basis + (((long long)b * N + n) * F + i) * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) row[k] *= sc;
}

template <int N>
__global__ __launch_bounds__(64) void k_pb_scale(const c128 *__restrict__ XY,
                                                 const c128 *__restrict__ YY, c128 *G,
                                                 long long nbins, int ref, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> A, Inv, Gm;
  load_mat<N>(A, YY + idx * (N * N));
  const bool ok = invert<N>(A, Inv);
  const c128 *xy = XY + idx * (N * N) + ref * N;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    c128 s = cmake(0.0, 0.0);
#pragma unroll
    for (int c = 0; c < N; ++c) cfma(s, xy[c], Inv.a[c][n]);
#pragma unroll
    for (int c = 0; c < N; ++c) Gm.a[n][c] = (c == n) ? s : cmake(0.0, 0.0);
  }
  store_mat<N>(Gm, G + idx * (N * N));
  if (!ok && info) atomicAdd(info, 1);
}

template <int N>
__global__ __launch_bounds__(64) void k_demix_from_cov(const c128 *__restrict__ YX,
                                                       const c128 *__restrict__ XX, c128 *W,
                                                       long long nbins, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> A, Inv, Y, Wm;
  load_mat<N>(A, XX + idx * (N * N));
  load_mat<N>(Y, YX + idx * (N * N));
  const bool ok = invert<N>(A, Inv);
  matmul<N>(Wm, Y, Inv);
  store_mat<N>(Wm, W + idx * (N * N));
  if (!ok && info) atomicAdd(info, 1);
}

// one block per mixture (256 threads; THREADS = 1024 for a handful of mixtures of up to 4 sources:
// 1025 bins on 256 threads are five dependent 4 x 4 eliminations per thread, 15.8 us -- a third of
// the loss bookkeeping of a one-mixture run; one bin per thread: see profiles/r05_call_timeline.txt)
template <int N, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void k_sum_logdet(const c128 *__restrict__ W, double *out,
                                                        int F) {
  __shared__ double scratch[THREADS / 64];
  const int b = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < F; i += blockDim.x) {
    Mat<N> A;
    load_mat<N>(A, W + ((long long)b * F + i) * (N * N));
    s += logabsdet<N>(A);
  }
  const double total = block_sum(s, scratch);
  if (threadIdx.x == 0) out[b] = total;
}

}  // namespace ssspy

using namespace ssspy;

namespace ssspy {
// ilrma_small.hip: the latency form (16-bin workgroups, four lanes per bin, operands in LDS)
#define DECL_SMALL_IP1(n)                                                                          \
  int ilrma_small_ip1_n##n(const void *, int, int, long long, const void *, void *, int, int, int, \
                           double, double *, int *, hipStream_t);                                  \
  int ilrma_small_ip1_logdet_n##n(const void *, int, int, long long, const void *, void *, int,    \
                                  int, int, double, double *, int *, double *, long long,          \
                                  hipStream_t);
DECL_SMALL_IP1(2) DECL_SMALL_IP1(3) DECL_SMALL_IP1(4)
#undef DECL_SMALL_IP1

static bool small_ip1_off() {
  static const bool off = std::getenv("SSSPY_AMD_NO_FAST") != nullptr;  // (with every tuned path)
  return off;
}

// Does the latency form of IP1 (ilrma_small.hip) serve this shape?  It also folds the partial
// records a split covariance pass leaves (ip1_from_records), so callers ask before they fold.
bool ip1_small_shape(int B, int F, int N) {
  return !small_ip1_off() && N >= 2 && N <= 4 && (long long)B * F <= 16384;
}

// IP1 (+ output power) straight from the partial covariance records of a pass whose every item was
// split: records[(b * groups + group) * nchunks + ch] hold rbins bins x N^3 complex each, rec_stride
// c128 apart.  Only for ip1_small_shape().
// logdet (optional): the kernel also leaves sum log|det W| of the filters as they come in, one share
// per tile of 16 bins at logdet[tile * logdet_stride + b] (k_ip1_small)
int ip1_from_records(void *W, const void *records, int nchunks, int rbins, long long rec_stride,
                     const void *C, double *qbuf, int B, int F, int N, int floor_kind,
                     double floor_eps, int *info, hipStream_t st, double *logdet,
                     long long logdet_stride) {
  if (logdet) switch (N) {
    case 2: return ilrma_small_ip1_logdet_n2(records, nchunks, rbins, rec_stride, C, W, B, F, floor_kind, floor_eps, qbuf, info, logdet, logdet_stride, st);
    case 3: return ilrma_small_ip1_logdet_n3(records, nchunks, rbins, rec_stride, C, W, B, F, floor_kind, floor_eps, qbuf, info, logdet, logdet_stride, st);
    case 4: return ilrma_small_ip1_logdet_n4(records, nchunks, rbins, rec_stride, C, W, B, F, floor_kind, floor_eps, qbuf, info, logdet, logdet_stride, st);
    default: return fail(SSSPY_ERR_INTERNAL, "ip1_from_records: shape off the small path");
  }
  switch (N) {
    case 2: return ilrma_small_ip1_n2(records, nchunks, rbins, rec_stride, C, W, B, F, floor_kind, floor_eps, qbuf, info, st);
    case 3: return ilrma_small_ip1_n3(records, nchunks, rbins, rec_stride, C, W, B, F, floor_kind, floor_eps, qbuf, info, st);
    case 4: return ilrma_small_ip1_n4(records, nchunks, rbins, rec_stride, C, W, B, F, floor_kind, floor_eps, qbuf, info, st);
    default: return fail(SSSPY_ERR_INTERNAL, "ip1_from_records: shape off the small path");
  }
}

int ip1_with_power(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N,
                   int floor_kind, double floor_eps, int *info, hipStream_t st) {
  const long long nbins = (long long)B * F;
  if (rt_sources_ok(N)) return rt_ip1(W, U, C, qbuf, B, F, N, floor_kind, floor_eps, info, st);
  // a handful of mixtures: one lane per bin would leave the chip to 17 waves running a chain of
  // ~5000 dependent fp64 instructions each (17 us at 1025 bins); four lanes per bin take 12
  if (ip1_small_shape(B, F, N))
    return ip1_from_records(W, U, 0, 0, 0, C, qbuf, B, F, N, floor_kind, floor_eps, info, st, nullptr, 0);
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  if (N > 4) {  // a bin on 8 lanes, one per row
    dim3 g8((unsigned)((nbins * 8 + 255) / 256)), b8(256);
    switch (N) {
      case 5: hipLaunchKernelGGL((k_ip1_rows<5, 8>), g8, b8, 0, st, (c128 *)W, (const c128 *)U, nbins, floor_kind, floor_eps, info, (const c128 *)C, qbuf); break;
      case 6: hipLaunchKernelGGL((k_ip1_rows<6, 8>), g8, b8, 0, st, (c128 *)W, (const c128 *)U, nbins, floor_kind, floor_eps, info, (const c128 *)C, qbuf); break;
      case 7: hipLaunchKernelGGL((k_ip1_rows<7, 8>), g8, b8, 0, st, (c128 *)W, (const c128 *)U, nbins, floor_kind, floor_eps, info, (const c128 *)C, qbuf); break;
      case 8: hipLaunchKernelGGL((k_ip1_rows<8, 8>), g8, b8, 0, st, (c128 *)W, (const c128 *)U, nbins, floor_kind, floor_eps, info, (const c128 *)C, qbuf); break;
      default: return fail(SSSPY_ERR_UNSUPPORTED, "n_sources must be in [1, 8]");
    }
    return check_launch("k_ip1_rows");
  }
  DISPATCH_N4(N, hipLaunchKernelGGL((k_ip1<NN>), grid, block, 0, st, (c128 *)W, (const c128 *)U,
                                    nbins, floor_kind, floor_eps, info, (const c128 *)C, qbuf));
  return check_launch("k_ip1");
}

// P[b, n, i, j] = |(W_i x_ij)_n|^2
int separate_power(const void *X, const void *W, double *P, int B, int N, int F, int T,
                   hipStream_t st) {
  if (rt_sources_ok(N)) return rt_separate(X, W, P, B, N, F, T, true, st);
  dim3 grid(F, B), block(256);
  DISPATCH_N(N, hipLaunchKernelGGL((k_separate<NN, true>), grid, block, 0, st, (const c128 *)X,
                                   (const c128 *)W, (c128 *)P, F, T));
  return check_launch("k_separate_power");
}

int row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st) {
  if (rt_sources_ok(N)) return rt_row_power(W, C, qbuf, B, F, N, st);
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_row_power<NN>), grid, block, 0, st, (const c128 *)W,
                                   (const c128 *)C, qbuf, nbins));
  return check_launch("k_row_power");
}
}  // namespace ssspy

extern "C" {

const char *ssspy_amd_version(void) { return "ssspy_amd 0.2.0 (gfx950)"; }
int ssspy_abi_version(void) { return SSSPY_ABI_VERSION; }
const char *ssspy_last_error(void) { return g_last_error; }

int ssspy_separate(const void *X, const void *W, void *Y, int B, int N, int F, int T,
                   void *stream) {
  SSSPY_REQUIRE(X && W && Y && B > 0 && F > 0 && T > 0, "separate: bad argument");
  if (rt_sources_ok(N)) return rt_separate(X, W, Y, B, N, F, T, false, as_stream(stream));
  dim3 grid(F, B), block(256);
  DISPATCH_N(N, hipLaunchKernelGGL((k_separate<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)X, (const c128 *)W, (c128 *)Y, F, T));
  return check_launch("k_separate");
}

static int congruence_sets(const void *C, const void *G, void *Cout, int B, int F, int S, int N,
                           const void *P, double *amp, int phase, void *stream) {
  SSSPY_REQUIRE(C && G && Cout && C != Cout && B > 0 && F > 0 && S >= 1 && N >= 1 &&
                    N <= SSSPY_RT_MAX_SOURCES,
                "covariance_congruence: bad argument");
  const long long nbins = (long long)B * F, total = nbins * S * N * N;
  if (N >= 2 && N <= 4) {
    const long long nmats = nbins * S;
    const dim3 grid((unsigned)((nmats + 255) / 256)), block(256);
    double *mine = amp ? amp + (size_t)(phase & 1) * 2 * B : nullptr;
    double *next = amp ? amp + (size_t)(~phase & 1) * 2 * B : nullptr;
#define SSSPY_CONGRUENCE(NN)                                                                      \
  if (N == NN) {                                                                                  \
    if (amp)                                                                                      \
      hipLaunchKernelGGL((k_covariance_congruence_n<NN, true>), grid, block, 0, as_stream(stream), \
                         (const c128 *)C, (const c128 *)G, (c128 *)Cout, nmats, S,                \
                         (const c128 *)P, mine, next, (long long)F * S, B);                       \
    else                                                                                          \
      hipLaunchKernelGGL((k_covariance_congruence_n<NN, false>), grid, block, 0,                  \
                         as_stream(stream), (const c128 *)C, (const c128 *)G, (c128 *)Cout, nmats, \
                         S, (const c128 *)nullptr, mine, next, (long long)F * S, B);              \
  }
    SSSPY_CONGRUENCE(2) SSSPY_CONGRUENCE(3) SSSPY_CONGRUENCE(4)
#undef SSSPY_CONGRUENCE
    return check_launch("k_covariance_congruence_n");
  }
  SSSPY_REQUIRE(!amp, "covariance_congruence: the amplification is tracked for 2..4 sources");
  hipLaunchKernelGGL(k_covariance_congruence, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), (const c128 *)C, (const c128 *)G, (c128 *)Cout, nbins, S, N);
  return check_launch("k_covariance_congruence");
}

int ssspy_covariance_congruence_sets(const void *C, const void *G, void *Cout, int B, int F,
                                     int S, int N, void *stream) {
  return congruence_sets(C, G, Cout, B, F, S, N, nullptr, nullptr, 0, stream);
}

int ssspy_covariance_congruence_tracked(const void *C, const void *G, void *Cout, int B, int F,
                                        int S, int N, const void *power, void *amplification,
                                        int phase, void *stream) {
  SSSPY_REQUIRE(amplification && power && S == N,
                "covariance_congruence_tracked: needs the amplification slots, the power and one "
                "set per source");
  return congruence_sets(C, G, Cout, B, F, S, N, power, (double *)amplification, phase, stream);
}

int ssspy_covariance_congruence(const void *C, const void *G, void *Cout, int B, int F, int N,
                                void *stream) {
  return ssspy_covariance_congruence_sets(C, G, Cout, B, F, 1, N, stream);
}

int ssspy_compose_filters(const void *G, const void *W, void *out, int B, int F, int N,
                          void *stream) {
  SSSPY_REQUIRE(G && W && out && out != G && out != W && B > 0 && F > 0 && N >= 1 &&
                    N <= SSSPY_RT_MAX_SOURCES,
                "compose_filters: bad argument");
  const long long nbins = (long long)B * F, total = nbins * N * N;
  hipLaunchKernelGGL(k_compose_filters, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), (const c128 *)G, (const c128 *)W, (c128 *)out, nbins, N);
  return check_launch("k_compose_filters");
}

int ssspy_weighted_covariance(const void *A, const double *weight, int weight_kind, void *U, int B,
                              int N, int S, int F, int T, void *stream) {
  SSSPY_REQUIRE(A && U && B > 0 && F > 0 && T > 0 && S > 0, "weighted_covariance: bad argument");
  SSSPY_REQUIRE(weight_kind == SSSPY_WEIGHT_UNIT || weight, "weighted_covariance: weight is NULL");
  SSSPY_REQUIRE(weight_kind != SSSPY_WEIGHT_UNIT || S == 1, "weighted_covariance: UNIT needs S=1");
  if (rt_sources_ok(N))
    return rt_covariance(A, A, weight, weight_kind, U, B, N, S, F, T, as_stream(stream));
  if (weight_kind == SSSPY_WEIGHT_FRAME && S == N && N >= 2 && N <= 4) {
    // AuxIVA's per-iteration pass: the tuned tile walk when the batch fills the chip
    int rc = -1;
    if (N == 2) rc = ilrma_fast_wcov_frame_n2(A, weight, U, B, F, T, as_stream(stream));
    if (N == 3) rc = ilrma_fast_wcov_frame_n3(A, weight, U, B, F, T, as_stream(stream));
    if (N == 4) rc = ilrma_fast_wcov_frame_n4(A, weight, U, B, F, T, as_stream(stream));
    if (rc >= 0) return rc;
  }
  if (wide_weighted_cov_ok(N, S, F, T, weight_kind))
    return wide_weighted_cov(A, weight, weight_kind, U, B, N, S, F, T, as_stream(stream));
  DISPATCH_N(N, return dispatch_weighted_cov<NN>((const c128 *)A, weight, weight_kind, (c128 *)U,
                                                 B, S, F, T, as_stream(stream)));
  return SSSPY_OK;
}

int ssspy_cross_covariance(const void *A, const void *Bm, void *C, int B, int N, int F, int T,
                           void *stream) {
  SSSPY_REQUIRE(A && Bm && C && B > 0 && F > 0 && T > 0, "cross_covariance: bad argument");
  if (rt_sources_ok(N))
    return rt_covariance(A, Bm, nullptr, SSSPY_WEIGHT_UNIT, C, B, N, 1, F, T, as_stream(stream));
  dim3 grid(F, B), block(256);
  DISPATCH_N(N, hipLaunchKernelGGL((k_cross_cov<NN, NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)A, (const c128 *)Bm, (c128 *)C, F, T));
  return check_launch("k_cross_cov");
}

int ssspy_update_by_ip1(void *W, const void *U, int B, int F, int N, int floor_kind,
                        double floor_eps, int *info, void *stream) {
  SSSPY_REQUIRE(W && U && B > 0 && F > 0, "update_by_ip1: bad argument");
  if (rt_sources_ok(N))
    return rt_ip1(W, U, nullptr, nullptr, B, F, N, floor_kind, floor_eps, info, as_stream(stream));
  // (above 4 sources: a bin on 8 lanes -- the lane-per-bin form spilled 134 / 502 / 888 VGPRs at
  //  6 / 7 / 8 sources and was what AuxIVA-IP1 and the functional update_by_ip1 ran; round 5)
  // (a handful of mixtures of up to 4 sources: the latency form, four lanes per bin -- round 6;
  //  the lane-per-bin kernel below was what a one-mixture AuxIVA-IP1 iteration ran: 17 against 12 us)
  if (N > 4 || ip1_small_shape(B, F, N))
    return ip1_with_power(W, U, nullptr, nullptr, B, F, N, floor_kind, floor_eps, info,
                          as_stream(stream));
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N4(N, hipLaunchKernelGGL((k_ip1<NN>), grid, block, 0, as_stream(stream), (c128 *)W,
                                    (const c128 *)U, nbins, floor_kind, floor_eps, info,
                                    (const c128 *)nullptr, (double *)nullptr));
  return check_launch("k_ip1");
}

int ssspy_update_by_ip1_logdet_slots(int B, int F, int N) {
  if (B <= 0 || F <= 0 || N < 1) return 0;
  return ip1_small_shape(B, F, N) ? (F + 15) / 16 : 1;
}

int ssspy_update_by_ip1_logdet(void *W, const void *U, int B, int F, int N, int floor_kind,
                               double floor_eps, int *info, double *logdet,
                               long long logdet_stride, void *stream) {
  SSSPY_REQUIRE(W && U && B > 0 && F > 0 && logdet && logdet_stride >= B,
                "update_by_ip1_logdet: bad argument");
  if (ip1_small_shape(B, F, N))
    return ip1_from_records(W, U, 0, 0, 0, nullptr, nullptr, B, F, N, floor_kind, floor_eps, info,
                            as_stream(stream), logdet, logdet_stride);
  const int rc = ssspy_sum_logdet(W, logdet, B, F, N, stream);  // (the finished sums: share 0)
  return rc ? rc : ssspy_update_by_ip1(W, U, B, F, N, floor_kind, floor_eps, info, stream);
}

int ssspy_ip1_source_solve(void *W, const void *U, double *denom, int source_idx, int B, int F,
                           int N, int *info, void *stream) {
  SSSPY_REQUIRE(W && U && denom && B > 0 && F > 0, "ip1_source_solve: bad argument");
  SSSPY_REQUIRE(source_idx >= 0 && source_idx < N, "ip1_source_solve: bad source index");
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_ip1_source_solve<NN>), grid, block, 0, as_stream(stream),
                                   (c128 *)W, (const c128 *)U, denom, nbins, source_idx, info));
  return check_launch("k_ip1_source_solve");
}

int ssspy_scale_filter_row(void *W, const double *denom, int source_idx, int B, int F, int N,
                           void *stream) {
  SSSPY_REQUIRE(W && denom && B > 0 && F > 0 && N >= 1 && N <= SSSPY_RT_MAX_SOURCES,
                "scale_filter_row: bad argument");
  SSSPY_REQUIRE(source_idx >= 0 && source_idx < N, "scale_filter_row: bad source index");
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_scale_filter_row, dim3((unsigned)((nbins * N + 255) / 256)), dim3(256), 0,
                     as_stream(stream), (c128 *)W, denom, nbins, N, source_idx);
  return check_launch("k_scale_filter_row");
}

int ssspy_iss1_transform(const void *Vc, void *G, int B, int F, int N, int floor_kind,
                         double floor_eps, void *stream) {
  SSSPY_REQUIRE(Vc && G && B > 0 && F > 0, "iss1_transform: bad argument");
  if (rt_sources_ok(N))
    return rt_iss1_transform(Vc, G, B, F, N, floor_kind, floor_eps, as_stream(stream));
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_iss1_transform<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)Vc, (c128 *)G, nbins, floor_kind, floor_eps));
  return check_launch("k_iss1_transform");
}

int ssspy_projection_back_filter(void *W, void *G, int B, int F, int N, int reference_id, int *info,
                                 void *stream) {
  SSSPY_REQUIRE(W && B > 0 && F > 0 && reference_id >= 0 && reference_id < N,
                "projection_back_filter: bad argument");
  if (rt_sources_ok(N)) return rt_pb_filter(W, G, B, F, N, reference_id, info, as_stream(stream));
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_pb_filter<NN>), grid, block, 0, as_stream(stream),
                                   (c128 *)W, (c128 *)G, nbins, reference_id, info));
  return check_launch("k_pb_filter");
}

int ssspy_mdp_scale(const void *YX, const void *YY, void *G, int B, int F, int N, int reference_id,
                    void *stream) {
  SSSPY_REQUIRE(YX && YY && G && B > 0 && F > 0 && N >= 1 && reference_id >= 0 &&
                    reference_id < N,
                "mdp_scale: bad argument");
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_mdp_scale, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0,
                     as_stream(stream), (const c128 *)YX, (const c128 *)YY, (c128 *)G, nbins, N,
                     reference_id);
  return check_launch("k_mdp_scale");
}

int ssspy_ilrma_scale_basis(double *basis, const void *G, int B, int N, int F, int K, double domain,
                            void *stream) {
  SSSPY_REQUIRE(basis && G && B > 0 && N >= 1 && F > 0 && K >= 1, "scale_basis: bad argument");
  hipLaunchKernelGGL(k_scale_basis_by_diag, dim3(F, N, B), dim3(256), 0, as_stream(stream), basis,
                     (const c128 *)G, N, F, K, domain);
  return check_launch("k_scale_basis_by_diag");
}

int ssspy_projection_back_scale(const void *XY, const void *YY, void *G, int B, int F, int N,
                                int reference_id, int *info, void *stream) {
  SSSPY_REQUIRE(XY && YY && G && B > 0 && F > 0 && reference_id >= 0 && reference_id < N,
                "projection_back_scale: bad argument");
  if (rt_sources_ok(N))
    return rt_pb_scale(XY, YY, G, B, F, N, reference_id, info, as_stream(stream));
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_pb_scale<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)XY, (const c128 *)YY, (c128 *)G, nbins,
                                   reference_id, info));
  return check_launch("k_pb_scale");
}

int ssspy_demix_from_covariance(const void *YX, const void *XX, void *W, int B, int F, int N,
                                int *info, void *stream) {
  SSSPY_REQUIRE(YX && XX && W && B > 0 && F > 0, "demix_from_covariance: bad argument");
  if (rt_sources_ok(N)) return rt_demix_from_cov(YX, XX, W, B, F, N, info, as_stream(stream));
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_demix_from_cov<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)YX, (const c128 *)XX, (c128 *)W, nbins, info));
  return check_launch("k_demix_from_cov");
}

int ssspy_sum_logdet(const void *W, double *out, int B, int F, int N, void *stream) {
  SSSPY_REQUIRE(W && out && B > 0 && F > 0, "sum_logdet: bad argument");
  if (rt_sources_ok(N)) return rt_sum_logdet(W, out, B, F, N, as_stream(stream));
  dim3 grid(B), block(256);
  if (N <= 4 && B <= 64 && F > 256) {
    switch (N) {
      case 1: hipLaunchKernelGGL((k_sum_logdet<1, 1024>), grid, dim3(1024), 0, as_stream(stream), (const c128 *)W, out, F); break;
      case 2: hipLaunchKernelGGL((k_sum_logdet<2, 1024>), grid, dim3(1024), 0, as_stream(stream), (const c128 *)W, out, F); break;
      case 3: hipLaunchKernelGGL((k_sum_logdet<3, 1024>), grid, dim3(1024), 0, as_stream(stream), (const c128 *)W, out, F); break;
      default: hipLaunchKernelGGL((k_sum_logdet<4, 1024>), grid, dim3(1024), 0, as_stream(stream), (const c128 *)W, out, F); break;
    }
    return check_launch("k_sum_logdet");
  }
  DISPATCH_N(N, hipLaunchKernelGGL((k_sum_logdet<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)W, out, F));
  return check_launch("k_sum_logdet");
}

}  // extern "C"
