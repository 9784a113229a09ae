// GaussMNMF: multichannel NMF with full-rank spatial covariance matrices (no partitioning).
//
// Model: lambda_nij = sum_k t_nik v_nkj, H_ni (M x M Hermitian), R_ij = to_psd(sum_n lambda_nij H_ni).
// Every (bin, frame) point owns an M x M eigenproblem (the eigenvalue floor of to_psd), so the
// unit of work is "one lane = one point": the lane forms R, runs the Jacobi sweeps in registers,
// and gets R^-1 = P diag(1/floor(lam)) P^H for free from the same decomposition.  The reference's
// instantaneous covariance XX_ij = to_psd(x x^H) is never materialised: its eigenvalues are
// (|x|^2, 0, ..., 0), so after the floor it is  c1 * x x^H + c0 * I  with
//   max floor: c1 = (max(|x|^2, eps) - eps) / |x|^2, c0 = eps;  add floor: c1 = 1, c0 = eps.
// Traces of the MM rules then reduce to quadratic forms in u = R^-1 x:
//   tr(R^-1 XX R^-1 H_n) = c1 u^H H_n u + c0 tr(R^-1 H_n R^-1),   tr(R^-1 H_n).
//
// replaces: ssspy/bss/mnmf.py:681-1073 (GaussMNMF), :300-414 (MNMF), special/psd.py, linalg/mean.py.
#include "common.hpp"
#include "hermitian.hpp"
#include "ssspy_amd.h"

namespace ssspy {

constexpr int GM_NMAX = SSSPY_MAX_SOURCES;

// coefficients of XX = c1 x x^H + c0 I after the eigenvalue floor
__device__ __forceinline__ void xx_floor_coeffs(double s, int floor_kind, double eps, double &c1,
                                                double &c0) {
  if (floor_kind == SSSPY_FLOOR_MAX) {
    c0 = eps;
    c1 = s > 0.0 ? (fmax(s, eps) - eps) / s : 0.0;
  } else if (floor_kind == SSSPY_FLOOR_ADD) {
    c0 = eps;
    c1 = 1.0;
  } else {
    c0 = 0.0;
    c1 = 1.0;
  }
}

// Per-point state: lambda_n, R^-1 (Hermitian), log det R (both of the floored R), u = R^-1 x.
template <int M>
struct Point {
  double lam[GM_NMAX];
  c128 Rinv[M][M];
  double logdet;
  c128 x[M], u[M];
};

// R^-1 and log det of to_psd(R).  The eigenvalue floor rarely does anything (R is a positive
// combination of PSD matrices), so the common path avoids the eigen-decomposition:
//   add floor:  to_psd(R) = R + eps I exactly -> Cholesky of that;
//   max floor:  if 1 / ||R^-1||_F > eps then every eigenvalue exceeds eps and to_psd(R) = R;
//   otherwise (or when Cholesky meets a non-positive pivot) the Jacobi path applies the floor.
template <int M>
__device__ __forceinline__ void psd_inverse(c128 (&R)[M][M], c128 (&Rinv)[M][M], double &logdet,
                                            int floor_kind, double eps) {
  hermitize<M>(R);
  c128 Lw[M][M];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = 0; c < M; ++c) Lw[a][c] = R[a][c];
  if (floor_kind == SSSPY_FLOOR_ADD) {
#pragma unroll
    for (int a = 0; a < M; ++a) Lw[a][a].x += eps;
  }
  bool ok = chol_inverse<M>(Lw, Rinv, logdet);
  if (ok && floor_kind == SSSPY_FLOOR_MAX) {
    double fro = 0.0;
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int c = 0; c < M; ++c) fro += cabs2(Rinv[a][c]);
    ok = fro * eps * eps < 1.0;  // 1 / ||R^-1||_F > eps
  }
  if (!ok) {
    c128 P[M][M];
    double ev[M], w[M];
    psd_eigen<M>(R, P, ev, floor_kind, eps);
    double ld = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      w[k] = 1.0 / ev[k];
      ld += log(ev[k]);
    }
    herm_rebuild<M>(P, w, Rinv);
    logdet = ld;
  }
}

// Hs: spatial matrices of this bin in LDS [n][M*M]; Ts: basis rows of this bin in LDS [n][K]
template <int M>
__device__ __forceinline__ void point_setup(Point<M> &pt, const c128 *__restrict__ Xb,
                                            const double *__restrict__ act_b, const c128 *Hs,
                                            const double *Ts, int N, int F, int T, int K, int i,
                                            int j, int floor_kind, double eps) {
  c128 R[M][M];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = 0; c < M; ++c) R[a][c] = cmake(0.0, 0.0);
#pragma unroll
  for (int n = 0; n < GM_NMAX; ++n) {
    double l = 0.0;
    if (n < N) {
      for (int k = 0; k < K; ++k) l = fma(Ts[n * K + k], act_b[((long long)n * K + k) * T + j], l);
#pragma unroll
      for (int a = 0; a < M; ++a)
#pragma unroll
        for (int c = 0; c < M; ++c) {
          const c128 h = Hs[n * M * M + a * M + c];
          R[a][c].x = fma(l, h.x, R[a][c].x);
          R[a][c].y = fma(l, h.y, R[a][c].y);
        }
    }
    pt.lam[n] = l;
  }
  psd_inverse<M>(R, pt.Rinv, pt.logdet, floor_kind, eps);
#pragma unroll
  for (int m = 0; m < M; ++m) pt.x[m] = Xb[((long long)m * F + i) * T + j];
#pragma unroll
  for (int a = 0; a < M; ++a) {
    c128 s = cmake(0.0, 0.0);
#pragma unroll
    for (int c = 0; c < M; ++c) cfma(s, pt.Rinv[a][c], pt.x[c]);
    pt.u[a] = s;
  }
}

// stage H[b, :, i] and basis[b, :, i, :] of one bin in LDS
template <int M>
__device__ __forceinline__ void stage_bin(c128 *Hs, double *Ts, const c128 *__restrict__ H,
                                          const double *__restrict__ basis, int b, int N, int F,
                                          int K, int i) {
  for (int e = threadIdx.x; e < N * M * M; e += blockDim.x) {
    const int n = e / (M * M), rem = e % (M * M);
    Hs[e] = H[(((long long)b * N + n) * F + i) * (M * M) + rem];
  }
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int n = e / K, k = e % K;
    Ts[e] = basis[(((long long)b * N + n) * F + i) * K + k];
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------ traces
// A[b,n,i,j] = tr(R^-1 XX R^-1 H_n), Bt[b,n,i,j] = tr(R^-1 H_n).  grid: (ceil(T/128), F, B)
template <int M>
__global__ __launch_bounds__(128) void k_gmnmf_traces(const c128 *__restrict__ X,
                                                      const double *__restrict__ basis,
                                                      const double *__restrict__ act,
                                                      const c128 *__restrict__ H,
                                                      double *__restrict__ A,
                                                      double *__restrict__ Bt, int N, int F, int T,
                                                      int K, int floor_kind, double eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Hs = reinterpret_cast<c128 *>(smem);
  double *Ts = reinterpret_cast<double *>(Hs + N * M * M);
  const int i = blockIdx.y, b = blockIdx.z;
  stage_bin<M>(Hs, Ts, H, basis, b, N, F, K, i);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= T) return;
  Point<M> pt;
  point_setup<M>(pt, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Hs, Ts, N, F, T,
                 K, i, j, floor_kind, eps);
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m) s += cabs2(pt.x[m]);
  double c1, c0;
  xx_floor_coeffs(s, floor_kind, eps, c1, c0);
  // tr(R^-1 H_n) = sum_ak Re(Rinv_ak H_ka) and tr(R^-1 H_n R^-1) = sum_ak Re(R2_ak H_ka) with
  // R2 = R^-1 R^-1 formed once per point: M^2 products per source instead of M^3
  c128 R2[M][M];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = a; c < M; ++c) {
      c128 r2 = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) cfma(r2, pt.Rinv[a][k], pt.Rinv[k][c]);
      R2[a][c] = r2;
      R2[c][a] = cconj(r2);
    }
  for (int n = 0; n < N; ++n) {
    const c128 *Hn = Hs + n * M * M;
    double trG = 0.0, trGR = 0.0;
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const c128 h = Hn[k * M + a];
        trG = fma(pt.Rinv[a][k].x, h.x, trG);
        trG = fma(-pt.Rinv[a][k].y, h.y, trG);
        trGR = fma(R2[a][k].x, h.x, trGR);
        trGR = fma(-R2[a][k].y, h.y, trGR);
      }
    double q = 0.0;
#pragma unroll
    for (int a = 0; a < M; ++a) {
      c128 hu = cmake(0.0, 0.0);
#pragma unroll
      for (int c = 0; c < M; ++c) cfma(hu, Hn[a * M + c], pt.u[c]);
      q = fma(pt.u[a].x, hu.x, q);
      q = fma(pt.u[a].y, hu.y, q);
    }
    const long long o = (((long long)b * N + n) * F + i) * T + j;
    A[o] = fma(c1, q, c0 * trGR);
    Bt[o] = trG;
  }
}

// basis[b,n,i,k] <- floor(basis * sqrt(sum_j V A / sum_j V Bt)).  grid: (F, N, B), 256 threads;
// wave w takes k = w, w+4, ...
// raw != NULL (partitioning): the (num, den) pairs go to raw[b,n,i,k,2] instead
__global__ __launch_bounds__(256) void k_gmnmf_basis(double *basis, const double *__restrict__ act,
                                                     const double *__restrict__ A,
                                                     const double *__restrict__ Bt, int N, int F,
                                                     int T, int K, int floor_kind, double eps,
                                                     double *raw) {
  const int i = blockIdx.x, n = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (((long long)b * N + n) * F + i) * T;
  for (int k = wave; k < K; k += 4) {
    const double *v = act + (((long long)b * N + n) * K + k) * T;
    double sn = 0.0, sd = 0.0;
    for (int j = lane; j < T; j += 64) {
      const double vv = v[j];
      sn = fma(vv, A[row + j], sn);
      sd = fma(vv, Bt[row + j], sd);
    }
    sn = wave_sum(sn);
    sd = wave_sum(sd);
    if (lane == 0) {
      const long long o = (((long long)b * N + n) * F + i) * K + k;
      if (raw) {
        raw[2 * o] = sn;
        raw[2 * o + 1] = sd;
      } else {
        basis[o] = apply_floor(basis[o] * sqrt(sn / sd), floor_kind, eps);
      }
    }
  }
}

// ---- partitioning (latent variables Z): shared t (B,F,K), v (B,K,T); the kernels above run on the
// expansion Teff = z t, Vrep = v and these recombine their per-source sums.
// ref: ssspy/bss/mnmf.py:836-901, :903-968, :1018-1073 (partitioning branches).
__global__ __launch_bounds__(256) void k_gm_expand(const double *__restrict__ basis,
                                                   const double *__restrict__ act,
                                                   const double *__restrict__ latent,
                                                   double *__restrict__ Teff,
                                                   double *__restrict__ Vrep, int N, int F, int T,
                                                   int K) {
  const int n = blockIdx.y, b = blockIdx.z;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const double *z = latent + ((long long)b * N + n) * K;
  if (e < (long long)F * K)
    Teff[((long long)b * N + n) * F * K + e] = z[e % K] * basis[(long long)b * F * K + e];
  if (e < (long long)K * T) Vrep[((long long)b * N + n) * K * T + e] = act[(long long)b * K * T + e];
}

// t_ik <- floor(t_ik sqrt(sum_n z_nk S_nik / sum_n z_nk D_nik)).  one thread per (b, i, k)
__global__ __launch_bounds__(256) void k_gm_part_basis(const double *__restrict__ raw,
                                                       const double *__restrict__ latent,
                                                       double *basis, int N, int F, int K,
                                                       int floor_kind, double eps) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)F * K) return;
  const int k = (int)(e % K);
  double sn = 0.0, sd = 0.0;
  for (int n = 0; n < N; ++n) {
    const double z = latent[((long long)b * N + n) * K + k];
    const double *r = raw + (((long long)b * N + n) * F * K + e) * 2;
    sn = fma(z, r[0], sn);
    sd = fma(z, r[1], sd);
  }
  double *dst = basis + (long long)b * F * K + e;
  *dst = apply_floor(*dst * sqrt(sn / sd), floor_kind, eps);
}

// v_kj <- floor(v_kj sqrt(sum_n num_nkj / sum_n den_nkj)); the sums were taken with Teff and
// carry z_nk already.  acc: [b][n][2][K][T].  one thread per (b, k, j)
__global__ __launch_bounds__(256) void k_gm_part_activation(const double *__restrict__ acc,
                                                            double *act, int N, int K, int T,
                                                            int floor_kind, double eps) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)K * T) return;
  const long long kt = (long long)K * T;
  double sn = 0.0, sd = 0.0;
  for (int n = 0; n < N; ++n) {
    sn += acc[(((long long)b * N + n) * 2) * kt + e];
    sd += acc[(((long long)b * N + n) * 2 + 1) * kt + e];
  }
  double *dst = act + (long long)b * kt + e;
  *dst = apply_floor(*dst * sqrt(sn / sd), floor_kind, eps);
}

// z_nk <- z_nk sqrt(sum_i t_ik S_nik / sum_i t_ik D_nik), columns renormalised.  grid: (B)
__global__ __launch_bounds__(256) void k_gm_part_latent(const double *__restrict__ raw,
                                                        const double *__restrict__ basis,
                                                        double *latent, int N, int F, int K) {
  extern __shared__ __attribute__((aligned(16))) double znew[];  // N K doubles (<= 64 KB)
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int n = e / K, k = e % K;
    double sn = 0.0, sd = 0.0;
    for (int i = 0; i < F; ++i) {
      const double t = basis[((long long)b * F + i) * K + k];
      const double *r = raw + ((((long long)b * N + n) * F + i) * K + k) * 2;
      sn = fma(t, r[0], sn);
      sd = fma(t, r[1], sd);
    }
    znew[e] = latent[((long long)b * N + n) * K + k] * sqrt(sn / sd);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int k = e % K;
    double col = 0.0;
    for (int n = 0; n < N; ++n) col += znew[n * K + k];
    latent[(long long)b * N * K + e] = znew[e] / col;
  }
}

// Activation sums: acc[b,n,0,k,j] = sum_i T A, acc[b,n,1,k,j] = sum_i T Bt.
// grid: (ceil(T/64), 1, N*B); wave w walks its quarter of the bins, lanes are frames, the four
// waves fold through LDS in wave order and the sum is STORED: no fp64 atomics, so the activation --
// and with it the whole trajectory -- is the same on every run.

__global__ __launch_bounds__(256) void k_gmnmf_activation_sums(const double *__restrict__ basis,
                                                               const double *__restrict__ A,
                                                               const double *__restrict__ Bt,
                                                               double *__restrict__ acc, int N,
                                                               int F, int T, int K, int k0,
                                                               int bins_per_chunk,
                                                               long long slab_stride) {
  // grid.y = bin chunks (round 4: a handful of mixtures left the chip to (T / 64) N B blocks walking
  // all bins); chunk c stores its sums to slab c (acc + c * slab_stride), k_fold_slabs adds the
  // slabs in chunk order -- one chunk stores straight to the sums
  __shared__ double fold[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int n = blockIdx.z % N, b = blockIdx.z / N;
  const int c_begin = blockIdx.y * bins_per_chunk;
  const int c_end = min(F, c_begin + bins_per_chunk);
  const int per_wave = (c_end - c_begin + 3) >> 2;
  const int i_begin = c_begin + wave * per_wave;
  const int i_end = min(c_end, i_begin + per_wave);
  acc += (long long)blockIdx.y * slab_stride;
  const int jc = min(j, T - 1);
  double sn[8], sd[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) sn[kk] = sd[kk] = 0.0;
  const double *tb = basis + ((long long)b * N + n) * F * K;
  const long long base = ((long long)b * N + n) * F * T + jc;
  for (int i = i_begin; i < i_end; ++i) {
    const double a = A[base + (long long)i * T], bt = Bt[base + (long long)i * T];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const double t = k0 + kk < K ? tb[(long long)i * K + k0 + kk] : 0.0;
      sn[kk] = fma(t, a, sn[kk]);
      sd[kk] = fma(t, bt, sd[kk]);
    }
  }
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    fold[wave][kk][lane] = sn[kk];
    fold[wave][8 + kk][lane] = sd[kk];
  }
  __syncthreads();
  // 16 rows x 64 frames = 1024 sums, 4 per thread
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int row = e >> 6, ln = e & 63;
    const int kk = row & 7, nd = row >> 3;
    const int jj = blockIdx.x * 64 + ln;
    if (jj < T && k0 + kk < K) {
      const double v = fold[0][row][ln] + fold[1][row][ln] + fold[2][row][ln] + fold[3][row][ln];
      acc[((((long long)b * N + n) * 2 + nd) * K + k0 + kk) * T + jj] = v;
    }
  }
}

// act <- floor(act * sqrt(num / den)) from the accumulated sums.  one thread per (b, n, k, j)
__global__ __launch_bounds__(256) void k_gmnmf_activation_apply(double *act,
                                                                const double *__restrict__ acc,
                                                                long long count, int K, int T,
                                                                int floor_kind, double eps) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const long long kt = (long long)K * T;
  const long long bn = e / kt, rem = e % kt;
  const double sn = acc[(bn * 2) * kt + rem], sd = acc[(bn * 2 + 1) * kt + rem];
  act[e] = apply_floor(act[e] * sqrt(sn / sd), floor_kind, eps);
}

// ---------------------------------------------------------------------------- spatial update
// Pacc[b,n,i] = sum_j lambda R^-1 ; Qacc[b,n,i] = sum_j lambda R^-1 XX R^-1, both Hermitian and
// stored packed as M*M doubles (diagonal, then re/im of the upper triangle).  grid: (F, B), one
// wave: lanes take frames, the per-chunk matrices go through LDS and thread (n, entry) folds the
// chunk with the N weights.
constexpr int GM_PB = 64;  // points per chunk (= block size of k_gmnmf_spatial_acc)

template <int M>
__device__ __forceinline__ void pack_hermitian(const c128 (&A)[M][M], double *dst) {
  int e = 0;
#pragma unroll
  for (int a = 0; a < M; ++a) dst[e++] = A[a][a].x;
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = a + 1; c < M; ++c) {
      dst[e++] = A[a][c].x;
      dst[e++] = A[a][c].y;
    }
}

template <int M>
__device__ __forceinline__ void unpack_hermitian(const double *src, c128 (&A)[M][M]) {
  int e = 0;
#pragma unroll
  for (int a = 0; a < M; ++a) A[a][a] = cmake(src[e++], 0.0);
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = a + 1; c < M; ++c) {
      const c128 z = cmake(src[e], src[e + 1]);
      e += 2;
      A[a][c] = z;
      A[c][a] = cconj(z);
    }
}

template <int M>
__global__ __launch_bounds__(GM_PB) void k_gmnmf_spatial_acc(const c128 *__restrict__ X,
                                                           const double *__restrict__ basis,
                                                           const double *__restrict__ act,
                                                           const c128 *__restrict__ H,
                                                           double *__restrict__ PQacc, int N, int F,
                                                           int T, int K, int floor_kind,
                                                           double eps) {
  constexpr int E = 2 * M * M;     // packed doubles per point: R^-1 then R^-1 XX R^-1
  constexpr int ROW = E + GM_NMAX;  // doubles per point in LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Hs = reinterpret_cast<c128 *>(smem);
  double *Ts = reinterpret_cast<double *>(Hs + N * M * M);
  double *pts = Ts + ((N * K + 1) & ~1);  // [GM_PB][ROW]
  const int i = blockIdx.x, b = blockIdx.y;
  stage_bin<M>(Hs, Ts, H, basis, b, N, F, K, i);
  const c128 *Xb = X + (long long)b * M * F * T;
  const double *act_b = act + (long long)b * N * K * T;
  // accumulators: this thread owns output slots idx = tid, tid + GM_PB, ... of the N*E sums
  constexpr int SLOTS = (GM_NMAX * E + GM_PB - 1) / GM_PB;
  double accum[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) accum[s] = 0.0;
  for (int j0 = 0; j0 < T; j0 += GM_PB) {
    const int j = j0 + threadIdx.x;
    double *mine = pts + threadIdx.x * ROW;
    if (j < T) {
      Point<M> pt;
      point_setup<M>(pt, Xb, act_b, Hs, Ts, N, F, T, K, i, j, floor_kind, eps);
      double s = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) s += cabs2(pt.x[m]);
      double c1, c0;
      xx_floor_coeffs(s, floor_kind, eps, c1, c0);
      c128 Q[M][M];  // R^-1 XX R^-1 = c1 u u^H + c0 R^-1 R^-1
#pragma unroll
      for (int a = 0; a < M; ++a)
#pragma unroll
        for (int c = a; c < M; ++c) {
          c128 r2 = cmake(0.0, 0.0);
#pragma unroll
          for (int k = 0; k < M; ++k) cfma(r2, pt.Rinv[a][k], pt.Rinv[k][c]);
          const c128 uu = cmulc(pt.u[a], pt.u[c]);
          Q[a][c] = cmake(fma(c1, uu.x, c0 * r2.x), fma(c1, uu.y, c0 * r2.y));
        }
      pack_hermitian<M>(pt.Rinv, mine);
      pack_hermitian<M>(Q, mine + M * M);
#pragma unroll
      for (int n = 0; n < GM_NMAX; ++n) mine[E + n] = pt.lam[n];
    } else {
      for (int e = 0; e < ROW; ++e) mine[e] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int idx = threadIdx.x + GM_PB * s;
      if (idx < N * E) {
        const int n = idx / E, e = idx % E;
        double v = accum[s];
        for (int p = 0; p < GM_PB; ++p) v = fma(pts[p * ROW + E + n], pts[p * ROW + e], v);
        accum[s] = v;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int idx = threadIdx.x + GM_PB * s;
    if (idx < N * E) {
      const int n = idx / E, e = idx % E;
      PQacc[(((long long)b * N + n) * F + i) * E + e] = accum[s];
    }
  }
}

// H <- to_psd(P^-1 # (H Q H)) with P, HQH floored first.  One lane per (b, n, i).
template <int M>
__global__ __launch_bounds__(64) void k_gmnmf_spatial_update(c128 *H,
                                                             const double *__restrict__ PQacc,
                                                             long long count, int floor_kind,
                                                             double eps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count) return;
  c128 Hm[M][M], Qm[M][M], Tm[M][M], C[M][M], Pv[M][M];
  double lam[M], w[M];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = 0; c < M; ++c) Hm[a][c] = H[idx * (M * M) + a * M + c];
  unpack_hermitian<M>(PQacc + idx * (2 * M * M) + M * M, Qm);
  // HQH, floored
  matmul<M>(Hm, Qm, Tm);
  matmul<M>(Tm, Hm, C);
  psd_eigen<M>(C, Pv, lam, floor_kind, eps);
  c128 HQH[M][M];
  herm_rebuild<M>(Pv, lam, HQH);
  // P floored, P^(1/2), P^(-1/2)
  unpack_hermitian<M>(PQacc + idx * (2 * M * M), C);
  psd_eigen<M>(C, Pv, lam, floor_kind, eps);
  c128 Ph[M][M], Pih[M][M];
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = sqrt(lam[k]);
  herm_rebuild<M>(Pv, w, Ph);
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = 1.0 / w[k];
  herm_rebuild<M>(Pv, w, Pih);
  // (P^1/2 HQH P^1/2)^1/2
  matmul<M>(Ph, HQH, Tm);
  matmul<M>(Tm, Ph, C);
  hermitize<M>(C);
  jacobi_eigh<M>(C, Pv);
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = sqrt(fmax(C[k][k].x, 0.0));
  herm_rebuild<M>(Pv, w, Qm);
  // G = P^-1/2 (...) P^-1/2, floored
  matmul<M>(Pih, Qm, Tm);
  matmul<M>(Tm, Pih, C);
  psd_eigen<M>(C, Pv, lam, floor_kind, eps);
  herm_rebuild<M>(Pv, lam, Hm);
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = 0; c < M; ++c) H[idx * (M * M) + a * M + c] = Hm[a][c];
}

// unit trace of H, the scale goes to the basis: H /= tr H, basis[n, i, :] *= tr H.
// ref: ssspy/bss/mnmf.py:391-414.  One lane per (b, n, i).
__global__ __launch_bounds__(64) void k_gmnmf_normalize(c128 *H, double *basis, long long count,
                                                        int M, int K) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count) return;
  c128 *h = H + idx * (M * M);
  double tr = 0.0;
  for (int a = 0; a < M; ++a) tr += h[a * M + a].x;
  for (int e = 0; e < M * M; ++e) h[e] = cmake(h[e].x / tr, h[e].y / tr);
  if (basis)  // with partitioning the scale cannot move into the shared basis (mnmf.py:404-413)
    for (int k = 0; k < K; ++k) basis[idx * K + k] *= tr;
}

// ------------------------------------------------------------------------------------- loss
// out[b] += sum_i mean_j ( tr(R^-1 XX) + log det R ).  grid: (ceil(T/128), F, B)
template <int M>
__global__ __launch_bounds__(128) void k_gmnmf_loss(const c128 *__restrict__ X,
                                                    const double *__restrict__ basis,
                                                    const double *__restrict__ act,
                                                    const c128 *__restrict__ H, double *out, int N,
                                                    int F, int T, int K, int floor_kind,
                                                    double eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[2];
  c128 *Hs = reinterpret_cast<c128 *>(smem);
  double *Ts = reinterpret_cast<double *>(Hs + N * M * M);
  const int i = blockIdx.y, b = blockIdx.z;
  stage_bin<M>(Hs, Ts, H, basis, b, N, F, K, i);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double term = 0.0;
  if (j < T) {
    Point<M> pt;
    point_setup<M>(pt, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Hs, Ts, N, F,
                   T, K, i, j, floor_kind, eps);
    double s = 0.0, xu = 0.0, trR = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      s += cabs2(pt.x[m]);
      xu = fma(pt.x[m].x, pt.u[m].x, xu);
      xu = fma(pt.x[m].y, pt.u[m].y, xu);
      trR += pt.Rinv[m][m].x;
    }
    double c1, c0;
    xx_floor_coeffs(s, floor_kind, eps, c1, c0);
    term = fma(c1, xu, c0 * trR) + pt.logdet;
  }
  const double total = block_sum(term, red);
  // one slot per (frame block, bin) of the mixture, [slot][B]; ssspy_gmnmf_loss folds them in order
  if (threadIdx.x == 0)
    out[((long long)blockIdx.y * gridDim.x + blockIdx.x) * gridDim.z + b] = total / (double)T;
}

// --------------------------------------------------------------------------- Wiener filter
// Y[b,n,i,j] = lambda_n (H_n R^-1 x)[ref].  grid: (ceil(T/128), F, B)
template <int M>
__global__ __launch_bounds__(128) void k_gmnmf_separate(const c128 *__restrict__ X,
                                                        const double *__restrict__ basis,
                                                        const double *__restrict__ act,
                                                        const c128 *__restrict__ H,
                                                        c128 *__restrict__ Y, int N, int F, int T,
                                                        int K, int ref, int floor_kind,
                                                        double eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Hs = reinterpret_cast<c128 *>(smem);
  double *Ts = reinterpret_cast<double *>(Hs + N * M * M);
  const int i = blockIdx.y, b = blockIdx.z;
  stage_bin<M>(Hs, Ts, H, basis, b, N, F, K, i);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= T) return;
  Point<M> pt;
  point_setup<M>(pt, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Hs, Ts, N, F, T,
                 K, i, j, floor_kind, eps);
  for (int n = 0; n < N; ++n) {
    c128 y = cmake(0.0, 0.0);
#pragma unroll
    for (int c = 0; c < M; ++c) cfma(y, Hs[n * M * M + ref * M + c], pt.u[c]);
    Y[(((long long)b * N + n) * F + i) * T + j] = cscale(y, pt.lam[n]);
  }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct GmnmfWs {
  size_t a, bt, pq, vacc, teff, vrep, raw, vslabs, total;
};
// bin chunks of the activation sums: enough blocks for the chip at small batches, at most 16
static inline int gm_act_chunks(int B, int N, int F, int T) {
  const long long blocks0 = (long long)((T + 63) / 64) * N * B;
  long long want = (1024 + blocks0 - 1) / blocks0;
  if (want > 16) want = 16;
  if (want > (F + 7) / 8) want = (F + 7) / 8;  // at least 8 bins per chunk
  return want < 1 ? 1 : (int)want;
}
static inline GmnmfWs gmnmf_ws(int B, int N, int M, int F, int T, int K) {
  GmnmfWs w;
  size_t off = 0;
  w.a = off;
  off += align256((size_t)B * N * F * T * sizeof(double));
  w.bt = off;
  off += align256((size_t)B * N * F * T * sizeof(double));
  w.pq = off;  // packed Hermitian sums of the spatial update
  off += align256((size_t)B * N * F * M * M * 2 * sizeof(double));
  w.vacc = off;  // activation sums (num, den)
  off += align256((size_t)B * N * 2 * K * T * sizeof(double));
  w.teff = off;  // partitioning: expanded pair and the (num, den) basis sums
  off += align256((size_t)B * N * F * K * sizeof(double));
  w.vrep = off;
  off += align256((size_t)B * N * K * T * sizeof(double));
  w.raw = off;
  off += align256((size_t)B * N * F * K * 2 * sizeof(double));
  w.vslabs = off;  // per-chunk slabs of the activation sums + the scratch of their fold
  {
    const int chunks = gm_act_chunks(B, N, F, T);
    const long long vtotal = 2ll * B * N * K * T;
    off += chunks > 1 ? align256((size_t)chunks * vtotal * sizeof(double) +
                                 fold_scratch_bytes(vtotal, chunks))
                      : 0;
  }
  w.total = off;
  return w;
}

static inline size_t bin_smem(int N, int M, int K) {
  return (size_t)N * M * M * sizeof(c128) + (size_t)((N * K + 1) & ~1) * sizeof(double);
}

#define GM_DISPATCH_M(M_, CALL)                                                              \
  switch (M_) {                                                                              \
    case 2: { constexpr int MM = 2; CALL; } break;                                           \
    case 3: { constexpr int MM = 3; CALL; } break;                                           \
    case 4: { constexpr int MM = 4; CALL; } break;                                           \
    case 5: { constexpr int MM = 5; CALL; } break;                                           \
    case 6: { constexpr int MM = 6; CALL; } break;                                           \
    case 7: { constexpr int MM = 7; CALL; } break;                                           \
    case 8: { constexpr int MM = 8; CALL; } break;                                           \
    default: return fail(SSSPY_ERR_UNSUPPORTED, "GaussMNMF: n_channels must be in [2, 8]");  \
  }

static int check_dims(int B, int N, int M, int F, int T, int K) {
  SSSPY_REQUIRE(B > 0 && F > 0 && T > 0, "GaussMNMF: bad shape");
  SSSPY_REQUIRE(N >= 1 && N <= SSSPY_MAX_SOURCES, "GaussMNMF: n_sources must be in [1, 8]");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "GaussMNMF: n_basis must be in [1, 1024]");
  if (M < 2 || M > 8) return fail(SSSPY_ERR_UNSUPPORTED, "GaussMNMF: n_channels must be in [2, 8]");
  return SSSPY_OK;
}

static int launch_traces(const void *X, const double *basis, const double *act, const void *H,
                         double *A, double *Bt, int B, int N, int M, int F, int T, int K,
                         int floor_kind, double eps, hipStream_t st) {
  dim3 grid((T + 127) / 128, F, B), block(128);
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_traces<MM>), grid, block, bin_smem(N, M, K), st,
                                      (const c128 *)X, basis, act, (const c128 *)H, A, Bt, N, F, T,
                                      K, floor_kind, eps));
  return check_launch("k_gmnmf_traces");
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

size_t ssspy_gmnmf_workspace_bytes(int B, int N, int M, int F, int T, int K) {
  if (B <= 0 || N <= 0 || M <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  return gmnmf_ws(B, N, M, F, T, K).total;
}

int ssspy_gmnmf_update(const void *X, double *basis, double *activation, double *latent,
                       void *spatial, int B, int N, int M, int F, int T, int K, int steps,
                       int floor_kind, double floor_eps, void *workspace, size_t workspace_bytes,
                       void *stream) {
  SSSPY_REQUIRE(X && basis && activation && spatial, "gmnmf_update: null argument");
  SSSPY_REQUIRE(latent || !(steps & SSSPY_GMNMF_LATENT), "gmnmf_update: latent step without latent");
  int rc = check_dims(B, N, M, F, T, K);
  if (rc) return rc;
  const GmnmfWs w = gmnmf_ws(B, N, M, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "gmnmf_update: workspace too small");
  char *ws = (char *)workspace;
  double *A = (double *)(ws + w.a), *Bt = (double *)(ws + w.bt);
  double *PQ = (double *)(ws + w.pq), *vacc = (double *)(ws + w.vacc);
  double *Teff = (double *)(ws + w.teff), *Vrep = (double *)(ws + w.vrep);
  double *raw = (double *)(ws + w.raw);
  hipStream_t st = as_stream(stream);
  const bool part = latent != nullptr;
  // the per-source (basis, activation) pair every kernel takes: the state itself, or the expansion
  const double *Tn = basis, *Vn = activation;
  auto refresh = [&]() -> int {
    if (!part) return SSSPY_OK;
    const long long per = (long long)K * (F > T ? F : T);
    hipLaunchKernelGGL(k_gm_expand, dim3((unsigned)((per + 255) / 256), N, B), dim3(256), 0, st,
                       (const double *)basis, (const double *)activation, (const double *)latent,
                       Teff, Vrep, N, F, T, K);
    Tn = Teff;
    Vn = Vrep;
    return check_launch("k_gm_expand");
  };
  auto basis_sums = [&](double *raw_out) -> int {
    int r = refresh();
    if (r) return r;
    r = launch_traces(X, Tn, Vn, spatial, A, Bt, B, N, M, F, T, K, floor_kind, floor_eps, st);
    if (r) return r;
    hipLaunchKernelGGL(k_gmnmf_basis, dim3(F, N, B), dim3(256), 0, st, basis, Vn,
                       (const double *)A, (const double *)Bt, N, F, T, K, floor_kind, floor_eps,
                       raw_out);
    return check_launch("k_gmnmf_basis");
  };
  if (steps & SSSPY_GMNMF_BASIS) {
    rc = basis_sums(part ? raw : nullptr);
    if (rc) return rc;
    if (part) {
      hipLaunchKernelGGL(k_gm_part_basis, dim3((unsigned)(((long long)F * K + 255) / 256), B),
                         dim3(256), 0, st, (const double *)raw, (const double *)latent, basis, N,
                         F, K, floor_kind, floor_eps);
      rc = check_launch("k_gm_part_basis");
      if (rc) return rc;
    }
  }
  if (steps & SSSPY_GMNMF_ACTIVATION) {
    rc = refresh();
    if (rc) return rc;
    rc = launch_traces(X, Tn, Vn, spatial, A, Bt, B, N, M, F, T, K, floor_kind, floor_eps, st);
    if (rc) return rc;
    const long long count = (long long)B * N * K * T;
    const int chunks = gm_act_chunks(B, N, F, T);
    const int bpc = (F + chunks - 1) / chunks;
    const long long vtotal = 2 * count;  // (num, den) sums
    double *slabs = chunks > 1 ? (double *)(ws + w.vslabs) : vacc;
    for (int k0 = 0; k0 < K; k0 += 8) {
      hipLaunchKernelGGL(k_gmnmf_activation_sums, dim3((T + 63) / 64, chunks, N * B), dim3(256), 0,
                         st, Tn, (const double *)A, (const double *)Bt, slabs, N, F, T, K, k0, bpc,
                         vtotal);
      rc = check_launch("k_gmnmf_activation_sums");
      if (rc) return rc;
    }
    if (chunks > 1) {
      rc = launch_fold_slabs(slabs, (char *)slabs + (size_t)chunks * vtotal * sizeof(double), vacc,
                             vtotal, chunks, st);
      if (rc) return rc;
    }
    if (part) {
      hipLaunchKernelGGL(k_gm_part_activation, dim3((unsigned)(((long long)K * T + 255) / 256), B),
                         dim3(256), 0, st, (const double *)vacc, activation, N, K, T, floor_kind,
                         floor_eps);
      rc = check_launch("k_gm_part_activation");
    } else {
      hipLaunchKernelGGL(k_gmnmf_activation_apply, dim3((unsigned)((count + 255) / 256)),
                         dim3(256), 0, st, activation, (const double *)vacc, count, K, T,
                         floor_kind, floor_eps);
      rc = check_launch("k_gmnmf_activation_apply");
    }
    if (rc) return rc;
  }
  if (steps & SSSPY_GMNMF_SPATIAL) {
    rc = refresh();
    if (rc) return rc;
    const size_t smem = bin_smem(N, M, K) + (size_t)GM_PB * (2 * M * M + GM_NMAX) * sizeof(double);
    GM_DISPATCH_M(M, {
      if (smem > 48 * 1024) {  // 8 channels: 64 points x 136 doubles; gfx950 has 160 KB per CU
        hipError_t e = hipFuncSetAttribute((const void *)k_gmnmf_spatial_acc<MM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
      }
      hipLaunchKernelGGL((k_gmnmf_spatial_acc<MM>), dim3(F, B), dim3(GM_PB), smem, st,
                         (const c128 *)X, Tn, Vn, (const c128 *)spatial, PQ, N, F, T, K,
                         floor_kind, floor_eps);
    });
    rc = check_launch("k_gmnmf_spatial_acc");
    if (rc) return rc;
    const long long count = (long long)B * N * F;
    GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_spatial_update<MM>),
                                        dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st,
                                        (c128 *)spatial, (const double *)PQ, count, floor_kind,
                                        floor_eps));
    rc = check_launch("k_gmnmf_spatial_update");
    if (rc) return rc;
  }
  if (steps & SSSPY_GMNMF_NORMALIZE) {
    const long long count = (long long)B * N * F;
    hipLaunchKernelGGL(k_gmnmf_normalize, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st,
                       (c128 *)spatial, part ? (double *)nullptr : basis, count, M, K);
    rc = check_launch("k_gmnmf_normalize");
    if (rc) return rc;
  }
  if (steps & SSSPY_GMNMF_LATENT) {
    rc = basis_sums(raw);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gm_part_latent, dim3(B), dim3(256), (size_t)N * K * sizeof(double), st,
                       (const double *)raw,
                       (const double *)basis, latent, N, F, K);
    rc = check_launch("k_gm_part_latent");
    if (rc) return rc;
  }
  return SSSPY_OK;
}

size_t ssspy_gmnmf_loss_workspace_bytes(int B, int F, int T) {
  if (B <= 0 || F <= 0 || T <= 0) return 0;
  return scalar_slots_bytes(B, ((T + 127) / 128) * F);
}

int ssspy_gmnmf_loss(const void *X, const double *basis, const double *activation,
                     const void *spatial, double *out, int B, int N, int M, int F, int T, int K,
                     int floor_kind, double floor_eps, void *workspace, size_t workspace_bytes,
                     void *stream) {
  SSSPY_REQUIRE(X && basis && activation && spatial && out, "gmnmf_loss: null argument");
  int rc = check_dims(B, N, M, F, T, K);
  if (rc) return rc;
  SSSPY_REQUIRE(workspace && workspace_bytes >= ssspy_gmnmf_loss_workspace_bytes(B, F, T),
                "gmnmf_loss: workspace too small (ssspy_gmnmf_loss_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  dim3 grid((T + 127) / 128, F, B), block(128);
  // (every block writes its slot; the fold stores out[b])
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_loss<MM>), grid, block, bin_smem(N, M, K), st,
                                      (const c128 *)X, basis, activation, (const c128 *)spatial,
                                      (double *)workspace, N, F, T, K, floor_kind, floor_eps));
  rc = check_launch("k_gmnmf_loss");
  return rc ? rc : scalar_slots_fold(workspace, B, (int)grid.x * F, out, 0, st);
}

int ssspy_gmnmf_separate(const void *X, const double *basis, const double *activation,
                         const void *spatial, void *Y, int B, int N, int M, int F, int T, int K,
                         int reference_id, int floor_kind, double floor_eps, void *stream) {
  SSSPY_REQUIRE(X && basis && activation && spatial && Y, "gmnmf_separate: null argument");
  int rc = check_dims(B, N, M, F, T, K);
  if (rc) return rc;
  SSSPY_REQUIRE(reference_id >= 0 && reference_id < M, "gmnmf_separate: bad reference_id");
  dim3 grid((T + 127) / 128, F, B), block(128);
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_separate<MM>), grid, block, bin_smem(N, M, K),
                                      as_stream(stream), (const c128 *)X, basis, activation,
                                      (const c128 *)spatial, (c128 *)Y, N, F, T, K, reference_id,
                                      floor_kind, floor_eps));
  return check_launch("k_gmnmf_separate");
}

}  // extern "C"
