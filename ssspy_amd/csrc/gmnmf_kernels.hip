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
#include <cstdlib>

#include "common.hpp"
#include "hermitian.hpp"
#include "herm_packed.hpp"
#include "ssspy_amd.h"

namespace ssspy {

// gmnmf_rows.hip
bool gmnmf_spatial_update_rows_wanted(int M);
int gmnmf_spatial_update_rows(void *H, const double *PQ, long long count, int M, int floor_kind,
                              double eps, int *flags, hipStream_t st);

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
    psd_eigen<M, (M >= 6)>(R, P, ev, floor_kind, eps);
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
                                                      int K, int floor_kind, double eps,
                                                      const int *__restrict__ flags) {
  if (flags && !flags[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) return;
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

// basis[b,n,i,k] <- floor(basis * sqrt(sum_j V A / sum_j V Bt)).  grid: (ceil(F / (4 bpw)), N, B), 256
// threads: the workgroup stages the activation rows of its (mixture, source) in LDS, as many basis
// indices at a time as fit, and wave w walks bpw bins against them with the bin's rows of A / Bt in
// registers (T <= 512).  Three shapes of this kernel were timed at 8 mixtures of 4 / 8 channels
// (F = 513, T = 256, K = 8): a workgroup per row with a wave per basis index 59 / 115 us, a wave per
// row re-reading everything 59 / 113 us, this one 55 / 102 us, and with the sums on the DPP path
// (wave_sum_dpp) 45 / 84 us -- 67 / 134 MB of compulsory traffic, so still far from a roof; what
// else paces it is not identified (profiles/r04_gmnmf_m*_b8_kernel_stats.csv).  The sum of a (bin, k) is taken lane-strided over the frames
// and folded by wave_sum_dpp: the same order whatever the grid.
// raw != NULL (partitioning): the (num, den) pairs go to raw[b,n,i,k,2] instead
// Sum over the 64 lanes on the DPP path (row shifts inside the rows of 16, two row broadcasts across
// them), total in every lane by a read of lane 63.  wave_sum() (common.hpp) goes through six
// ds_bpermute pairs per value: 12 trips through the LDS crossbar, and this kernel takes 2 K sums
// per bin.  A fixed order, like wave_sum's.
__device__ __forceinline__ double dpp_shift_add(double v, int ctrl_tag) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int slo, shi;
  switch (ctrl_tag) {  // (the control word must be a compile-time constant)
    case 1:
      slo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);  // row_shr:1
      shi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
      break;
    case 2:
      slo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xF, 0xF, true);  // row_shr:2
      shi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xF, 0xF, true);
      break;
    case 4:
      slo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xF, 0xF, true);  // row_shr:4
      shi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xF, 0xF, true);
      break;
    case 8:
      slo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xF, 0xF, true);  // row_shr:8
      shi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xF, 0xF, true);
      break;
    case 15:
      slo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1, 3
      shi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, true);
      break;
    default:
      slo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2, 3
      shi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, true);
      break;
  }
  return v + __hiloint2double(shi, slo);
}

__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = dpp_shift_add(v, 1);
  v = dpp_shift_add(v, 2);
  v = dpp_shift_add(v, 4);
  v = dpp_shift_add(v, 8);   // lane 15 of every row: the row's sum
  v = dpp_shift_add(v, 15);  // lanes 31, 63: rows 0 + 1, rows 2 + 3
  v = dpp_shift_add(v, 31);  // lane 63: everything
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

constexpr int GMB_LDS = 4096;     // staged activation values (32 KB)
// bins per wave: up to 16, fewer while that leaves the launch under ~2048 workgroups
static inline int gmb_bins_per_wave(int B, int N, int F) {
  const long long rows = (long long)B * N * F;
  long long bpw = rows / (4 * 2048);
  if (bpw < 1) bpw = 1;
  if (bpw > 16) bpw = 16;
  return (int)bpw;
}

__global__ __launch_bounds__(256) void k_gmnmf_basis(double *basis, const double *__restrict__ act,
                                                     const double *__restrict__ A,
                                                     const double *__restrict__ Bt, int N, int F,
                                                     int T, int K, int floor_kind, double eps,
                                                     double *raw, int bpw) {
  __shared__ double vs[GMB_LDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, b = blockIdx.z;
  const int i_begin = (blockIdx.x * 4 + wave) * bpw;
  const int i_end = min(F, i_begin + bpw);
  const double *vbase = act + ((long long)b * N + n) * K * T;
  // frames in slices that fit the LDS tile with at least one basis index; basis indices in chunks
  const int tslice = T <= GMB_LDS ? T : GMB_LDS;  // frames staged at a time
  const int kc = max(1, GMB_LDS / tslice);         // basis indices staged at a time
  if (tslice == T) {
    for (int k0 = 0; k0 < K; k0 += kc) {
      const int kn = min(kc, K - k0);
      __syncthreads();
      for (int e = threadIdx.x; e < kn * T; e += 256) vs[e] = vbase[(long long)k0 * T + e];
      __syncthreads();
      for (int i = i_begin; i < i_end; ++i) {
        const long long row = (((long long)b * N + n) * F + i) * T;
        // the row's traces stay in registers over the basis walk while they fit (T <= 512)
        constexpr int MT = 8;
        double ar[MT], br[MT];
        const bool in_regs = T <= 64 * MT;
        if (in_regs) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int j = lane + 64 * m;
            ar[m] = j < T ? A[row + j] : 0.0;
            br[m] = j < T ? Bt[row + j] : 0.0;
          }
        }
        for (int kk = 0; kk < kn; ++kk) {
          double sn = 0.0, sd = 0.0;
          if (in_regs) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const int j = lane + 64 * m;
              if (j < T) {  // (same frames in the same order as the loop below)
                const double vv = vs[kk * T + j];
                sn = fma(vv, ar[m], sn);
                sd = fma(vv, br[m], sd);
              }
            }
          } else {
            for (int j = lane; j < T; j += 64) {
              const double vv = vs[kk * T + j];
              sn = fma(vv, A[row + j], sn);
              sd = fma(vv, Bt[row + j], sd);
            }
          }
          sn = wave_sum_dpp(sn);
          sd = wave_sum_dpp(sd);
          if (lane == 0) {
            const long long o = (((long long)b * N + n) * F + i) * K + k0 + kk;
            if (raw) {
              raw[2 * o] = sn;
              raw[2 * o + 1] = sd;
            } else {
              basis[o] = apply_floor(basis[o] * sqrt(sn / sd), floor_kind, eps);
            }
          }
        }
      }
    }
  } else {  // more frames than the tile holds: the activation straight from memory
    for (int i = i_begin; i < i_end; ++i) {
      const long long row = (((long long)b * N + n) * F + i) * T;
      for (int k = 0; k < K; ++k) {
        const double *v = vbase + (long long)k * T;
        double sn = 0.0, sd = 0.0;
        for (int j = lane; j < T; j += 64) {
          const double vv = v[j];
          sn = fma(vv, A[row + j], sn);
          sd = fma(vv, Bt[row + j], sd);
        }
        sn = wave_sum_dpp(sn);
        sd = wave_sum_dpp(sd);
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
                                                           double eps,
                                                           const int *__restrict__ flags) {
  if (flags && !flags[blockIdx.y * gridDim.x + blockIdx.x]) return;
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
// the literal update's products: memory-resident from 5 channels on (eight M x M matrices per lane:
// unrolled on registers the kernel spilled 136-5 858 VGPRs at 5-8 channels; it runs for flagged
// blocks only)
template <int M>
__device__ __forceinline__ void su_matmul(const c128 (&A)[M][M], const c128 (&B)[M][M],
                                          c128 (&C)[M][M]) {
  if constexpr (M >= 5) matmul_rolled<M>(A, B, C);
  else matmul<M>(A, B, C);
}
template <int M>
__device__ __forceinline__ void su_rebuild(const c128 (&P)[M][M], const double (&w)[M],
                                           c128 (&Out)[M][M]) {
  if constexpr (M >= 5) herm_rebuild_rolled<M>(P, w, Out);
  else herm_rebuild<M>(P, w, Out);
}

template <int M>
__global__ __launch_bounds__(64) void k_gmnmf_spatial_update(c128 *H,
                                                             const double *__restrict__ PQacc,
                                                             long long count, int floor_kind,
                                                             double eps,
                                                             const int *__restrict__ flags) {
  if (flags && !flags[blockIdx.x]) return;
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
  su_matmul<M>(Hm, Qm, Tm);
  su_matmul<M>(Tm, Hm, C);
  psd_eigen<M, (M >= 5)>(C, Pv, lam, floor_kind, eps);
  c128 HQH[M][M];
  su_rebuild<M>(Pv, lam, HQH);
  // P floored, P^(1/2), P^(-1/2)
  unpack_hermitian<M>(PQacc + idx * (2 * M * M), C);
  psd_eigen<M, (M >= 5)>(C, Pv, lam, floor_kind, eps);
  c128 Ph[M][M], Pih[M][M];
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = sqrt(lam[k]);
  su_rebuild<M>(Pv, w, Ph);
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = 1.0 / w[k];
  su_rebuild<M>(Pv, w, Pih);
  // (P^1/2 HQH P^1/2)^1/2
  su_matmul<M>(Ph, HQH, Tm);
  su_matmul<M>(Tm, Ph, C);
  hermitize<M>(C);
  if constexpr (M >= 5) jacobi_eigh_rolled<M>(C, Pv);
  else jacobi_eigh<M>(C, Pv);
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = sqrt(fmax(C[k][k].x, 0.0));
  su_rebuild<M>(Pv, w, Qm);
  // G = P^-1/2 (...) P^-1/2, floored
  su_matmul<M>(Pih, Qm, Tm);
  su_matmul<M>(Tm, Pih, C);
  psd_eigen<M, (M >= 5)>(C, Pv, lam, floor_kind, eps);
  su_rebuild<M>(Pv, lam, Hm);
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
                                                    double eps, const int *__restrict__ flags) {
  if (flags && !flags[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x]) return;
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
                                                        double eps, int only_marked) {
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  // only_marked: k_gmnmf_separate_p ran first and left NaN in the first source of the points it
  // could not finish (a NaN the data itself produced is recomputed to the same NaN)
  bool mine = j < T;
  if (only_marked && mine) {
    const double probe = Y[(((long long)b * N) * F + i) * T + j].x;
    mine = probe != probe;
  }
  if (only_marked && !__syncthreads_or(mine ? 1 : 0)) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Hs = reinterpret_cast<c128 *>(smem);
  double *Ts = reinterpret_cast<double *>(Hs + N * M * M);
  stage_bin<M>(Hs, Ts, H, basis, b, N, F, K, i);
  if (!mine) return;
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

// ===================================================================== packed per-point path
// The kernels above keep R, its Cholesky factor, R^-1 and R^-2 as four full M x M complex matrices
// per lane: 1 000-1 300 spilled registers at 8 channels, and every spatial matrix entry is an LDS
// read all lanes make at the same address (the LDS return path, not the FMAs, sets their pace).
// The *_p kernels below are the same computations on ONE packed Hermitian matrix per lane, inverted
// in place (herm_packed.hpp), with the spatial matrices and the basis row of the block's bin read
// through the scalar cache (they are wave-uniform: s_load, operands in SGPRs, no LDS at all):
//   R = sum_n lambda_n (H_n + H_n^H) / 2   (what hermitize() leaves of sum_n lambda_n H_n),
//   tr(R^-1 H), tr(E H) for Hermitian R^-1, E: sum_a A_aa Re H_aa + sum_{a<c} [ Re A_ac (Re H_ac +
//   Re H_ca) + Im A_ac (Im H_ac - Im H_ca) ]  -- exact for any H, Hermitian or not.
// They take the fast route of psd_inverse() only (Cholesky, no eigenvalue below the floor); a block
// in which any point leaves it raises flags[block], and the full-storage kernel of the same name
// then recomputes exactly the flagged blocks (launched after every *_p kernel; `flags == nullptr`
// means "all blocks").
template <int M>
struct PointP {
  double lam[GM_NMAX];
  HermP<M> Rinv;
  double logdet;
  c128 x[M], u[M];
  bool ok;
};

// R^-1, log det, x, u from the accumulated R (shared tail of the two point_setup_p forms)
template <int M>
__device__ __forceinline__ void point_finish_p(PointP<M> &pt, const c128 *__restrict__ Xb, int F,
                                               int T, int i, int j, int floor_kind, double eps) {
  if (floor_kind == SSSPY_FLOOR_ADD) {
#pragma unroll
    for (int a = 0; a < M; ++a) pt.Rinv.d[a] += eps;
  }
  bool ok = hp_chol_inverse<M>(pt.Rinv, pt.logdet);
  if (floor_kind == SSSPY_FLOOR_MAX) ok = ok && (hp_fro2<M>(pt.Rinv) * eps * eps < 1.0);
  pt.ok = ok;
#pragma unroll
  for (int m = 0; m < M; ++m) pt.x[m] = Xb[((long long)m * F + i) * T + j];
  hp_matvec<M>(pt.Rinv, pt.x, pt.u);
}

// Hb: spatial matrices of this (mixture, bin), source n at Hb + n * hstride (hstride = F M M);
// Tb: basis rows of this (mixture, bin), source n at Tb + n * tstride (tstride = F K); both uniform
// over the block
template <int M>
__device__ __forceinline__ void point_setup_p(PointP<M> &pt, const c128 *__restrict__ Xb,
                                              const double *__restrict__ act_b,
                                              const c128 *__restrict__ Hb, long long hstride,
                                              const double *__restrict__ Tb, long long tstride,
                                              int N, int F, int T, int K, int i, int j,
                                              int floor_kind, double eps) {
  hp_clear<M>(pt.Rinv);
#pragma unroll
  for (int n = 0; n < GM_NMAX; ++n) {
    double l = 0.0;
    if (n < N) {
      for (int k = 0; k < K; ++k)
        l = fma(Tb[n * tstride + k], act_b[((long long)n * K + k) * T + j], l);
      const c128 *Hn = Hb + n * hstride;
      const double hl = 0.5 * l;
#pragma unroll
      for (int a = 0; a < M; ++a) {
        pt.Rinv.d[a] = fma(l, Hn[a * M + a].x, pt.Rinv.d[a]);
#pragma unroll
        for (int c = a + 1; c < M; ++c) {
          const c128 hu = Hn[a * M + c], hd = Hn[c * M + a];
          c128 &r = pt.Rinv.o[tri<M>(a, c)];
          r.x = fma(hl, hu.x + hd.x, r.x);
          r.y = fma(hl, hu.y - hd.y, r.y);
        }
      }
    }
    pt.lam[n] = l;
  }
  point_finish_p<M>(pt, Xb, F, T, i, j, floor_kind, eps);
}

// The same from the packed symmetric parts k_gm_pack_spatial leaves per (mixture, bin):
// Hq[n * M * M + ...] = diagonal Re H_aa, then per upper entry (Re H_ac + Re H_ca, Im H_ac - Im H_ca);
// sources n >= N hold zeros.  One scalar operand per FMA, no additions, no branches.
template <int M>
__device__ __forceinline__ void point_setup_q(PointP<M> &pt, const c128 *__restrict__ Xb,
                                              const double *__restrict__ act_b,
                                              const double *__restrict__ Hq,
                                              const double *__restrict__ Tb, long long tstride,
                                              int N, int F, int T, int K, int i, int j,
                                              int floor_kind, double eps) {
  hp_clear<M>(pt.Rinv);
#pragma unroll
  for (int n = 0; n < GM_NMAX; ++n) {
    double l = 0.0;
    if (n < N) {
      for (int k = 0; k < K; ++k)
        l = fma(Tb[n * tstride + k], act_b[((long long)n * K + k) * T + j], l);
      const double *Hn = Hq + n * (M * M);
      const double hl = 0.5 * l;
#pragma unroll
      for (int a = 0; a < M; ++a) pt.Rinv.d[a] = fma(l, Hn[a], pt.Rinv.d[a]);
#pragma unroll
      for (int e = 0; e < (M * (M - 1)) / 2; ++e) {
        pt.Rinv.o[e].x = fma(hl, Hn[M + 2 * e], pt.Rinv.o[e].x);
        pt.Rinv.o[e].y = fma(hl, Hn[M + 2 * e + 1], pt.Rinv.o[e].y);
      }
    }
    pt.lam[n] = l;
  }
  point_finish_p<M>(pt, Xb, F, T, i, j, floor_kind, eps);
}

// Hq[b, i, n < 8, M M] from H[b, n, i, M, M]; one thread per (b, i, n, slot)
template <int M>
__global__ __launch_bounds__(256) void k_gm_pack_spatial(const c128 *__restrict__ H,
                                                         double *__restrict__ Hq, int N, int F,
                                                         long long count) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  const int slot = (int)(e % (M * M));
  const int n = (int)((e / (M * M)) % GM_NMAX);
  const long long bi = e / ((long long)M * M * GM_NMAX);
  const int i = (int)(bi % F);
  const long long b = bi / F;
  double v = 0.0;
  if (n < N) {
    const c128 *Hn = H + ((b * N + n) * F + i) * (M * M);
    if (slot < M) {
      v = Hn[slot * M + slot].x;
    } else {
      // slot M + 2 t (+ 1): upper entry t in row-major order
      const int t = (slot - M) >> 1;
      int a = 0, rem = t;
      while (rem >= M - 1 - a) {
        rem -= M - 1 - a;
        ++a;
      }
      const int c = a + 1 + rem;
      const c128 hu = Hn[a * M + c], hd = Hn[c * M + a];
      v = ((slot - M) & 1) ? hu.y - hd.y : hu.x + hd.x;
    }
  }
  Hq[e] = v;
}

// raise the block's flag when any of its points left the fast route (every thread calls this)
__device__ __forceinline__ void flag_block(bool ok, int *__restrict__ flags, int block) {
  const int bad = __syncthreads_or(ok ? 0 : 1);
  if (threadIdx.x == 0) flags[block] = bad;
}

__device__ __forceinline__ int flat_block() {
  return (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
}

// A, Bt as k_gmnmf_traces, from the packed symmetric parts Hq (k_gm_pack_spatial).
// NP = 4 or 8: sources compiled in (the padding sources of Hq are zero).  grid: (ceil(T/128), F, B)
template <int M, int NP>
__global__ __launch_bounds__(128) void k_gmnmf_traces_p(const c128 *__restrict__ X,
                                                        const double *__restrict__ basis,
                                                        const double *__restrict__ act,
                                                        const double *__restrict__ Hq,
                                                        double *__restrict__ A,
                                                        double *__restrict__ Bt, int N, int F,
                                                        int T, int K, int floor_kind, double eps,
                                                        int *__restrict__ flags) {
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = j < T;
  const int jc = live ? j : T - 1;
  const double *Hb = Hq + ((long long)b * F + i) * (GM_NMAX * M * M);
  PointP<M> pt;
  point_setup_q<M>(pt, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Hb,
                   basis + ((long long)b * N * F + i) * K, (long long)F * K, N, F, T, K, i, jc,
                   floor_kind, eps);
  double s = 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m) s += cabs2(pt.x[m]);
  double c1, c0;
  xx_floor_coeffs(s, floor_kind, eps, c1, c0);
  double accA[NP], accB[NP];
#pragma unroll
  for (int n = 0; n < NP; ++n) accA[n] = accB[n] = 0.0;
  // entry by entry: E = c1 u u^H + c0 R^-2 and G = R^-1, both Hermitian; the entry is formed once
  // and met with the sources' spatial entries (scalar operands)
#pragma unroll
  for (int a = 0; a < M; ++a) {
    const double r2 = c0 != 0.0 ? hp_square_entry<M>(pt.Rinv, a, a).x : 0.0;
    const double e = fma(c1, cabs2(pt.u[a]), c0 * r2), g = pt.Rinv.d[a];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const double h = Hb[n * (M * M) + a];
      accA[n] = fma(h, e, accA[n]);
      accB[n] = fma(h, g, accB[n]);
    }
  }
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = a + 1; c < M; ++c) {
      c128 r2 = cmake(0.0, 0.0);
      if (c0 != 0.0) r2 = hp_square_entry<M>(pt.Rinv, a, c);
      const c128 uu = cmulc(pt.u[a], pt.u[c]);  // u_a conj(u_c)
      const double ex = fma(c1, uu.x, c0 * r2.x), ey = fma(c1, uu.y, c0 * r2.y);
      const c128 g = pt.Rinv.o[tri<M>(a, c)];
      const int slot = M + 2 * tri<M>(a, c);
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        const double hs = Hb[n * (M * M) + slot], hm = Hb[n * (M * M) + slot + 1];
        accA[n] = fma(ex, hs, accA[n]);
        accA[n] = fma(ey, hm, accA[n]);
        accB[n] = fma(g.x, hs, accB[n]);
        accB[n] = fma(g.y, hm, accB[n]);
      }
    }
  if (live) {
#pragma unroll
    for (int n = 0; n < NP; ++n)
      if (n < N) {
        const long long o = (((long long)b * N + n) * F + i) * T + j;
        A[o] = accA[n];
        Bt[o] = accB[n];
      }
  }
  flag_block(pt.ok || !live, flags, flat_block());
}

// loss slots as k_gmnmf_loss.  grid: (ceil(T/128), F, B)
template <int M>
__global__ __launch_bounds__(128) void k_gmnmf_loss_p(const c128 *__restrict__ X,
                                                      const double *__restrict__ basis,
                                                      const double *__restrict__ act,
                                                      const c128 *__restrict__ H, double *out,
                                                      int N, int F, int T, int K, int floor_kind,
                                                      double eps, int *__restrict__ flags) {
  __shared__ double red[2];
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = j < T;
  const int jc = live ? j : T - 1;
  PointP<M> pt;
  point_setup_p<M>(pt, X + (long long)b * M * F * T, act + (long long)b * N * K * T,
                   H + ((long long)b * N * F + i) * (M * M), (long long)F * (M * M),
                   basis + ((long long)b * N * F + i) * K, (long long)F * K, N, F, T, K, i, jc,
                   floor_kind, eps);
  double s = 0.0, xu = 0.0, trR = 0.0;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    s += cabs2(pt.x[m]);
    xu = fma(pt.x[m].x, pt.u[m].x, xu);
    xu = fma(pt.x[m].y, pt.u[m].y, xu);
    trR += pt.Rinv.d[m];
  }
  double c1, c0;
  xx_floor_coeffs(s, floor_kind, eps, c1, c0);
  const double term = live ? fma(c1, xu, c0 * trR) + pt.logdet : 0.0;
  const double total = block_sum(term, red);
  if (threadIdx.x == 0)
    out[((long long)blockIdx.y * gridDim.x + blockIdx.x) * gridDim.z + b] = total / (double)T;
  flag_block(pt.ok || !live, flags, flat_block());
}

// Y as k_gmnmf_separate; a point that leaves the fast route stores NaN in its first source, which
// the full-storage kernel (launched next with `only_marked`) recomputes.  grid: (ceil(T/128), F, B)
template <int M>
__global__ __launch_bounds__(128) void k_gmnmf_separate_p(const c128 *__restrict__ X,
                                                          const double *__restrict__ basis,
                                                          const double *__restrict__ act,
                                                          const c128 *__restrict__ H,
                                                          c128 *__restrict__ Y, int N, int F,
                                                          int T, int K, int ref, int floor_kind,
                                                          double eps) {
  const int i = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= T) return;
  const long long hstride = (long long)F * (M * M);
  const c128 *Hb = H + ((long long)b * N * F + i) * (M * M);
  PointP<M> pt;
  point_setup_p<M>(pt, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Hb, hstride,
                   basis + ((long long)b * N * F + i) * K, (long long)F * K, N, F, T, K, i, j,
                   floor_kind, eps);
  const double nan = __builtin_nan("");
#pragma unroll
  for (int n = 0; n < GM_NMAX; ++n)
    if (n < N) {
      c128 y = cmake(0.0, 0.0);
#pragma unroll
      for (int c = 0; c < M; ++c) cfma(y, Hb[n * hstride + ref * M + c], pt.u[c]);
      y = cscale(y, pt.lam[n]);
      if (n == 0 && !pt.ok) y = cmake(nan, nan);
      Y[(((long long)b * N + n) * F + i) * T + j] = y;
    }
}

// PQacc as k_gmnmf_spatial_acc.  grid: (F, B), one wave: lanes take frames; per chunk of 64 frames
// the packed matrices of the points go through LDS and are folded with the N weights in register
// tiles: thread (eg, ng) owns the value pair 2 eg, 2 eg + 1 for NPT sources, so a point costs it
// one pair read + NPT weight reads for 2 NPT FMAs (the slot-per-thread fold of the full-storage
// kernel reads two values per FMA, and its 70 KB of LDS at 8 channels leave half the SIMDs idle).
// From GM_SPLIT_FROM channels on the two matrices of a point (R^-1, then R^-1 XX R^-1) take turns in
// the rows: 37 KB per workgroup at 8 channels (one workgroup per SIMD), 13 KB at 4.
constexpr int GM_SPLIT_FROM = 4;  // channels from which the two matrices take turns in the rows
constexpr int gm_pow2_floor(int v) { return v >= 8 ? 8 : (v >= 4 ? 4 : (v >= 2 ? 2 : 1)); }

template <int EW>  // value slots per row (even); the N weights follow at EW .. EW + 7
struct GmFold {
  static constexpr int EG = EW / 2;
  static constexpr int NG = gm_pow2_floor(64 / EG);
  static constexpr int NPT = GM_NMAX / NG;
  static constexpr int ROW = EW + GM_NMAX + 1;  // odd: rows of neighbouring lanes on distinct banks
};

template <int EW>
__device__ __forceinline__ void gm_fold_chunk(const double *pts,
                                              double (&acc)[GmFold<EW>::NPT][2]) {
  using S = GmFold<EW>;
  const int eg = threadIdx.x % S::EG, ng = threadIdx.x / S::EG;
  if (ng >= S::NG) return;
  const double *v = pts + 2 * eg, *l = pts + EW + ng * S::NPT;
  for (int p = 0; p < GM_PB; ++p) {
    const double v0 = v[p * S::ROW], v1 = v[p * S::ROW + 1];
#pragma unroll
    for (int q = 0; q < S::NPT; ++q) {
      const double lq = l[p * S::ROW + q];
      acc[q][0] = fma(lq, v0, acc[q][0]);
      acc[q][1] = fma(lq, v1, acc[q][1]);
    }
  }
}

template <int EW>
__device__ __forceinline__ void gm_fold_store(const double (&acc)[GmFold<EW>::NPT][2],
                                              double *__restrict__ dst, int N, int limit,
                                              long long nstride) {
  using S = GmFold<EW>;
  const int eg = threadIdx.x % S::EG, ng = threadIdx.x / S::EG;
  if (ng >= S::NG) return;
#pragma unroll
  for (int q = 0; q < S::NPT; ++q) {
    const int n = ng * S::NPT + q;
    if (n < N) {
      if (2 * eg < limit) dst[n * nstride + 2 * eg] = acc[q][0];
      if (2 * eg + 1 < limit) dst[n * nstride + 2 * eg + 1] = acc[q][1];
    }
  }
}

template <int M>
__global__ __launch_bounds__(GM_PB) void k_gmnmf_spatial_acc_p(const c128 *__restrict__ X,
                                                             const double *__restrict__ basis,
                                                             const double *__restrict__ act,
                                                             const double *__restrict__ Hq,
                                                             double *__restrict__ PQacc, int N,
                                                             int F, int T, int K, int floor_kind,
                                                             double eps, int *__restrict__ flags) {
  constexpr bool SPLIT = M >= GM_SPLIT_FROM;
  constexpr int MM2 = M * M;
  constexpr int EW = SPLIT ? ((MM2 + 1) & ~1) : 2 * MM2;  // value slots per row
  using S = GmFold<EW>;
  constexpr int ROW = S::ROW;
  constexpr int PASSES = SPLIT ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *pts = reinterpret_cast<double *>(smem);  // [GM_PB][ROW]
  const int i = blockIdx.x, b = blockIdx.y;
  const c128 *Xb = X + (long long)b * M * F * T;
  const double *act_b = act + (long long)b * N * K * T;
  const double *Hb = Hq + ((long long)b * F + i) * (GM_NMAX * M * M);
  const double *Tb = basis + ((long long)b * N * F + i) * K;
  double acc[PASSES][S::NPT][2];
#pragma unroll
  for (int h = 0; h < PASSES; ++h)
#pragma unroll
    for (int q = 0; q < S::NPT; ++q) acc[h][q][0] = acc[h][q][1] = 0.0;
  bool all_ok = true;
  double *mine = pts + threadIdx.x * ROW;
  for (int j0 = 0; j0 < T; j0 += GM_PB) {
    const int j = j0 + threadIdx.x;
    const bool live = j < T;
    PointP<M> pt;
    point_setup_q<M>(pt, Xb, act_b, Hb, Tb, (long long)F * K, N, F, T, K, i, live ? j : T - 1,
                     floor_kind, eps);
    all_ok = all_ok && (pt.ok || !live);
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) s += cabs2(pt.x[m]);
    double c1, c0;
    xx_floor_coeffs(s, floor_kind, eps, c1, c0);
    // packed R^-1 (diagonal first, then re / im of the upper triangle, as pack_hermitian), weights
#pragma unroll
    for (int a = 0; a < M; ++a) mine[a] = pt.Rinv.d[a];
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) {
      mine[M + 2 * e] = pt.Rinv.o[e].x;
      mine[M + 2 * e + 1] = pt.Rinv.o[e].y;
    }
    if (SPLIT && (MM2 & 1)) mine[MM2] = 0.0;
#pragma unroll
    for (int n = 0; n < GM_NMAX; ++n) mine[EW + n] = live ? pt.lam[n] : 0.0;
    constexpr int QOFF = SPLIT ? 0 : MM2;  // where packed Q = c1 u u^H + c0 R^-2 goes
    if (SPLIT) {
      __syncthreads();
      gm_fold_chunk<EW>(pts, acc[0]);
      __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < M; ++a) {
      const double r2 = c0 != 0.0 ? hp_square_entry<M>(pt.Rinv, a, a).x : 0.0;
      mine[QOFF + a] = fma(c1, cabs2(pt.u[a]), c0 * r2);
    }
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int c = a + 1; c < M; ++c) {
        const int e = M + 2 * tri<M>(a, c);
        c128 r2 = cmake(0.0, 0.0);
        if (c0 != 0.0) r2 = hp_square_entry<M>(pt.Rinv, a, c);
        const c128 uu = cmulc(pt.u[a], pt.u[c]);
        mine[QOFF + e] = fma(c1, uu.x, c0 * r2.x);
        mine[QOFF + e + 1] = fma(c1, uu.y, c0 * r2.y);
      }
    __syncthreads();
    gm_fold_chunk<EW>(pts, acc[PASSES - 1]);
    __syncthreads();
  }
  double *dst = PQacc + ((long long)b * N * F + i) * (2 * MM2);
  const long long nstride = (long long)F * (2 * MM2);
  if (SPLIT) {
    gm_fold_store<EW>(acc[0], dst, N, MM2, nstride);
    gm_fold_store<EW>(acc[PASSES - 1], dst + MM2, N, MM2, nstride);
  } else {
    gm_fold_store<EW>(acc[0], dst, N, 2 * MM2, nstride);
  }
  flag_block(all_ok, flags, blockIdx.y * gridDim.x + blockIdx.x);
}

// H <- to_psd(P^-1 # to_psd(H Q H)) as k_gmnmf_spatial_update, on its fast route: while no eigenvalue
// floor acts (every to_psd is the identity, or "+ eps I" for the add floor) the geometric mean
//   P^-1 # B = V (V^-1 B V^-H)^(1/2) V^H   for ANY factor P^-1 = V V^H
// (G P G = B has one positive definite solution), and with P = U^H U (Cholesky), V = U^-1 this needs
// ONE eigen-decomposition -- of U B U^H -- where the literal route takes four, on packed matrices
// that fit the register file.  "No floor acts" is checked, not assumed: B, P and the result must
// have every eigenvalue above eps (sufficient bounds: 1 / ||A^-1||_F > eps; for P the cheaper
// ||U^-1||_F^2 < 1 / eps); a block of 64 matrices with any failure stores nothing, raises its flag
// and is redone by the literal kernel.  One lane per (b, n, i); the block's H matrices are staged
// in LDS ([entry][lane], coalesced both ways) and the region then parks U^-1 during the sweeps.
constexpr int GSU_LD = 65;  // lanes per staged entry + 1: the transposing copies hit distinct banks
// c128 entries of the region per lane: the M x M matrix, or (two turns) the rows of the first turn
// followed by the parked S0 (M^2 doubles)
template <int M>
constexpr int gsu_park_at() {
  return M >= 7 ? 4 * M : 0;
}
template <int M>
constexpr int gsu_entries() {
  constexpr int need = (M + 1) / 2 + (M * (M - 1)) / 2;
  return M >= 7 && gsu_park_at<M>() + need > M * M ? gsu_park_at<M>() + need : M * M;
}

// One turn of the eigenvector accumulation of k_gmnmf_spatial_update_p: rows TURN * NRH ... of
// W = V J (V = U^-1 from the packed P at `pq`, formed here; J diagonalises S0) into the region,
// w = sqrt(max(eigenvalues, 0)).  S0 is consumed; with two turns the first parks it behind its rows
// and the second picks it up there.
// IDENT: the rows start from the identity instead (W = J: the eigenvectors themselves, for the
// eigenvalue floor of to_psd) and `ev` returns the raw eigenvalues.
template <int M, int TURN, bool IDENT>
__device__ __forceinline__ void gsu_turn(HermP<M> &S0, double (&w)[M], c128 *park,
                                         const double *__restrict__ pq, int floor_kind,
                                         double eps) {
  constexpr int NRH = M >= 7 ? 4 : M;  // rows per turn
  constexpr int TURNS = (M + NRH - 1) / NRH;
  constexpr int AT = gsu_park_at<M>();
  if (TURNS > 1 && TURN == 0) {
#pragma unroll
    for (int a = 0; a + 1 < M; a += 2) park[(AT + a / 2) * GSU_LD] = cmake(S0.d[a], S0.d[a + 1]);
    if (M & 1) park[(AT + M / 2) * GSU_LD] = cmake(S0.d[M - 1], 0.0);
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) park[(AT + (M + 1) / 2 + e) * GSU_LD] = S0.o[e];
  }
  if (TURN > 0) {
#pragma unroll
    for (int a = 0; a + 1 < M; a += 2) {
      const c128 z = park[(AT + a / 2) * GSU_LD];
      S0.d[a] = z.x;
      S0.d[a + 1] = z.y;
    }
    if (M & 1) S0.d[M - 1] = park[(AT + M / 2) * GSU_LD].x;
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) S0.o[e] = park[(AT + (M + 1) / 2 + e) * GSU_LD];
  }
  c128 W[NRH][M];
  if (IDENT) {
#pragma unroll
    for (int r = 0; r < NRH; ++r)
#pragma unroll
      for (int c = 0; c < M; ++c) W[r][c] = cmake(TURN * NRH + r == c ? 1.0 : 0.0, 0.0);
  } else {
    HermP<M> Vm;
#pragma unroll
    for (int a = 0; a < M; ++a) Vm.d[a] = pq[a] + (floor_kind == SSSPY_FLOOR_ADD ? eps : 0.0);
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) Vm.o[e] = cmake(pq[M + 2 * e], pq[M + 2 * e + 1]);
    double dinv[M], ld;
    hp_chol_upper<M>(Vm, dinv, ld);
    hp_trtri_upper<M>(Vm, dinv);
#pragma unroll
    for (int r = 0; r < NRH; ++r) {
      constexpr int row0 = TURN * NRH;
#pragma unroll
      for (int c = 0; c < M; ++c) {
        const int row = row0 + r;
        W[r][c] = (row < M && c > row) ? Vm.o[tri<M>(row < M - 1 ? row : 0, c > row ? c : row + 1)]
                                       : cmake(0.0, 0.0);
        if (row < M && c == row) W[r][c] = cmake(Vm.d[row], 0.0);
      }
    }
  }
  hp_jacobi_rows<M, NRH>(S0, W);
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = IDENT ? S0.d[k] : sqrt(fmax(S0.d[k], 0.0));
#pragma unroll
  for (int r = 0; r < NRH; ++r) {
    const int row = TURN * NRH + r;
    if (row < M) {
#pragma unroll
      for (int c = 0; c < M; ++c) park[(row * M + c) * GSU_LD] = W[r][c];
    }
  }
}

// Out = W diag(w) W^H from the rows the turns left in the region
template <int M>
__device__ __forceinline__ void gsu_rebuild(const c128 *park, const double (&w)[M], HermP<M> &Out) {
#pragma unroll
  for (int a = 0; a < M; ++a) {
    c128 ra[M];  // w_k W_ak
#pragma unroll
    for (int k = 0; k < M; ++k) ra[k] = cscale(park[(a * M + k) * GSU_LD], w[k]);
    double dd = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const c128 wk = park[(a * M + k) * GSU_LD];
      dd = fma(ra[k].x, wk.x, dd);
      dd = fma(ra[k].y, wk.y, dd);
    }
    Out.d[a] = dd;
#pragma unroll
    for (int c = a + 1; c < M; ++c) {
      c128 s2 = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) {  // ra[k] conj(W_ck)
        const c128 wc = park[(c * M + k) * GSU_LD];
        s2.x = fma(ra[k].x, wc.x, s2.x);
        s2.x = fma(ra[k].y, wc.y, s2.x);
        s2.y = fma(ra[k].y, wc.x, s2.y);
        s2.y = fma(-ra[k].x, wc.y, s2.y);
      }
      Out.o[tri<M>(a, c)] = s2;
    }
  }
}

// A <- to_psd(A) for the max floor: nothing when every eigenvalue is provably above eps in every
// lane of the wave (1 / ||A^-1||_F > eps); else the whole wave eigen-decomposes, floors and rebuilds
// (as the spatial matrices converge towards rank one this becomes the common case: from iteration
// 40 of an 8-channel run every block has such matrices)
template <int M>
__device__ __forceinline__ void gsu_floor_max(HermP<M> &A, c128 *park, double eps) {
  bool above;
  {
    HermP<M> tmp = A;
    double ld;
    above = hp_chol_inverse<M>(tmp, ld);
    above = above && (hp_fro2<M>(tmp) * eps * eps < 1.0);
  }
  if (__all(above)) return;
  double ev[M];
  gsu_turn<M, 0, true>(A, ev, park, nullptr, SSSPY_FLOOR_MAX, eps);
  if constexpr (M >= 7) gsu_turn<M, 1, true>(A, ev, park, nullptr, SSSPY_FLOOR_MAX, eps);
#pragma unroll
  for (int k = 0; k < M; ++k) ev[k] = fmax(ev[k], eps);
  gsu_rebuild<M>(park, ev, A);
}

template <int M>
__global__ __launch_bounds__(64) void k_gmnmf_spatial_update_p(c128 *H,
                                                               const double *__restrict__ PQacc,
                                                               long long count, int floor_kind,
                                                               double eps, int *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Hs = reinterpret_cast<c128 *>(smem);  // [gsu_entries<M>()][GSU_LD]
  const int t = threadIdx.x;
  const long long idx0 = (long long)blockIdx.x * 64, idx = idx0 + t;
  const bool live = idx < count;
  const long long idc = live ? idx : count - 1;
  for (int g = t; g < 64 * M * M; g += 64) {
    const int tl = g / (M * M), e = g % (M * M);
    const long long src = idx0 * (M * M) + g;
    Hs[e * GSU_LD + tl] = src < count * (M * M) ? H[src] : cmake(0.0, 0.0);
  }
  __syncthreads();
  const c128 *Hme = Hs + (live ? t : (int)(idc - idx0));  // entry e at Hme[e * GSU_LD]
  bool ok = true;
  double ld;
  HermP<M> Bm;
  {  // B = ((H Q) H + its adjoint) / 2
    HermP<M> Qm;
    const double *src = PQacc + idc * (2 * M * M) + M * M;
#pragma unroll
    for (int a = 0; a < M; ++a) Qm.d[a] = src[a];
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) Qm.o[e] = cmake(src[M + 2 * e], src[M + 2 * e + 1]);
    hp_clear<M>(Bm);
#pragma unroll
    for (int a = 0; a < M; ++a) {
      c128 ta[M];  // row a of H Q
#pragma unroll
      for (int k = 0; k < M; ++k) ta[k] = cmake(0.0, 0.0);
#pragma unroll
      for (int l = 0; l < M; ++l) {
        const c128 hal = Hme[(a * M + l) * GSU_LD];
#pragma unroll
        for (int k = 0; k < M; ++k) cfma(ta[k], hal, hp_get<M>(Qm, l, k));
      }
      c128 za[M];  // row a of (H Q) H
#pragma unroll
      for (int c = 0; c < M; ++c) za[c] = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k)
#pragma unroll
        for (int c = 0; c < M; ++c) cfma(za[c], ta[k], Hme[(k * M + c) * GSU_LD]);
#pragma unroll
      for (int c = 0; c < M; ++c) {
        if (c == a) {
          Bm.d[a] += za[c].x;
        } else if (c > a) {
          Bm.o[tri<M>(a, c)].x = fma(0.5, za[c].x, Bm.o[tri<M>(a, c)].x);
          Bm.o[tri<M>(a, c)].y = fma(0.5, za[c].y, Bm.o[tri<M>(a, c)].y);
        } else {
          Bm.o[tri<M>(c, a)].x = fma(0.5, za[c].x, Bm.o[tri<M>(c, a)].x);
          Bm.o[tri<M>(c, a)].y = fma(-0.5, za[c].y, Bm.o[tri<M>(c, a)].y);
        }
      }
    }
    if (floor_kind == SSSPY_FLOOR_ADD) {
#pragma unroll
      for (int a = 0; a < M; ++a) Bm.d[a] += eps;
    }
  }
  __syncthreads();  // every lane is done with the staged H: the region now serves the turns
  c128 *park = Hs + t;  // entry e at park[e * GSU_LD]
  if (floor_kind == SSSPY_FLOOR_MAX) gsu_floor_max<M>(Bm, park, eps);
  HermP<M> S0;
  {  // P = U^H U; S0 = U B U^H; V = U^-1 parked in LDS
    HermP<M> Um;
    const double *src = PQacc + idc * (2 * M * M);
#pragma unroll
    for (int a = 0; a < M; ++a) Um.d[a] = src[a] + (floor_kind == SSSPY_FLOOR_ADD ? eps : 0.0);
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) Um.o[e] = cmake(src[M + 2 * e], src[M + 2 * e + 1]);
    double dinv[M];
    ok = hp_chol_upper<M>(Um, dinv, ld) && ok;
    hp_congruence_upper<M>(Um, Bm, S0);
    hp_trtri_upper<M>(Um, dinv);
    if (floor_kind == SSSPY_FLOOR_MAX) ok = ok && (hp_fro2_upper<M>(Um) * eps < 1.0);
  }
  // W = V J, J the eigenvectors of S0 (S0 = J diag(lam) J^H): G = V S0^(1/2) V^H = W diag(sqrt lam) W^H.
  // From 7 channels on W (128 doubles) does not fit beside the working copy of S0: its rows go in
  // two turns of 4, each through the whole rotation sequence; S0 waits in the second half of the
  // region meanwhile, V is formed again for the second turn (one Cholesky + triangular inverse).
  double w[M];
  gsu_turn<M, 0, false>(S0, w, park, PQacc + idc * (2 * M * M), floor_kind, eps);
  if constexpr (M >= 7)
    gsu_turn<M, 1, false>(S0, w, park, PQacc + idc * (2 * M * M), floor_kind, eps);
  HermP<M> Gm;
  gsu_rebuild<M>(park, w, Gm);
  if (floor_kind == SSSPY_FLOOR_ADD) {
#pragma unroll
    for (int a = 0; a < M; ++a) Gm.d[a] += eps;
  }
  if (floor_kind == SSSPY_FLOOR_MAX) gsu_floor_max<M>(Gm, park, eps);
  const int bad = __syncthreads_or((ok || !live) ? 0 : 1);
  if (t == 0) flags[blockIdx.x] = bad;
  if (bad) return;
  // coalesced store through the staging region
  c128 *mine = Hs + t;
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int c = 0; c < M; ++c) mine[(a * M + c) * GSU_LD] = hp_get<M>(Gm, a, c);
  __syncthreads();
  for (int g = t; g < 64 * M * M; g += 64) {
    const int tl = g / (M * M), e = g % (M * M);
    const long long dst = idx0 * (M * M) + g;
    if (dst < count * (M * M)) H[dst] = Hs[e * GSU_LD + tl];
  }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct GmnmfWs {
  size_t a, bt, pq, vacc, teff, vrep, raw, vslabs, flags, hq, total;
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
  w.flags = off;  // one int per block of the per-point kernels: left the fast route (packed path)
  off += align256((size_t)((T + 127) / 128) * F * B * sizeof(int));
  w.hq = off;  // packed symmetric parts of the spatial matrices, [b][i][8][M M] (packed path)
  off += align256((size_t)B * F * GM_NMAX * M * M * sizeof(double));
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
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "GaussMNMF: n_basis must be in [1, 65536]");
  if (M < 2 || M > 8) return fail(SSSPY_ERR_UNSUPPORTED, "GaussMNMF: n_channels must be in [2, 8]");
  return SSSPY_OK;
}

// SSSPY_AMD_GMNMF_FULL=1: the full-storage kernels only (A / B, debugging)
// (2 and 3 channels keep the full-storage kernels: nothing spills there and the packed route's
// extra launches -- packing, the flag-gated repair kernels -- cost 10 % of a 0.15-0.25 ms iteration)
static bool packed_points(int M) { return M >= 4; }

static int launch_pack_spatial(const void *H, double *Hq, int B, int N, int M, int F,
                               hipStream_t st) {
  const long long count = (long long)B * F * GM_NMAX * M * M;
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gm_pack_spatial<MM>),
                                      dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                                      (const c128 *)H, Hq, N, F, count));
  return check_launch("k_gm_pack_spatial");
}

// flags, Hq: workspace of the packed path (both or neither)
static int launch_traces(const void *X, const double *basis, const double *act, const void *H,
                         double *A, double *Bt, int B, int N, int M, int F, int T, int K,
                         int floor_kind, double eps, int *flags, double *Hq, bool *hq_valid,
                         hipStream_t st) {
  dim3 grid((T + 127) / 128, F, B), block(128);
  const bool packed = packed_points(M) && flags != nullptr && Hq != nullptr;
  if (packed) {
    int rc = SSSPY_OK;
    if (!*hq_valid) rc = launch_pack_spatial(H, Hq, B, N, M, F, st);
    if (rc) return rc;
    *hq_valid = true;
    if (N <= 4) {
      GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_traces_p<MM, 4>), grid, block, 0, st,
                                          (const c128 *)X, basis, act, (const double *)Hq, A, Bt,
                                          N, F, T, K, floor_kind, eps, flags));
    } else {
      GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_traces_p<MM, 8>), grid, block, 0, st,
                                          (const c128 *)X, basis, act, (const double *)Hq, A, Bt,
                                          N, F, T, K, floor_kind, eps, flags));
    }
    rc = check_launch("k_gmnmf_traces_p");
    if (rc) return rc;
  }
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_traces<MM>), grid, block, bin_smem(N, M, K), st,
                                      (const c128 *)X, basis, act, (const c128 *)H, A, Bt, N, F, T,
                                      K, floor_kind, eps, packed ? (const int *)flags : nullptr));
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
  int *flags = (int *)(ws + w.flags);
  double *Hq = (double *)(ws + w.hq);
  bool hq_valid = false;  // Hq holds the packed form of the CURRENT spatial matrices (this call only)
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
    r = launch_traces(X, Tn, Vn, spatial, A, Bt, B, N, M, F, T, K, floor_kind, floor_eps, flags, Hq, &hq_valid, st);
    if (r) return r;
    {
      const int bpw = gmb_bins_per_wave(B, N, F);
      hipLaunchKernelGGL(k_gmnmf_basis, dim3((F + 4 * bpw - 1) / (4 * bpw), N, B), dim3(256), 0, st,
                       basis, Vn,
                       (const double *)A, (const double *)Bt, N, F, T, K, floor_kind, floor_eps,
                       raw_out, bpw);
    }
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
    rc = launch_traces(X, Tn, Vn, spatial, A, Bt, B, N, M, F, T, K, floor_kind, floor_eps, flags, Hq, &hq_valid, st);
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
      const bool packed = packed_points(M);
      if (packed) {
        const int ew_p = MM >= GM_SPLIT_FROM ? ((MM * MM + 1) & ~1) : 2 * MM * MM;
        const size_t smem_p = (size_t)GM_PB * (ew_p + GM_NMAX + 1) * sizeof(double);
        if (smem_p > 48 * 1024) {
          hipError_t e = hipFuncSetAttribute((const void *)k_gmnmf_spatial_acc_p<MM>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)smem_p);
          if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
        }
        if (!hq_valid) rc = launch_pack_spatial(spatial, Hq, B, N, M, F, st);
        if (rc) return rc;
        hq_valid = true;
        hipLaunchKernelGGL((k_gmnmf_spatial_acc_p<MM>), dim3(F, B), dim3(GM_PB), smem_p, st,
                           (const c128 *)X, Tn, Vn, (const double *)Hq, PQ, N, F, T, K,
                           floor_kind, floor_eps, flags);
      }
      hipLaunchKernelGGL((k_gmnmf_spatial_acc<MM>), dim3(F, B), dim3(GM_PB), smem, st,
                         (const c128 *)X, Tn, Vn, (const c128 *)spatial, PQ, N, F, T, K,
                         floor_kind, floor_eps, packed ? (const int *)flags : nullptr);
    });
    rc = check_launch("k_gmnmf_spatial_acc");
    if (rc) return rc;
    const long long count = (long long)B * N * F;
    const bool packed_su = packed_points(M);
    if (packed_su && gmnmf_spatial_update_rows_wanted(M)) {
      // 7 / 8 channels: a matrix on 8 lanes (gmnmf_rows.hip); same flags, same repair kernel below
      rc = gmnmf_spatial_update_rows(spatial, (const double *)PQ, count, M, floor_kind, floor_eps,
                                     flags, st);
      if (rc) return rc;
    } else if (packed_su) {
      // (4-6 channels; the 7 / 8-channel instantiations -- 3 511 spilled VGPRs at 8 -- went with
      //  round 5's 8-lane kernel)
      switch (M) {
#define SSSPY_GSU_P(MM_)                                                                         \
  case MM_: {                                                                                    \
    const size_t smem_su = (size_t)gsu_entries<MM_>() * GSU_LD * sizeof(c128);                   \
    if (smem_su > 48 * 1024) {                                                                   \
      hipError_t e = hipFuncSetAttribute((const void *)k_gmnmf_spatial_update_p<MM_>,            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                         (int)smem_su);                                          \
      if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));                     \
    }                                                                                            \
    hipLaunchKernelGGL((k_gmnmf_spatial_update_p<MM_>), dim3((unsigned)((count + 63) / 64)),     \
                       dim3(64), smem_su, st, (c128 *)spatial, (const double *)PQ, count,        \
                       floor_kind, floor_eps, flags);                                            \
  } break;
        SSSPY_GSU_P(4)
        SSSPY_GSU_P(5)
        SSSPY_GSU_P(6)
#undef SSSPY_GSU_P
        default: return fail(SSSPY_ERR_INTERNAL, "GaussMNMF: packed spatial update off its range");
      }
      rc = check_launch("k_gmnmf_spatial_update_p");
      if (rc) return rc;
    }
    GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_spatial_update<MM>),
                                        dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st,
                                        (c128 *)spatial, (const double *)PQ, count, floor_kind,
                                        floor_eps, packed_su ? (const int *)flags : nullptr));
    rc = check_launch("k_gmnmf_spatial_update");
    if (rc) return rc;
    hq_valid = false;
  }
  if (steps & SSSPY_GMNMF_NORMALIZE) {
    const long long count = (long long)B * N * F;
    hipLaunchKernelGGL(k_gmnmf_normalize, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st,
                       (c128 *)spatial, part ? (double *)nullptr : basis, count, M, K);
    rc = check_launch("k_gmnmf_normalize");
    if (rc) return rc;
    hq_valid = false;
  }
  if (steps & SSSPY_GMNMF_LATENT) {
    // (the latent variables of all sources sit in the LDS of one workgroup)
    if (K > SSSPY_MAX_PARTITION_BASIS)
      return fail(SSSPY_ERR_UNSUPPORTED, "GaussMNMF: partitioning takes n_basis up to 1024");
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
  // the loss slots, then one flag per block (packed path)
  return align256(scalar_slots_bytes(B, ((T + 127) / 128) * F)) +
         align256((size_t)((T + 127) / 128) * F * B * sizeof(int));
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
  int *flags = (int *)((char *)workspace + align256(scalar_slots_bytes(B, (int)grid.x * F)));
  const bool packed = packed_points(M);
  if (packed) {
    GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_loss_p<MM>), grid, block, 0, st, (const c128 *)X,
                                        basis, activation, (const c128 *)spatial,
                                        (double *)workspace, N, F, T, K, floor_kind, floor_eps,
                                        flags));
    rc = check_launch("k_gmnmf_loss_p");
    if (rc) return rc;
  }
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_loss<MM>), grid, block, bin_smem(N, M, K), st,
                                      (const c128 *)X, basis, activation, (const c128 *)spatial,
                                      (double *)workspace, N, F, T, K, floor_kind, floor_eps,
                                      packed ? (const int *)flags : nullptr));
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
  const bool packed = packed_points(M);
  if (packed) {
    GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_separate_p<MM>), grid, block, 0,
                                        as_stream(stream), (const c128 *)X, basis, activation,
                                        (const c128 *)spatial, (c128 *)Y, N, F, T, K,
                                        reference_id, floor_kind, floor_eps));
    rc = check_launch("k_gmnmf_separate_p");
    if (rc) return rc;
  }
  GM_DISPATCH_M(M, hipLaunchKernelGGL((k_gmnmf_separate<MM>), grid, block, bin_smem(N, M, K),
                                      as_stream(stream), (const c128 *)X, basis, activation,
                                      (const c128 *)spatial, (c128 *)Y, N, F, T, K, reference_id,
                                      floor_kind, floor_eps, packed ? 1 : 0));
  return check_launch("k_gmnmf_separate");
}

}  // extern "C"
