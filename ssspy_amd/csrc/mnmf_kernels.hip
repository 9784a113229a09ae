// FastGaussMNMF (jointly diagonalisable full-rank spatial model) kernels for gfx950.
//
// State (reference layout, leading batch axis B): X (B,M,F,T) c128, diagonaliser Q (B,F,M,M) c128,
// diagonal spatial D (B,F,N,M) f64, basis T (B,N,F,K), activation V (B,N,K,T).
// With lambda_nij = (T V)_nij, R~_ijm = sum_n lambda_nij d_inm and y~_imj = |(Q_i x_ij)_m| the
// four multiplicative updates + the IP1 update of Q all reduce, per (bin, frame), to a handful of
// fp64 operations on N + M numbers followed by a contraction over frames or over bins -- the same
// shape as the ILRMA passes, so the same two MFMA tile orientations are used (ilrma_kernels.hip):
//   bin-major tile   : basis update, diagonaliser covariance, spatial update, loss
//   frame-major tile : activation update
// Compiled once per N (-DSSSPY_N=2..4); M (channels) is a template parameter dispatched at launch.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "cov_core.hpp"
#include "hermitian.hpp"
#include "nmf_tile.hpp"
#include "smallmat.hpp"
#include "tail_plan.hpp"

#ifndef SSSPY_N
#error "compile with -DSSSPY_N=<n_sources>"
#endif

#define SSSPY_CAT_(a, b) a##b
#define SSSPY_CAT(a, b) SSSPY_CAT_(a, b)
#define LAUNCHER(name) SSSPY_CAT(SSSPY_CAT(name, _n), SSSPY_N)

namespace ssspy {
namespace SSSPY_CAT(mnmf_n, SSSPY_N) {

constexpr int N = SSSPY_N;

struct Dims {
  int B, F, T, K;
};

// per-(bin, frame) quantities: qx2[m] = |(Q x)_m|^2, g[m] = 1 / R~_m, R~_m = sum_n lam[n] D[n][m]
template <int M>
__device__ __forceinline__ void frame_terms(const c128 (&Q)[M][M], const double (&D)[N][M],
                                            const c128 (&x)[M], const double (&lam)[N],
                                            double (&qx2)[M], double (&rc)[M]) {
#pragma unroll
  for (int m = 0; m < M; ++m) {
    c128 y = cmake(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < M; ++a) cfma(y, Q[m][a], x[a]);
    qx2[m] = cabs2(y);
    double r = 0.0;
#pragma unroll
    for (int n = 0; n < N; ++n) r = fma(lam[n], D[n][m], r);
    rc[m] = r;
  }
}

// R~_m alone (|(Q x)_m|^2 comes from the handed-over tensor, see k_mnmf_binmajor_fast)
template <int M>
__device__ __forceinline__ void rc_terms(const double (&D)[N][M], const double (&lam)[N],
                                         double (&rc)[M]) {
#pragma unroll
  for (int m = 0; m < M; ++m) {
    double r = 0.0;
#pragma unroll
    for (int n = 0; n < N; ++n) r = fma(lam[n], D[n][m], r);
    rc[m] = r;
  }
}

template <int M>
__device__ __forceinline__ void load_bin(c128 (&Q)[M][M], double (&D)[N][M],
                                         const c128 *__restrict__ Qp,
                                         const double *__restrict__ Dp) {
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int a = 0; a < M; ++a) Q[m][a] = Qp[m * M + a];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) D[n][m] = Dp[n * M + m];
}

// ================================================================================ basis update
// t_nik <- floor(t_nik sqrt( sum_j v_nkj A_nij / sum_j v_nkj Bq_nij )),
// A_nij = sum_m d_inm y~^2/R~^2, Bq_nij = sum_m d_inm / R~.   grid: (bin tiles, k tiles, B)
template <int M, bool KSMALL>
__global__ __launch_bounds__(256) void k_mnmf_basis(const c128 *__restrict__ X,
                                                    const c128 *__restrict__ Q,
                                                    const double *__restrict__ Dsp,
                                                    const double *basis, double *basis_out,
                                                    const double *__restrict__ act, Dims d,
                                                    int floor_kind, double eps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z, kt = blockIdx.y;
  const int i0 = blockIdx.x * 16;
  const int bin = min(i0 + c, F - 1);
  c128 Qb[M][M];
  double Db[N][M];
  load_bin<M>(Qb, Db, Q + ((long long)b * F + bin) * (M * M), Dsp + ((long long)b * F + bin) * (N * M));
  double tb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      tb[n][ks] = (KSMALL && kk < K) ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  double4_t num[N], den[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    num[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    den[n] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
  const int ntiles = (T + 15) >> 4;
  const int k2 = kt * 16 + c;
  const bool k2valid = k2 < K;
  const int k2c = k2valid ? k2 : K - 1;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
    double4_t lamR[N];
#pragma unroll
    for (int n = 0; n < N; ++n)
      lamR[n] = nmf_rt_tile<KSMALL>(act + ((long long)b * N + n) * K * T,
                                    basis + (((long long)b * N + n) * F + bin) * K, tb[n], K, T, j0,
                                    c, q);
    double a[N][4], bq[N][4];
    bool fval[4];
    int jcl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j0 + 4 * q + r;
      fval[r] = jj < T;
      jcl[r] = fval[r] ? jj : T - 1;
      c128 x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + bin) * T + jcl[r]];
      double lam[N], qx2[M], rc[M];
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      frame_terms<M>(Qb, Db, x, lam, qx2, rc);
      double g[M], h[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        g[m] = 1.0 / rc[m];
        h[m] = qx2[m] * g[m] * g[m];
      }
#pragma unroll
      for (int n = 0; n < N; ++n) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          sa = fma(Db[n][m], h[m], sa);
          sb = fma(Db[n][m], g[m], sb);
        }
        a[n][r] = fval[r] ? sa : 0.0;
        bq[n][r] = fval[r] ? sb : 0.0;
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *Vn = act + ((long long)b * N + n) * K * T;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double vb = Vn[(long long)k2c * T + jcl[r]];
        vb = (k2valid && fval[r]) ? vb : 0.0;
        num[n] = mfma_f64(a[n][r], vb, num[n]);
        den[n] = mfma_f64(bq[n][r], vb, den[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lds[((wave * N + n) * 2 + 0) * 256 + r * 64 + lane] = num[n][r];
      lds[((wave * N + n) * 2 + 1) * 256 + r * 64 + lane] = den[n][r];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < N * 256; e += blockDim.x) {
    const int n = e >> 8, r = (e >> 6) & 3, ln = e & 63;
    double sn = 0.0, sd = 0.0;
    for (int wv = 0; wv < nw; ++wv) {
      sn += lds[((wv * N + n) * 2 + 0) * 256 + r * 64 + ln];
      sd += lds[((wv * N + n) * 2 + 1) * 256 + r * 64 + ln];
    }
    const int ob = i0 + (ln >> 4) + 4 * r, ok = kt * 16 + (ln & 15);
    if (ob < F && ok < K) {
      const long long o = (((long long)b * N + n) * F + ob) * K + ok;
      basis_out[o] = apply_floor(basis[o] * sqrt(sn / sd), floor_kind, eps);
    }
  }
}

// =========================================================================== activation update
// grid: (frame groups, bin chunks, B * ktiles); partials part[b][chunk][n][nd][K][T]
template <int M, bool KSMALL>
__global__ __launch_bounds__(256) void k_mnmf_activation(const c128 *__restrict__ X,
                                                         const c128 *__restrict__ Q,
                                                         const double *__restrict__ Dsp,
                                                         const double *__restrict__ basis,
                                                         const double *__restrict__ act,
                                                         double *__restrict__ part, Dims d,
                                                         int ktiles, int tiles_per_chunk,
                                                         int nchunks) {
  __shared__ c128 ql[16 * M * M];
  __shared__ double dl[16 * N * M];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int kt = blockIdx.z % ktiles, b = blockIdx.z / ktiles;
  const int chunk = blockIdx.y;
  const int j0 = (blockIdx.x * nw + wave) * 16;
  const int jf = j0 + c;
  const bool fvalid = jf < T;
  const int jc = fvalid ? jf : T - 1;
  double vb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      vb[n][ks] = (KSMALL && kk < K && fvalid) ? act[(((long long)b * N + n) * K + kk) * T + jc] : 0.0;
    }
  double4_t numv[N], denv[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    numv[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    denv[n] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
  const int ntiles = (F + 15) >> 4;
  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = min(ntiles, t_begin + tiles_per_chunk);
  const int k2 = kt * 16 + c;
  const bool k2valid = k2 < K;
  const int k2c = k2valid ? k2 : K - 1;
  for (int it = t_begin; it < t_end; ++it) {
    const int i0 = it * 16;
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * M * M; e += blockDim.x) {
      const int bi = min(i0 + e / (M * M), F - 1);
      ql[e] = Q[((long long)b * F + bi) * (M * M) + e % (M * M)];
    }
    for (int e = threadIdx.x; e < 16 * N * M; e += blockDim.x) {
      const int bi = min(i0 + e / (N * M), F - 1);
      dl[e] = Dsp[((long long)b * F + bi) * (N * M) + e % (N * M)];
    }
    __syncthreads();
    double4_t lamR[N];
    const int ab = min(i0 + c, F - 1);
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *Tn = basis + ((long long)b * N + n) * F * K;
      double4_t R = {0.0, 0.0, 0.0, 0.0};
      if (KSMALL) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks * 4 < K) {
            const int kk = ks * 4 + q;
            double ta = Tn[(long long)ab * K + (kk < K ? kk : K - 1)];
            ta = kk < K ? ta : 0.0;
            R = mfma_f64(ta, vb[n][ks], R);
          }
        }
      } else {
        for (int k0 = 0; k0 < K; k0 += 4) {
          const int kk = k0 + q, kc = kk < K ? kk : K - 1;
          double ta = Tn[(long long)ab * K + kc];
          ta = kk < K ? ta : 0.0;
          double v = act[(((long long)b * N + n) * K + kc) * T + jc];
          v = (kk < K && fvalid) ? v : 0.0;
          R = mfma_f64(ta, v, R);
        }
      }
      lamR[n] = R;
    }
    double a[N][4], bq[N][4];
    bool bval[4];
    int bcl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bl = q + 4 * r;
      bval[r] = i0 + bl < F;
      bcl[r] = bval[r] ? i0 + bl : F - 1;
      c128 x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + bcl[r]) * T + jc];
      c128 Qb[M][M];
      double Db[N][M];
      load_bin<M>(Qb, Db, ql + bl * M * M, dl + bl * N * M);
      double lam[N], qx2[M], rc[M];
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      frame_terms<M>(Qb, Db, x, lam, qx2, rc);
      double g[M], h[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        g[m] = 1.0 / rc[m];
        h[m] = qx2[m] * g[m] * g[m];
      }
      const bool valid = bval[r] && fvalid;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          sa = fma(Db[n][m], h[m], sa);
          sb = fma(Db[n][m], g[m], sb);
        }
        a[n][r] = valid ? sa : 0.0;
        bq[n][r] = valid ? sb : 0.0;
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *Tn = basis + ((long long)b * N + n) * F * K;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double ta = Tn[(long long)bcl[r] * K + k2c];
        ta = (k2valid && bval[r]) ? ta : 0.0;
        numv[n] = mfma_f64(ta, a[n][r], numv[n]);
        denv[n] = mfma_f64(ta, bq[n][r], denv[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ok = kt * 16 + q + 4 * r;
      if (ok < K && fvalid) {
        const long long base = ((((long long)b * nchunks + chunk) * N + n) * 2) * K;
        part[(base + ok) * T + jf] = numv[n][r];
        part[(base + K + ok) * T + jf] = denv[n][r];
      }
    }
}

// ============================================================ diagonaliser covariance (for IP1)
// U[b,i,m] = (1/T) sum_j x x^H / R~_ijm    -> (B,F,M,M,M).   grid: (bin tiles, 1, B)
template <int M, bool KSMALL>
__global__ __launch_bounds__(256) void k_mnmf_wcov(const c128 *__restrict__ X,
                                                   const double *__restrict__ Dsp,
                                                   const double *__restrict__ basis,
                                                   const double *__restrict__ act,
                                                   c128 *__restrict__ U, Dims d) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z;
  const int i0 = blockIdx.x * 16;
  const int bin = min(i0 + c, F - 1);
  double Db[N][M];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) Db[n][m] = Dsp[((long long)b * F + bin) * (N * M) + n * M + m];
  double tb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      tb[n][ks] = (KSMALL && kk < K) ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  CovAcc<M, M> acc;
  acc.clear();
  const int ntiles = (T + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
    double4_t lamR[N];
#pragma unroll
    for (int n = 0; n < N; ++n)
      lamR[n] = nmf_rt_tile<KSMALL>(act + ((long long)b * N + n) * K * T,
                                    basis + (((long long)b * N + n) * F + bin) * K, tb[n], K, T, j0,
                                    c, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j0 + 4 * q + r;
      const bool valid = jj < T;
      const int jc = valid ? jj : T - 1;
      c128 x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + bin) * T + jc];
      double phi[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double rr = 0.0;
#pragma unroll
        for (int n = 0; n < N; ++n) rr = fma(lamR[n][r], Db[n][m], rr);
        phi[m] = valid ? 1.0 / rr : 0.0;
      }
      acc.add(x, phi);
    }
  }
  acc.fold_q();
  cov_reduce_store<M, M>(acc, lds, U, (long long)b * F, i0, F, M, 0, M, 1.0 / (double)T);
}

// ================================================================================ spatial update
// d_inm <- d_inm sqrt( sum_j lam y~^2 / R~^2  /  sum_j lam / R~ )  (no floor).  grid: (bin tiles,1,B)
template <int M, bool KSMALL>
__global__ __launch_bounds__(256) void k_mnmf_spatial(const c128 *__restrict__ X,
                                                      const c128 *__restrict__ Q, double *Dsp,
                                                      const double *__restrict__ basis,
                                                      const double *__restrict__ act, Dims d) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z;
  const int i0 = blockIdx.x * 16;
  const int bin = min(i0 + c, F - 1);
  c128 Qb[M][M];
  double Db[N][M];
  load_bin<M>(Qb, Db, Q + ((long long)b * F + bin) * (M * M), Dsp + ((long long)b * F + bin) * (N * M));
  double tb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      tb[n][ks] = (KSMALL && kk < K) ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  double sn[N][M], sd[N][M];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) sn[n][m] = sd[n][m] = 0.0;
  const int ntiles = (T + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
    double4_t lamR[N];
#pragma unroll
    for (int n = 0; n < N; ++n)
      lamR[n] = nmf_rt_tile<KSMALL>(act + ((long long)b * N + n) * K * T,
                                    basis + (((long long)b * N + n) * F + bin) * K, tb[n], K, T, j0,
                                    c, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j0 + 4 * q + r;
      const bool valid = jj < T;
      const int jc = valid ? jj : T - 1;
      c128 x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + bin) * T + jc];
      double lam[N], qx2[M], rc[M];
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      frame_terms<M>(Qb, Db, x, lam, qx2, rc);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const double g = 1.0 / rc[m];
        const double h = qx2[m] * g * g;
#pragma unroll
        for (int n = 0; n < N; ++n) {
          sn[n][m] += valid ? lam[n] * h : 0.0;
          sd[n][m] += valid ? lam[n] * g : 0.0;
        }
      }
    }
  }
  // fold q lanes, then waves (LDS: [wave][nd][n*M+m][16 bins])
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double v = sn[n][m];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      double w = sd[n][m];
      w += __shfl_xor(w, 16, 64);
      w += __shfl_xor(w, 32, 64);
      if (q == 0) {
        lds[((wave * 2 + 0) * (N * M) + n * M + m) * 16 + c] = v;
        lds[((wave * 2 + 1) * (N * M) + n * M + m) * 16 + c] = w;
      }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * N * M; e += blockDim.x) {
    const int cb = e & 15, nm = e >> 4;
    double a = 0.0, bb = 0.0;
    for (int wv = 0; wv < nw; ++wv) {
      a += lds[((wv * 2 + 0) * (N * M) + nm) * 16 + cb];
      bb += lds[((wv * 2 + 1) * (N * M) + nm) * 16 + cb];
    }
    const int ob = i0 + cb;
    if (ob < F) {
      double *dst = Dsp + ((long long)b * F + ob) * (N * M) + nm;
      *dst = sqrt(a / bb) * (*dst);
    }
  }
}

// ============================================================== throughput variants (K <= 16, even T)
// Same restructuring as ilrma_fast.hip: the 4 waves of a workgroup own 4 adjacent 16-bin tiles and
// walk the frame tiles together; the activation tile of every source is staged once per workgroup in
// LDS (double buffered, one barrier per tile); x of the next tile is prefetched into registers; 1/R~
// is v_rcp_f64 + 2 Newton steps; accumulators stay with the owning wave (no cross-wave fold).
constexpr int VROW = 18;

__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}

struct VStage {
  double2 v[(N * 16 * 8 + 255) / 256];
};
__device__ __forceinline__ void vstage_load(VStage &st, const double *__restrict__ act_b, int K,
                                            int T, int j0) {
#pragma unroll
  for (int u = 0; u < (N * 16 * 8 + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int row = idx >> 3, chunk = idx & 7;
    const int n = row >> 4, k = row & 15;
    const int j = j0 + 2 * chunk;
    double2 val = make_double2(0.0, 0.0);
    if (idx < N * 16 * 8 && k < K) val = load_pair_in_row(act_b + ((long long)n * K + k) * T, j, T);
    st.v[u] = val;
  }
}
__device__ __forceinline__ void vstage_store(const VStage &st, double *buf) {
#pragma unroll
  for (int u = 0; u < (N * 16 * 8 + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int row = idx >> 3, chunk = idx & 7;
    if (idx < N * 16 * 8) *reinterpret_cast<double2 *>(buf + row * VROW + 2 * chunk) = st.v[u];
  }
}
// ksteps = ceil(K / 4): k-slabs beyond n_basis are zero on both sides and are skipped
template <int KQ>
__device__ __forceinline__ double4_t rt_from_lds(const double *vs_n, const double (&tb)[KQ], int c,
                                                 int q, int ksteps) {
  double4_t R = {0.0, 0.0, 0.0, 0.0};
  const int col = tile_pi(c);
#pragma unroll
  for (int ks = 0; ks < KQ; ++ks)
    if (ks < ksteps) R = mfma_f64(vs_n[(4 * ks + q) * VROW + col], tb[ks], R);
  return R;
}
template <int M>
struct XTileM {
  c128 x[M][4];
};
// Through a buffer descriptor of the mixture's tensor: one 32-bit lane offset per tile, channel
// stride in the scalar offset, the four frames in the immediate (flat loads cost ~5 VALU
// instructions of 64-bit address arithmetic each, 80 per tile, in kernels that are issue-bound).
// Frames beyond T are not clamped: they read the next row (finite data, zeros past the end of the
// tensor) and every consumer masks them with `valid`.  Needs M * F * T * 16 < 2^32.
template <int M>
__device__ __forceinline__ void xtile_load(XTileM<M> &xt, __amdgpu_buffer_rsrc_t xr, int F, int T,
                                           int bin, int j0, int q) {
  const unsigned voff = ((unsigned)bin * (unsigned)T + (unsigned)(j0 + 4 * q)) * 16u;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const unsigned soff = (unsigned)m * (unsigned)F * (unsigned)T * 16u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, voff + 16u * r, soff, 0);
      xt.x[m][r] = cmake(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
    }
  }
}

// |(Q x)_m|^2 handed over from the spatial pass (B, M, F, T) f64: the same walk as xtile_load at
// half the bytes and without the M x M complex products.  Needs M * F * T * 8 < 2^32 and an even
// T (16-byte aligned rows; odd T keeps the x path).
template <int M>
struct PTileM {
  double p[M][4];
};
template <int M>
__device__ __forceinline__ void ptile_load(PTileM<M> &pt, __amdgpu_buffer_rsrc_t pr, int F, int T,
                                           int bin, int j0, int q) {
  const unsigned voff = ((unsigned)bin * (unsigned)T + (unsigned)(j0 + 4 * q)) * 8u;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const unsigned soff = (unsigned)m * (unsigned)F * (unsigned)T * 8u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(pr, voff + 16u * h, soff, 0);
      pt.p[m][2 * h] = __hiloint2double((int)v[1], (int)v[0]);
      pt.p[m][2 * h + 1] = __hiloint2double((int)v[3], (int)v[2]);
    }
  }
}

// which of the bin-major passes a kernel instance performs (MODE_LOSS: the data term of the
// negative log-likelihood from the |Q x|^2 hand-over, P_READ only)
enum { MODE_BASIS = 0, MODE_WCOV = 1, MODE_SPATIAL = 2, MODE_LOSS = 3 };

// sum of logs as a running product of mantissas plus a sum of exponents (v_frexp_*; one log at the
// end): 2 VALU instructions per value instead of an fp64 log (same device as ilrma_fast.hip's)
struct LogSumM {
  double mant;
  int expo;
  __device__ __forceinline__ void clear() {
    mant = 1.0;
    expo = 0;
  }
  __device__ __forceinline__ void mul(double x) {
    mant *= __builtin_amdgcn_frexp_mant(x);
    expo += __builtin_amdgcn_frexp_exp(x);
  }
  __device__ __forceinline__ void renorm() {
    expo += __builtin_amdgcn_frexp_exp(mant);
    mant = __builtin_amdgcn_frexp_mant(mant);
  }
  __device__ __forceinline__ double value() const {
    return log(mant) + 0.6931471805599453094 * (double)expo;
  }
};
// the |Q x|^2 hand-over: none, read it instead of x (basis pass), write it (spatial pass)
enum { P_NONE = 0, P_READ = 1, P_WRITE = 2 };
#ifndef SSSPY_MNMF_PBASIS_WAVES
#define SSSPY_MNMF_PBASIS_WAVES 2
#endif
constexpr int PBASIS_WAVES = SSSPY_MNMF_PBASIS_WAVES;  // waves per SIMD of the P_READ basis pass

// grid: 1-D, see TailPlan (tail_plan.hpp): a work item is (mixture, 64-bin group), wave w owns bins
// [64 group + 16 w, +16).  Unsplit items finish their bins in place; the split items of the last
// scheduling round write partial sums to `tailpart` ([tail item][chunk][...]) and a small kernel
// folds them (k_mnmf_basis_finalize / k_mnmf_wcov_fold / k_mnmf_spatial_finalize).
template <int M>
constexpr int mnmf_tail_doubles() {
  // per (tail item, chunk): basis N*64*16*2, covariance 64*M^3 complex, spatial 64*N*M*2
  constexpr int a = N * 64 * 16 * 2, b = 64 * M * M * M * 2, c = 64 * N * M * 2;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

// KQ: k-slabs of 4 the instance carries (4: n_basis <= 16; 2: n_basis <= 8, half the basis registers)
template <int M, int MODE, int PMODE = P_NONE, int KQ = 4>
__global__ __launch_bounds__(256, PMODE == P_READ ? PBASIS_WAVES : 1) void k_mnmf_binmajor_fast(
    const c128 *__restrict__ X, const c128 *__restrict__ Q, double *Dsp, double *basis,
    const double *__restrict__ act, c128 *__restrict__ U, int F, int T, int K, int floor_kind,
    double eps, TailPlan plan, double *__restrict__ tailpart, double *__restrict__ P,
    const double *__restrict__ pscale) {
  static_assert((PMODE == P_NONE && MODE != MODE_LOSS) ||
                    (PMODE == P_READ && (MODE == MODE_BASIS || MODE == MODE_LOSS)) ||
                    (PMODE == P_WRITE && MODE == MODE_SPATIAL),
                "the basis and loss passes read the hand-over, the spatial pass writes it");
  __shared__ __attribute__((aligned(16))) double vs[2][N * 16 * VROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const BlockWork work = block_work(plan);  // XCD-contiguous: a mixture's groups share its V tile
  const int b = work.b, nchunks = work.nchunks;
  const int i0 = work.group * 64 + wave * 16;
  const int bin = min(i0 + c, F - 1);
  const int ksteps = (K + 3) >> 2;
  // PMODE == P_READ: the descriptor of the mixture's |Q x|^2 slab instead of x
  const __amdgpu_buffer_rsrc_t xr =
      PMODE == P_READ
          ? make_rsrc(P + (long long)b * M * F * T, (unsigned)M * (unsigned)F * (unsigned)T * 8u)
          : make_rsrc(X + (long long)b * M * F * T, (unsigned)M * (unsigned)F * (unsigned)T * 16u);
  const double *act_b = act + (long long)b * N * K * T;
  c128 Qb[M][M];
  double Db[N][M];
  double ps[M];
#pragma unroll
  for (int m = 0; m < M; ++m) ps[m] = PMODE == P_READ ? pscale[b * M + m] : 1.0;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int a2 = 0; a2 < M; ++a2)
      Qb[m][a2] = (MODE != MODE_WCOV && PMODE != P_READ)
                      ? Q[((long long)b * F + bin) * (M * M) + m * M + a2]
                      : cmake(0.0, 0.0);  // neither the covariance pass nor the hand-over needs Q
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) Db[n][m] = Dsp[((long long)b * F + bin) * (N * M) + n * M + m];
  double tb[N][KQ];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
      const int kk = 4 * ks + q;
      tb[n][ks] = kk < K ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  // accumulators of the three modes (only the selected mode's are live)
  double4_t num[N], den[N];
  CovAcc<M, M> acc;
  double sn[N][M], sd[N][M];
  if (MODE == MODE_BASIS) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      num[n] = double4_t{0.0, 0.0, 0.0, 0.0};
      den[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
  }
  if (MODE == MODE_WCOV) acc.clear();
  if (MODE == MODE_SPATIAL) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int m = 0; m < M; ++m) sn[n][m] = sd[n][m] = 0.0;
  }
  double lacc = 0.0;  // MODE_LOSS: sum of |Q x|^2 / R~ over this lane's (bin, frame, channel)
  LogSumM lr;         // ... and of log R~
  lr.clear();
  const bool bin_valid = i0 + c < F;
  const int ntiles = (T + 15) >> 4;
  const int tpc = (ntiles + nchunks - 1) / nchunks;
  const int jt_begin = work.chunk * tpc, jt_end = min(ntiles, jt_begin + tpc);
  VStage st;
  using Tile = typename std::conditional<PMODE == P_READ, PTileM<M>, XTileM<M>>::type;
  Tile cur, nxt;
  auto load_tile = [&](Tile &t, const int j0) __attribute__((always_inline)) {
    if constexpr (PMODE == P_READ) ptile_load<M>(t, xr, F, T, bin, j0, q);
    else xtile_load<M>(t, xr, F, T, bin, j0, q);
  };
  // (round 5, tried and dropped: letting a wave whose 16 bins all lie beyond F -- three of the four
  //  waves of the 17th bin group at F = 1025 -- skip its tile loads and arithmetic.  The branch around
  //  the tile body costs the scheduler more than the 4.4 % of skipped wave tiles give back: the
  //  hand-over basis pass 219 -> 227 us, and the ILRMA basis pass 1.112 -> 1.209 ms at 128 mixtures.
  //  The LDS-DMA kernels below, whose walk is hand-scheduled, keep it.)
  vstage_load(st, act_b, K, T, min(jt_begin, ntiles - 1) * 16);
  load_tile(cur, min(jt_begin, ntiles - 1) * 16);
  vstage_store(st, vs[0]);
  __syncthreads();
  // one tile of the walk: compute on `xc`, prefetch the next tile into `xn`.  The walk calls it with
  // the two register sets swapping roles (no 64-register copy per tile).
  auto tile = [&](const Tile &xc, Tile &xn, const int jt) __attribute__((always_inline)) {
    const int j0 = jt * 16;
    const int jn = min(jt + 1, jt_end - 1) * 16;
    vstage_load(st, act_b, K, T, jn);
    load_tile(xn, jn);
    const double *vcur = vs[(jt - jt_begin) & 1];
    double4_t lamR[N];
#pragma unroll
    for (int n = 0; n < N; ++n) lamR[n] = rt_from_lds<KQ>(vcur + n * 16 * VROW, tb[n], c, q, ksteps);
    double a[N][4], bq[N][4];
    double pw[M][4];  // P_WRITE: this tile's |Q x|^2
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool valid = j0 + 4 * q + r < T;
      c128 x[M];
      double lam[N], qx2[M], rc[M];
      if constexpr (PMODE == P_READ) {
#pragma unroll
        for (int m = 0; m < M; ++m) qx2[m] = xc.p[m][r] * ps[m];
      } else {
#pragma unroll
        for (int m = 0; m < M; ++m) x[m] = xc.x[m][r];
      }
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      if (MODE == MODE_WCOV) {
        double phi[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          double rr = 0.0;
#pragma unroll
          for (int n = 0; n < N; ++n) rr = fma(lam[n], Db[n][m], rr);
          phi[m] = valid ? rcp_nr(rr) : 0.0;
        }
        acc.add(x, phi);
      } else {
        if constexpr (PMODE == P_READ) rc_terms<M>(Db, lam, rc);
        else frame_terms<M>(Qb, Db, x, lam, qx2, rc);
        if constexpr (PMODE == P_WRITE) {
#pragma unroll
          for (int m = 0; m < M; ++m) pw[m][r] = qx2[m];
        }
        double g[M], h[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          g[m] = rcp_nr(rc[m]);
          h[m] = qx2[m] * g[m] * g[m];
        }
        if (MODE == MODE_LOSS) {
          const bool lv = valid && bin_valid;
#pragma unroll
          for (int m = 0; m < M; ++m) {
            lacc += lv ? qx2[m] * g[m] : 0.0;  // (select the product: g is inf on padded frames)
            lr.mul(lv ? rc[m] : 1.0);
          }
        } else if (MODE == MODE_BASIS) {
#pragma unroll
          for (int n = 0; n < N; ++n) {
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
              sa = fma(Db[n][m], h[m], sa);
              sb = fma(Db[n][m], g[m], sb);
            }
            if constexpr (PMODE == P_READ) {
              // straight into the contraction over frames (no 32-register a / bq tile to hold)
              const double vbr = vcur[(n * 16 + c) * VROW + 4 * q + r];
              num[n] = mfma_f64(valid ? sa : 0.0, vbr, num[n]);
              den[n] = mfma_f64(valid ? sb : 0.0, vbr, den[n]);
            } else {
              a[n][r] = valid ? sa : 0.0;
              bq[n][r] = valid ? sb : 0.0;
            }
          }
        } else if (MODE == MODE_SPATIAL) {
#pragma unroll
          for (int m = 0; m < M; ++m)
#pragma unroll
            for (int n = 0; n < N; ++n) {
              sn[n][m] += valid ? lam[n] * h[m] : 0.0;
              sd[n][m] += valid ? lam[n] * g[m] : 0.0;
            }
        }
      }
    }
    if (MODE == MODE_LOSS) lr.renorm();
    if (MODE == MODE_BASIS && PMODE != P_READ) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const double *vn = vcur + n * 16 * VROW;
        const double2 v01 = *reinterpret_cast<const double2 *>(vn + c * VROW + 4 * q);
        const double2 v23 = *reinterpret_cast<const double2 *>(vn + c * VROW + 4 * q + 2);
        const double vb[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          num[n] = mfma_f64(a[n][r], vb[r], num[n]);
          den[n] = mfma_f64(bq[n][r], vb[r], den[n]);
        }
      }
    }
    if constexpr (PMODE == P_WRITE) {
      if (i0 + c < F) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          double *dst = P + (((long long)b * M + m) * F + bin) * T + j0 + 4 * q;
          if (j0 + 4 * q + 3 < T) {
            *reinterpret_cast<double2 *>(dst) = make_double2(pw[m][0], pw[m][1]);
            *reinterpret_cast<double2 *>(dst + 2) = make_double2(pw[m][2], pw[m][3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (j0 + 4 * q + r < T) dst[r] = pw[m][r];
          }
        }
      }
    }
    vstage_store(st, vs[(jt - jt_begin + 1) & 1]);
    __syncthreads();
  };
  for (int jt = jt_begin; jt < jt_end; jt += 2) {
    tile(cur, nxt, jt);
    if (jt + 1 < jt_end) tile(nxt, cur, jt + 1);
  }
  if (MODE == MODE_LOSS) {
    // `tailpart` holds the loss slots here, [slot][B] with one slot per (bin group, chunk, wave) of
    // the mixture; this mode has no floor, and the launcher passes the slots' stride (B, or the
    // caller's for raw slots) in `floor_kind`
    lacc += lr.value();
    lacc = wave_sum(lacc);
    const int maxsplit = plan.split > 1 ? plan.split : 1;
    if (lane == 0)
      tailpart[(long long)((work.group * maxsplit + work.chunk) * 4 + wave) * floor_kind + b] =
          lacc / (double)T;
    return;
  }
  const long long slot = (long long)work.tail_idx * nchunks + work.chunk;
  double *tp = tailpart + slot * mnmf_tail_doubles<M>();
  if (MODE == MODE_BASIS) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ob = i0 + q + 4 * r;
        if (ob < F && c < K) {
          if (nchunks == 1) {
            double *dst = basis + (((long long)b * N + n) * F + ob) * K + c;
            *dst = apply_floor((*dst) * sqrt(num[n][r] / den[n][r]), floor_kind, eps);
          } else {
            double *dst = tp + ((n * 64 + (ob - work.group * 64)) * 16 + c) * 2;
            dst[0] = num[n][r];
            dst[1] = den[n][r];
          }
        }
      }
  }
  if (MODE == MODE_WCOV) {
    acc.fold_q();
    const double scale = 1.0 / (double)T;
    const int ob = i0 + c;
    if (ob < F && q == 0) {
      c128 *dst = nchunks == 1 ? U + ((long long)b * F + ob) * (long long)(M * M * M)
                               : reinterpret_cast<c128 *>(tp) +
                                     (long long)(ob - work.group * 64) * (M * M * M);
#pragma unroll
      for (int s = 0; s < M; ++s) {
        int e = 0;
#pragma unroll
        for (int aa = 0; aa < M; ++aa) {
          dst[(s * M + aa) * M + aa] = cmake(acc.diag[s][aa] * scale, 0.0);
#pragma unroll
          for (int bb = aa + 1; bb < M; ++bb) {
            const c128 z = acc.off[s][e];
            dst[(s * M + aa) * M + bb] = cmake(z.x * scale, z.y * scale);
            dst[(s * M + bb) * M + aa] = cmake(z.x * scale, -z.y * scale);
            ++e;
          }
        }
      }
    }
  }
  if (MODE == MODE_SPATIAL) {
    const int ob = i0 + c;
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double v = sn[n][m], w = sd[n][m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        w += __shfl_xor(w, 16, 64);
        w += __shfl_xor(w, 32, 64);
        if (q == 0 && ob < F) {
          if (nchunks == 1) {
            double *dst = Dsp + ((long long)b * F + ob) * (N * M) + n * M + m;
            *dst = sqrt(v / w) * Db[n][m];
          } else {
            double *dst = tp + (((ob - work.group * 64) * N + n) * M + m) * 2;
            dst[0] = v;
            dst[1] = w;
          }
        }
      }
  }
}

// ============================================== x-reading passes with an LDS-DMA tile pipeline
// Round 5.  The covariance and spatial passes above hold one workgroup per CU (Q, D, the statistics
// and two x tiles in registers) and keep ONE 16 KB tile per wave in flight: 16 MB over the chip,
// which at the loaded HBM latency is ~3.5 TB/s -- what they measure.  This form takes the tiles out of
// the register file: `buffer_load_dwordx4 ... lds` writes them straight into a two-slot ring per wave
// (128 KB of the CU's 160 KB at 4 channels), the activation tiles into a three-slot ring shared by
// the workgroup (24 KB); the wave copies the landed tile into ONE register set and re-issues the slot
// at once -- two tiles in flight behind the one being computed, 64 fewer VGPRs.  hipcc does not track
// LDS-DMA per slot (it waits vmcnt(0) before any LDS read it can see while a DMA is pending, and at
// __syncthreads()), so inside the walk every LDS read is an asm ds_read, every wait a counted
// s_waitcnt, the barrier a raw s_barrier, and the number of VMEM instructions per tile is fixed: the
// |Q x|^2 stores go through a buffer descriptor with an out-of-range offset for lanes that have
// nothing to store, never through a branch.
//
// LDS image of an x tile (16 bins x 16 frames x M channels): instruction u = 4 m + quad is 1 KB in
// lane order; lane l = 4 c' + p fetches sample r = p ^ (c' >> 2 & 3) of frames j0 + 4 quad + r of bin
// c' -- four adjacent lanes cover one 64-byte run (the register form fetches 64 scattered samples per
// instruction) -- and the consumer lane (c, q) finds frame r of channel m at slot
// 4 c + (r ^ (c >> 2 & 3)) of instruction 4 m + q: each of ds_read_b128's 16-lane groups covers the
// 16 slots mod 16 once (no bank conflict; tests/test_host_logic.py replays the index maps).
// Shapes: T % 16 == 0 (whole tiles: no frame masks anywhere), n_basis <= 16, the hand-over buffer
// present (spatial pass); everything else takes k_mnmf_binmajor_fast.
#ifndef SSSPY_GLDS_DBG
#define SSSPY_GLDS_DBG 0  // experiments (benchmarks/tools/build_variant.sh): 1 no arithmetic, 2 no DMA in the walk, 3 no barrier
#endif
typedef double d2_t __attribute__((ext_vector_type(2)));
#define SSSPY_LDS_ADDR(p) ((unsigned)(size_t)((__attribute__((address_space(3))) void *)(p)))

template <int NWAIT>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NWAIT) : "memory");
}
__device__ __forceinline__ void glds_b128(__amdgpu_buffer_rsrc_t r, unsigned lds_addr, unsigned voff,
                                          unsigned soff) {
  // (no instruction offset: the hardware adds it to the LDS address as well)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      r, (__attribute__((address_space(3))) void *)(size_t)lds_addr, 16, voff, soff, 0, 0);
}
// a value the compiler must treat as used and redefined here: pins the compiler's own waits for
// ordinary loads in front of the first DMA
__device__ __forceinline__ void pin(double &a) { asm volatile("" : "+v"(a)); }

template <int M>
struct XRegs {
  d2_t x[M][4];
};
// the tile of this lane out of ring slot `base` (byte address of this lane's frame-0 sample of
// channel 0, the r-th entry already swizzled), then lgkmcnt(0) tied to every register read
template <int M>
__device__ __forceinline__ void xtile_from_lds(XRegs<M> &t, const unsigned (&base)[4]) {
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t.x[m][r]) : "v"(base[r]), "n"(m * 4096));
#pragma unroll
  for (int m = 0; m < M; ++m)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t.x[m][0]), "+v"(t.x[m][1]), "+v"(t.x[m][2]), "+v"(t.x[m][3]));
}
// GEMM1 of one source out of the activation ring: A operand V[k = 4 ks + q][frame tile_pi(c)]
template <int KQ>
__device__ __forceinline__ void vrow_from_lds(double (&a)[KQ], unsigned addr, int n, unsigned src_bytes) {
#pragma unroll
  for (int ks = 0; ks < KQ; ++ks)
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[ks]) : "v"(addr + (unsigned)n * src_bytes), "n"(ks * 512));
}

// PRIV (n_basis <= 8 only): every wave fetches its OWN copy of the activation tile (two 4 KB slots per
// wave: with the x ring the whole 160 KB of the CU) and copies it into registers with the x tile, so
// nothing is shared between the waves any more: no barrier in the walk (it cost 30 of the covariance
// pass's 250 us at 32 mixtures: every tile all four waves waited for the slowest DMA), and a wave
// without bins leaves at once.  Costs 4 KB more L2 -> LDS traffic per wave tile (16 KB of x).
// TSTORE (spatial pass, shared activation ring): the |Q x|^2 tile of a channel goes through a 2 KB
// patch per wave (the last 8 KB of the CU's LDS) so that four adjacent lanes store one 64-byte run
// (lanes are bins: without it an instruction stores 64 scattered 16-byte pieces).
template <int M, int MODE, int KQ, bool PRIV = false, bool TSTORE = false>
__global__ __launch_bounds__(256, 1) void k_mnmf_binmajor_glds(
    const c128 *__restrict__ X, const c128 *__restrict__ Q, double *Dsp,
    const double *__restrict__ basis, const double *__restrict__ act, c128 *__restrict__ U, int F,
    int T, int K, TailPlan plan, double *__restrict__ tailpart, double *__restrict__ P) {
  static_assert(MODE == MODE_WCOV || MODE == MODE_SPATIAL, "the two passes that read x");
  static_assert(KQ == 2 || KQ == 4, "k-slabs of 4 carried: n_basis <= 8 or <= 16");
  static_assert(!PRIV || KQ == 2, "private activation tiles: n_basis <= 8");
  static_assert(!TSTORE || (MODE == MODE_SPATIAL && !PRIV), "transposed stores: spatial pass, shared ring");
  constexpr int XI = 4 * M;                                 // DMA instructions per x tile and wave
  constexpr int NV = PRIV ? N : KQ / 2;                     // ... per activation tile and wave
  constexpr int NS = MODE == MODE_SPATIAL ? 2 * M : 0;      // stores per tile and wave
  constexpr unsigned XSLOT = 4u * XI * 1024u;               // bytes of one x ring slot (4 waves)
  // bytes of one activation ring slot (PRIV: rows 0..7 of every source, per wave)
  constexpr unsigned VSLOT = PRIV ? (unsigned)N * 8u * 16u * 8u : (unsigned)N * 16u * 16u * 8u;
  constexpr unsigned VSRC = PRIV ? 1024u : 2048u;           // bytes between two sources in a slot
  __shared__ __attribute__((aligned(16))) double xring[2 * 4 * XI * 128];
  __shared__ __attribute__((aligned(16))) double vring[PRIV ? 4 * 2 * N * 8 * 16 : 3 * N * 16 * 16];
  __shared__ __attribute__((aligned(16))) double ppatch[TSTORE ? 4 * 256 : 2];  // 2 KB per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, q = lane >> 4;
  const BlockWork work = block_work(plan);
  const int b = work.b, nchunks = work.nchunks;
  const int i0 = work.group * 64 + wave * 16;
  const int bin = min(i0 + c, F - 1);
  const bool bin_valid = i0 + c < F;
  const __amdgpu_buffer_rsrc_t xr =
      make_rsrc(X + (long long)b * M * F * T, (unsigned)M * (unsigned)F * (unsigned)T * 16u);
  const __amdgpu_buffer_rsrc_t vr =
      make_rsrc(act + (long long)b * N * K * T, (unsigned)N * (unsigned)K * (unsigned)T * 8u);
  c128 Qb[M][M];
  double Db[N][M];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int a2 = 0; a2 < M; ++a2)
      Qb[m][a2] = MODE == MODE_SPATIAL ? Q[((long long)b * F + bin) * (M * M) + m * M + a2]
                                       : cmake(0.0, 0.0);
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int m = 0; m < M; ++m) Db[n][m] = Dsp[((long long)b * F + bin) * (N * M) + n * M + m];
  double tb[N][KQ];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
      const int kk = 4 * ks + q;
      tb[n][ks] = kk < K ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  // every ordinary load has landed before the first DMA is issued (see the header)
  if (MODE == MODE_SPATIAL) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int a2 = 0; a2 < M; ++a2) {
        pin(Qb[m][a2].x);
        pin(Qb[m][a2].y);
      }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int m = 0; m < M; ++m) pin(Db[n][m]);
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) pin(tb[n][ks]);
  }
  wait_vm<0>();

  CovAcc<M, M> acc;
  double sn[N][M], sd[N][M];
  if (MODE == MODE_WCOV) acc.clear();
  if (MODE == MODE_SPATIAL) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int m = 0; m < M; ++m) sn[n][m] = sd[n][m] = 0.0;
  }
  const int ntiles = (T + 15) >> 4;
  const int tpc = (ntiles + nchunks - 1) / nchunks;
  const int jt_begin = work.chunk * tpc, jt_end = min(ntiles, jt_begin + tpc);

  // ---- producer side: which sample of a tile this lane fetches
  const int cl = lane >> 2, rl = (lane & 3) ^ ((cl >> 2) & 3);
  const unsigned xvoff = ((unsigned)min(i0 + cl, F - 1) * (unsigned)T + (unsigned)rl) * 16u;
  const unsigned xlds = SSSPY_LDS_ADDR(xring) + (unsigned)wave * (XI * 1024u);
  // shared ring: wave w fetches the rows of source w % N (instruction h: rows 8 h .. 8 h + 7);
  // PRIV: every wave fetches rows 0 .. 7 of every source (instruction h = source) into its own slots
  const int nv = wave % N;
  const unsigned vlds = SSSPY_LDS_ADDR(vring) +
                        (PRIV ? (unsigned)wave * 2u * VSLOT : (unsigned)nv * 2048u);
  unsigned vvoff[NV];
#pragma unroll
  for (int h = 0; h < NV; ++h) {
    const int k = min((PRIV ? 0 : 8 * h) + (lane >> 3), K - 1);
    vvoff[h] = (((unsigned)(PRIV ? h : nv) * (unsigned)K + (unsigned)k) * (unsigned)T +
                2u * (lane & 7)) * 8u;
  }
  const unsigned chan = (unsigned)F * (unsigned)T * 16u;
  // a wave whose 16 bins all lie beyond F (F = 1025: three of the four waves of every mixture's 17th
  // bin group) fetches no x, computes and stores nothing -- it only brings its share of the
  // activation tiles and keeps the barriers (4.4 % of the wave tiles of these shapes)
  const bool active = i0 < F;
  if (PRIV && !active) return;  // (nothing shared: no barrier to keep)
  auto issue = [&](const int jt, const int xslot, const int vslot) __attribute__((always_inline)) {
    const unsigned j0 = (unsigned)jt * 16u;
    if (active) {
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          glds_b128(xr, xlds + (unsigned)xslot * XSLOT + (unsigned)(4 * m + qd) * 1024u,
                    xvoff + j0 * 16u, (unsigned)m * chan + (unsigned)qd * 64u);
    }
#pragma unroll
    for (int h = 0; h < NV; ++h)
      glds_b128(vr, vlds + (unsigned)vslot * VSLOT + (unsigned)h * 1024u, vvoff[h] + j0 * 8u, 0u);
  };
  // ---- consumer side
  unsigned xrd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    xrd[r] = xlds + (unsigned)q * 1024u + (unsigned)c * 64u + 16u * (unsigned)(r ^ ((c >> 2) & 3));
  const unsigned vrd = (PRIV ? vlds : SSSPY_LDS_ADDR(vring)) + (unsigned)(q * 16 + tile_pi(c)) * 8u;
  // |Q x|^2 out: lane (c, q) owns frames j0 + 4 q .. + 3 of its bin, 32 bytes per channel
  const __amdgpu_buffer_rsrc_t pr =
      make_rsrc(MODE == MODE_SPATIAL ? P + (long long)b * M * F * T : nullptr,
                MODE == MODE_SPATIAL ? (unsigned)M * (unsigned)F * (unsigned)T * 8u : 0u);
  const unsigned pvoff = ((unsigned)bin * (unsigned)T + 4u * (unsigned)q) * 8u;
  const unsigned pchan = (unsigned)F * (unsigned)T * 8u;

  if (jt_begin < jt_end) issue(jt_begin, 0, 0);
  if (jt_begin + 1 < jt_end) issue(jt_begin + 1, 1, 1);
  int vslot = 0;  // (t - jt_begin) % 3
  for (int t = jt_begin; t < jt_end; ++t) {
    const int i = t - jt_begin;
    const int j0 = t * 16;
    const bool more = t + 2 < jt_end;
    // x(t) and V(t) have landed; what was issued after them may still be in flight: the DMAs of
    // tile t + 1 and the stores of tiles t - 2 and t - 1 (the first two tiles have fewer stores
    // behind them, the last two no younger DMAs: vmcnt is an upper bound on what may be pending)
    if (!more || !active) wait_vm<0>();
    else if (i >= 2) wait_vm<XI + NV + 2 * NS>();
    else if (i == 1) wait_vm<XI + NV + NS>();
    else wait_vm<XI + NV>();
    XRegs<M> cur;
    if (active) {
      unsigned base[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) base[r] = xrd[r] + (unsigned)(i & 1) * XSLOT;
      xtile_from_lds<M>(cur, base);
    }
    double va[N][KQ];
    if constexpr (PRIV) {
      // the wave's own activation tile goes to registers with the x tile; both slots are free again
      const unsigned vmine = vrd + (unsigned)(i & 1) * VSLOT;
#pragma unroll
      for (int n = 0; n < N; ++n) vrow_from_lds<KQ>(va[n], vmine, n, VSRC);
      // (landed in registers before the slot is handed to the next DMA)
#pragma unroll
      for (int n = 0; n < N; ++n)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[n][0]), "+v"(va[n][1]));
    } else {
      // every wave's share of V(t) is in LDS, and every wave is done with V(t - 1)
#if SSSPY_GLDS_DBG != 3
      __builtin_amdgcn_s_barrier();
#endif
    }
#if SSSPY_GLDS_DBG != 2
    if (more) issue(t + 2, i & 1, PRIV ? (i & 1) : (vslot == 0 ? 2 : vslot - 1));
#endif
    const unsigned vcur = vrd + (unsigned)vslot * VSLOT;
    vslot = vslot == 2 ? 0 : vslot + 1;
    if (!active) continue;
    double4_t lamR[N];
    {
      if constexpr (!PRIV) {
#pragma unroll
        for (int n = 0; n < N; ++n) vrow_from_lds<KQ>(va[n], vcur, n, VSRC);
      }
      // (lgkmcnt(0): scalar loads share the counter and return out of order)
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (KQ == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[n][0]), "+v"(va[n][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(va[n][0]), "+v"(va[n][1]), "+v"(va[n][KQ - 2]), "+v"(va[n][KQ - 1]));
        double4_t R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) R = mfma_f64(va[n][ks], tb[n][ks], R);
        lamR[n] = R;
      }
    }
    double pw[M][4];
#if SSSPY_GLDS_DBG == 1
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) pw[m][r] = cur.x[m][r].x + lamR[m % N][r];
    if (MODE == MODE_WCOV) acc.diag[0][0] += pw[0][0] + pw[M - 1][3];
#else
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c128 x[M];
      double lam[N];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = cmake(cur.x[m][r].x, cur.x[m][r].y);
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      if (MODE == MODE_WCOV) {
        double phi[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          double rr = 0.0;
#pragma unroll
          for (int n = 0; n < N; ++n) rr = fma(lam[n], Db[n][m], rr);
          phi[m] = rcp_nr(rr);
        }
        acc.add(x, phi);
      } else {
        double qx2[M], rc[M];
        frame_terms<M>(Qb, Db, x, lam, qx2, rc);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          pw[m][r] = qx2[m];
          const double g = rcp_nr(rc[m]);
          const double h = qx2[m] * g * g;
#pragma unroll
          for (int n = 0; n < N; ++n) {
            sn[n][m] = fma(lam[n], h, sn[n][m]);
            sd[n][m] = fma(lam[n], g, sd[n][m]);
          }
        }
      }
    }
#endif
    if constexpr (MODE == MODE_SPATIAL && TSTORE) {
      // channel by channel through the wave's patch: lane (c, q) writes its 32 bytes (logical 16-byte
      // slots 2 q, 2 q + 1 of bin row c at physical slot ^ (c & 7): conflict-free for the 8-lane write
      // groups), lane 4 c' + p reads slot 4 s + p of bin row c' back (frames 8 s + 2 p, + 1) and stores
      // it: LDS operations of a wave complete in order, so the only wait is before the stores
      const unsigned pbase = SSSPY_LDS_ADDR(ppatch) + (unsigned)wave * 2048u;
      const unsigned wr0 = pbase + (unsigned)c * 128u + 16u * (unsigned)((2 * q) ^ (c & 7));
      const unsigned wr1 = pbase + (unsigned)c * 128u + 16u * (unsigned)((2 * q + 1) ^ (c & 7));
      const int cs = lane >> 2, ps = lane & 3;
      const unsigned rd0 = pbase + (unsigned)cs * 128u + 16u * (unsigned)(ps ^ (cs & 7));
      const unsigned rd1 = pbase + (unsigned)cs * 128u + 16u * (unsigned)((4 + ps) ^ (cs & 7));
      const bool svalid = i0 + cs < F;
      const unsigned soff = svalid ? ((unsigned)min(i0 + cs, F - 1) * (unsigned)T + (unsigned)j0 +
                                      2u * (unsigned)ps) * 8u
                                   : 0x80000000u;
      d2_t got[M][2];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const d2_t lo = {pw[m][0], pw[m][1]}, hi = {pw[m][2], pw[m][3]};
        asm volatile("ds_write_b128 %0, %1" : : "v"(wr0), "v"(lo) : "memory");
        asm volatile("ds_write_b128 %0, %1" : : "v"(wr1), "v"(hi) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(got[m][0]) : "v"(rd0) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(got[m][1]) : "v"(rd1) : "memory");
      }
#pragma unroll
      for (int m = 0; m < M; ++m)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(got[m][0]), "+v"(got[m][1]));
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          u32x4_t v;
          v[0] = (unsigned)__double2loint(got[m][sx].x);
          v[1] = (unsigned)__double2hiint(got[m][sx].x);
          v[2] = (unsigned)__double2loint(got[m][sx].y);
          v[3] = (unsigned)__double2hiint(got[m][sx].y);
          __builtin_amdgcn_raw_buffer_store_b128(v, pr, soff + (unsigned)m * pchan + 64u * sx, 0, 0);
        }
    } else if (MODE == MODE_SPATIAL) {
      // (2 M stores whatever the lane holds: see the header)
      const unsigned off = bin_valid ? pvoff + (unsigned)j0 * 8u : 0x80000000u;
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4_t v;
          v[0] = (unsigned)__double2loint(pw[m][2 * h]);
          v[1] = (unsigned)__double2hiint(pw[m][2 * h]);
          v[2] = (unsigned)__double2loint(pw[m][2 * h + 1]);
          v[3] = (unsigned)__double2hiint(pw[m][2 * h + 1]);
          __builtin_amdgcn_raw_buffer_store_b128(v, pr, off + (unsigned)m * pchan + 16u * h, 0, 0);
        }
    }
  }
  const long long slot = (long long)work.tail_idx * nchunks + work.chunk;
  double *tp = tailpart + slot * mnmf_tail_doubles<M>();
  if (MODE == MODE_WCOV) {
    acc.fold_q();
    const double scale = 1.0 / (double)T;
    const int ob = i0 + c;
    if (ob < F && q == 0) {
      c128 *dst = nchunks == 1 ? U + ((long long)b * F + ob) * (long long)(M * M * M)
                               : reinterpret_cast<c128 *>(tp) +
                                     (long long)(ob - work.group * 64) * (M * M * M);
#pragma unroll
      for (int s = 0; s < M; ++s) {
        int e = 0;
#pragma unroll
        for (int aa = 0; aa < M; ++aa) {
          dst[(s * M + aa) * M + aa] = cmake(acc.diag[s][aa] * scale, 0.0);
#pragma unroll
          for (int bb = aa + 1; bb < M; ++bb) {
            const c128 z = acc.off[s][e];
            dst[(s * M + aa) * M + bb] = cmake(z.x * scale, z.y * scale);
            dst[(s * M + bb) * M + aa] = cmake(z.x * scale, -z.y * scale);
            ++e;
          }
        }
      }
    }
  }
  if (MODE == MODE_SPATIAL) {
    const int ob = i0 + c;
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        double v = sn[n][m], w = sd[n][m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        w += __shfl_xor(w, 16, 64);
        w += __shfl_xor(w, 32, 64);
        if (q == 0 && ob < F) {
          if (nchunks == 1) {
            double *dst = Dsp + ((long long)b * F + ob) * (N * M) + n * M + m;
            *dst = sqrt(v / w) * Db[n][m];
          } else {
            double *dst = tp + (((ob - work.group * 64) * N + n) * M + m) * 2;
            dst[0] = v;
            dst[1] = w;
          }
        }
      }
  }
}

// ---- folds of the split (tail) items; grid.y = tail item, one thread per output value
// basis <- floor(basis * sqrt(sum num / sum den)); grid: (N*64*16/256, tail)
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_basis_finalize(double *basis,
                                                             const double *__restrict__ tailpart,
                                                             int F, int K, TailPlan plan,
                                                             int floor_kind, double eps) {
  const int tail_idx = blockIdx.y;
  const int item = plan.full + tail_idx;
  const int b = item / plan.groups, group = item - b * plan.groups;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (n, local bin, k16)
  const int k = e & 15, lb = (e >> 4) & 63, n = e >> 10;
  const int bin = group * 64 + lb;
  if (n >= N || k >= K || bin >= F) return;
  // (chunks in order, eight loads per round trip: ordered_sum, common.hpp)
  double *dst = basis + (((long long)b * N + n) * F + bin) * K + k;
  const double told = *dst;
  const double2 s = ordered_sum(
      reinterpret_cast<const double2 *>(tailpart + (long long)tail_idx * plan.split *
                                                       mnmf_tail_doubles<M>()) + e,
      (long long)(mnmf_tail_doubles<M>() / 2), plan.split);
  *dst = apply_floor(told * sqrt(s.x / s.y), floor_kind, eps);
}

// U[tail items] = sum of their chunks; grid: (64*M^3/256 rounded up, tail)
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_wcov_fold(c128 *__restrict__ U,
                                                        const double *__restrict__ tailpart, int F,
                                                        TailPlan plan) {
  constexpr int PER = 64 * M * M * M;
  const int tail_idx = blockIdx.y;
  const int item = plan.full + tail_idx;
  const int b = item / plan.groups, group = item - b * plan.groups;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int lb = e / (M * M * M);
  if (e >= PER || group * 64 + lb >= F) return;
  const double2 s = ordered_sum(
      reinterpret_cast<const double2 *>(tailpart + (long long)tail_idx * plan.split *
                                                       mnmf_tail_doubles<M>()) + e,
      (long long)(mnmf_tail_doubles<M>() / 2), plan.split);
  U[((long long)b * F + group * 64) * (long long)(M * M * M) + e] = cmake(s.x, s.y);
}

// d <- d * sqrt(sum a / sum b); grid: (64*N*M/256 rounded up, tail)
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_spatial_finalize(double *Dsp,
                                                               const double *__restrict__ tailpart,
                                                               int F, TailPlan plan) {
  constexpr int PER = 64 * N * M;
  const int tail_idx = blockIdx.y;
  const int item = plan.full + tail_idx;
  const int b = item / plan.groups, group = item - b * plan.groups;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (local bin, n, m)
  const int lb = e / (N * M);
  if (e >= PER || group * 64 + lb >= F) return;
  double *dst = Dsp + ((long long)b * F + group * 64) * (N * M) + e;
  const double dold = *dst;
  const double2 s = ordered_sum(
      reinterpret_cast<const double2 *>(tailpart + (long long)tail_idx * plan.split *
                                                       mnmf_tail_doubles<M>()) + e,
      (long long)(mnmf_tail_doubles<M>() / 2), plan.split);
  *dst = sqrt(s.x / s.y) * dold;
}

// ------------------------------------------------------------ activation, frame-major fast variant
// grid: (ceil(T/64), chunks, B); wave w owns frames [64 bx + 16 w, +16) and walks the bin tiles of
// its chunk; the basis tile of every source, the 16 diagonalisers and the 16 spatial matrices of the
// tile are staged in LDS (double buffered, one barrier per bin tile).
constexpr int TROW = 17;

template <int M>
struct TStageM {
  double t[(N * 256 + 255) / 256];
  c128 qv;
  double dv;
};

template <int M, bool WITHQ = true>
__device__ __forceinline__ void tstage_load(TStageM<M> &st, const double *__restrict__ basis_b,
                                            const c128 *__restrict__ Q_b,
                                            const double *__restrict__ D_b, int F, int K, int i0) {
#pragma unroll
  for (int u = 0; u < (N * 256 + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;  // (n, bin, k)
    const int k = idx & 15, bl = (idx >> 4) & 15, n = idx >> 8;
    const int bi = i0 + bl;
    double v = 0.0;
    if (idx < N * 256 && k < K && bi < F) v = basis_b[((long long)n * F + bi) * K + k];
    st.t[u] = v;
  }
  {
    const int idx = threadIdx.x;
    const int bi = min(i0 + idx / (M * M), F - 1);
    st.qv = (WITHQ && idx < 16 * M * M) ? Q_b[(long long)bi * (M * M) + idx % (M * M)]
                                        : cmake(0.0, 0.0);
    const int bj = min(i0 + idx / (N * M), F - 1);
    st.dv = idx < 16 * N * M ? D_b[(long long)bj * (N * M) + idx % (N * M)] : 0.0;
  }
}

template <int M, bool WITHQ = true>
__device__ __forceinline__ void tstage_store(const TStageM<M> &st, double *tbuf, c128 *qbuf,
                                             double *dbuf) {
#pragma unroll
  for (int u = 0; u < (N * 256 + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int k = idx & 15, row = idx >> 4;
    if (idx < N * 256) tbuf[row * TROW + k] = st.t[u];
  }
  if (WITHQ && threadIdx.x < 16 * M * M) qbuf[threadIdx.x] = st.qv;
  if (threadIdx.x < 16 * N * M) dbuf[threadIdx.x] = st.dv;
}

// USEP: |(Q x)_m|^2 comes from the hand-over tensor P (B, M, F, T) times pscale (B, M) instead of
// x and Q (half the bytes, no M x M complex products).
template <int M, bool USEP>
__global__ __launch_bounds__(256, USEP ? 2 : 1) void k_mnmf_activation_fast(
    const c128 *__restrict__ X, const c128 *__restrict__ Q, const double *__restrict__ Dsp,
    const double *__restrict__ basis, const double *__restrict__ act, double *__restrict__ part,
    int F, int T, int K, int tiles_per_chunk, int nchunks, const double *__restrict__ P,
    const double *__restrict__ pscale) {
  __shared__ __attribute__((aligned(16))) double ts[2][N * 16 * TROW];
  __shared__ __attribute__((aligned(16))) c128 ql[2][16 * M * M];
  __shared__ __attribute__((aligned(16))) double dl[2][16 * N * M];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const GridItem gi = xcd_contiguous_grid();  // frame groups of a (mixture, chunk) share its tiles
  const int b = gi.z, chunk = gi.y;
  const int j0 = (gi.x * 4 + wave) * 16;
  const int jf = j0 + c;
  const bool fvalid = jf < T;
  const int jc = fvalid ? jf : T - 1;
  const __amdgpu_buffer_rsrc_t xsrc =
      USEP ? make_rsrc(P + (long long)b * M * F * T, (unsigned)M * (unsigned)F * (unsigned)T * 8u)
           : make_rsrc(X + (long long)b * M * F * T, (unsigned)M * (unsigned)F * (unsigned)T * 16u);
  double ps[M];
#pragma unroll
  for (int m = 0; m < M; ++m) ps[m] = USEP ? pscale[b * M + m] : 1.0;
  const double *basis_b = basis + (long long)b * N * F * K;
  const c128 *Q_b = Q + (long long)b * F * M * M;
  const double *D_b = Dsp + (long long)b * F * N * M;
  double vb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = 4 * ks + q;
      vb[n][ks] = (kk < K && fvalid) ? act[(((long long)b * N + n) * K + kk) * T + jc] : 0.0;
    }
  double4_t numv[N], denv[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    numv[n] = double4_t{0.0, 0.0, 0.0, 0.0};
    denv[n] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
  const int ntiles = (F + 15) >> 4;
  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = min(ntiles, t_begin + tiles_per_chunk);
  const int ksteps = (K + 3) >> 2;  // k-slabs beyond n_basis are zero on both sides
  TStageM<M> st;
  tstage_load<M, !USEP>(st, basis_b, Q_b, D_b, F, K, t_begin * 16);
  tstage_store<M, !USEP>(st, ts[0], ql[0], dl[0]);
  __syncthreads();
  for (int it = t_begin; it < t_end; ++it) {
    const int i0 = it * 16;
    const int in = min(it + 1, t_end - 1) * 16;
    // x through the mixture's buffer descriptor: one lane offset per tile, (channel, bin row) in the
    // scalar offset.  Bins beyond F read the next channel's rows (zeros past the tensor) and are masked.
    c128 x[M][4];
    double pv[M][4];
    if constexpr (USEP) {
      const unsigned voff = ((unsigned)(i0 + q) * (unsigned)T + (unsigned)jc) * 8u;
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned soff = ((unsigned)m * (unsigned)F + 4u * r) * (unsigned)T * 8u;
          const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(xsrc, voff, soff, 0);
          pv[m][r] = __hiloint2double((int)v[1], (int)v[0]);
        }
    } else {
      const unsigned voff = ((unsigned)(i0 + q) * (unsigned)T + (unsigned)jc) * 16u;
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned soff = ((unsigned)m * (unsigned)F + 4u * r) * (unsigned)T * 16u;
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xsrc, voff, soff, 0);
          x[m][r] =
              cmake(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
        }
    }
    tstage_load<M, !USEP>(st, basis_b, Q_b, D_b, F, K, in);
    const int pb = (it - t_begin) & 1;
    const double *tcur = ts[pb];
    double4_t lamR[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      double4_t R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        if (ks < ksteps) R = mfma_f64(tcur[(n * 16 + c) * TROW + 4 * ks + q], vb[n][ks], R);
      lamR[n] = R;
    }
    double a[N][4], bq[N][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bl = q + 4 * r;
      const bool valid = fvalid && (i0 + bl < F);
      c128 Qb[M][M];
      double Db[N][M];
      load_bin<M>(Qb, Db, ql[pb] + bl * M * M, dl[pb] + bl * N * M);
      double lam[N], qx2[M], rc[M];
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      if constexpr (USEP) {
#pragma unroll
        for (int m = 0; m < M; ++m) qx2[m] = pv[m][r] * ps[m];
        rc_terms<M>(Db, lam, rc);
      } else {
        c128 xr[M];
#pragma unroll
        for (int m = 0; m < M; ++m) xr[m] = x[m][r];
        frame_terms<M>(Qb, Db, xr, lam, qx2, rc);
      }
      double g[M], h[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        g[m] = rcp_nr(rc[m]);
        h[m] = qx2[m] * g[m] * g[m];
      }
#pragma unroll
      for (int n = 0; n < N; ++n) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          sa = fma(Db[n][m], h[m], sa);
          sb = fma(Db[n][m], g[m], sb);
        }
        a[n][r] = valid ? sa : 0.0;
        bq[n][r] = valid ? sb : 0.0;
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double ta = tcur[(n * 16 + q + 4 * r) * TROW + c];
        numv[n] = mfma_f64(ta, a[n][r], numv[n]);
        denv[n] = mfma_f64(ta, bq[n][r], denv[n]);
      }
    tstage_store<M, !USEP>(st, ts[pb ^ 1], ql[pb ^ 1], dl[pb ^ 1]);
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ok = q + 4 * r;
      if (ok < K && fvalid) {
        const long long base = ((((long long)b * nchunks + chunk) * N + n) * 2) * K;
        part[(base + ok) * T + jf] = numv[n][r];
        part[(base + K + ok) * T + jf] = denv[n][r];
      }
    }
}

// ======================================================================================= loss
// out[b] += sum_i (1/T) sum_j sum_m ( y~^2 / R~ + log R~ )
template <int M, bool KSMALL>
__global__ __launch_bounds__(256) void k_mnmf_loss(const c128 *__restrict__ X,
                                                   const c128 *__restrict__ Q,
                                                   const double *__restrict__ Dsp,
                                                   const double *__restrict__ basis,
                                                   const double *__restrict__ act, double *out,
                                                   Dims d) {
  __shared__ double scratch[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z;
  const int i0 = blockIdx.x * 16;
  const bool binvalid = i0 + c < F;
  const int bin = min(i0 + c, F - 1);
  c128 Qb[M][M];
  double Db[N][M];
  load_bin<M>(Qb, Db, Q + ((long long)b * F + bin) * (M * M), Dsp + ((long long)b * F + bin) * (N * M));
  double tb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      tb[n][ks] = (KSMALL && kk < K) ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  double local = 0.0;
  const int ntiles = (T + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
    double4_t lamR[N];
#pragma unroll
    for (int n = 0; n < N; ++n)
      lamR[n] = nmf_rt_tile<KSMALL>(act + ((long long)b * N + n) * K * T,
                                    basis + (((long long)b * N + n) * F + bin) * K, tb[n], K, T, j0,
                                    c, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j0 + 4 * q + r;
      const bool valid = binvalid && jj < T;
      const int jc = jj < T ? jj : T - 1;
      c128 x[M];
#pragma unroll
      for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + bin) * T + jc];
      double lam[N], qx2[M], rc[M];
#pragma unroll
      for (int n = 0; n < N; ++n) lam[n] = lamR[n][r];
      frame_terms<M>(Qb, Db, x, lam, qx2, rc);
      double term = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) term += qx2[m] / rc[m] + log(rc[m]);
      local += valid ? term : 0.0;
    }
  }
  const double total = block_sum(local, scratch);
  // one slot per bin tile of the mixture ([slot][B]); the launcher adds them in order
  if (threadIdx.x == 0) out[(long long)blockIdx.x * d.B + b] = total / (double)T;
}

// ================================================================ normalisation of Q rows and D
// psi_m = floor(sqrt(mean_i q[i][m])); Q[:,m,:] /= psi_m; D[:,:,m] /= psi_m^2. grid (ceil(F/64), B)
// |Q x|^2 of one mixture straight from x and Q (the hand-over tensor when nothing valid is at hand:
// first iteration, or Q / x changed by the caller) and pscale <- 1.  grid: (ceil(T/256), F, B)
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_qx2(const c128 *__restrict__ X,
                                                  const c128 *__restrict__ Q,
                                                  double *__restrict__ P, double *pscale, int F,
                                                  int T) {
  const int b = blockIdx.z, bin = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && bin == 0 && threadIdx.x < M) pscale[b * M + threadIdx.x] = 1.0;
  if (j >= T) return;
  const c128 *Qb = Q + ((long long)b * F + bin) * (M * M);
  c128 x[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + bin) * T + j];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    c128 y = cmake(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < M; ++a) cfma(y, Qb[m * M + a], x[a]);
    P[(((long long)b * M + m) * F + bin) * T + j] = cabs2(y);
  }
}

__global__ void k_mnmf_fill_ones(double *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1.0;
}

// pscale (B, M), when given, takes the same 1 / psi_m^2 as D: |(Q x)_m|^2 = P * pscale stays true.
// pscale_fresh: P was written by the spatial pass of this very call (scale 1): store 1 / psi^2
// instead of dividing -- the pass then need not reset the scale first.
// grid: (ceil(F / 64), B).  Every block folds q over all bins (fixed order), but asks for its 64
// diagonalisers and spatial rows BEFORE the reduction: one round trip instead of three.
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_norm_scale(c128 *Q, double *Dsp,
                                                         const double *__restrict__ qbuf, int F,
                                                         int floor_kind, double eps,
                                                         double *pscale, int pscale_fresh,
                                                         const double *__restrict__ tailpart,
                                                         int split) {
  __shared__ double wsum[4][M];
  __shared__ double psi[M];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * 64;
  const int nb = min(64, F - i0);
  constexpr int QPT = (64 * M * M + 255) / 256, DPT = (64 * N * M + 255) / 256;
  c128 *Qb = Q + ((long long)b * F + i0) * M * M;
  double *Db = Dsp + ((long long)b * F + i0) * N * M;
  c128 qv[QPT];
  double dv[DPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const int e = threadIdx.x + 256 * u;
    qv[u] = Qb[min(e, nb * M * M - 1)];
  }
#pragma unroll
  for (int u = 0; u < DPT; ++u) {
    const int e = threadIdx.x + 256 * u;
    dv[u] = Db[min(e, nb * N * M - 1)];
  }
  if (tailpart) {
    // the spatial pass of this call left the (a, b) sums of this bin group as `split` partial
    // records (every item split: a handful of mixtures); its fold d <- d sqrt(sum a / sum b)
    // (k_mnmf_spatial_finalize) happens here, on the way
    const double2 *rec = reinterpret_cast<const double2 *>(
        tailpart + ((long long)b * gridDim.x + blockIdx.x) * split * mnmf_tail_doubles<M>());
#pragma unroll
    for (int u = 0; u < DPT; ++u) {
      const int e = min((int)threadIdx.x + 256 * u, nb * N * M - 1);
      const double2 s = ordered_sum(rec + e, (long long)(mnmf_tail_doubles<M>() / 2), split);
      dv[u] = sqrt(s.x / s.y) * dv[u];
    }
  }
  const double *qb = qbuf + (long long)b * F * M;
  {
    const int m = threadIdx.x % M;
    const int stride = (256 / M) * M;  // thread t sums entries t, t + stride, ...: all of channel m
    double local = 0.0;
    if ((int)threadIdx.x < stride) {
      const int total = F * M;
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
    for (int m2 = 0; m2 < M; ++m2) {
      const double tot = wave_sum(m == m2 ? local : 0.0);
      if (lane == 0) wsum[wave][m2] = tot;
    }
    __syncthreads();
    if ((int)threadIdx.x < M) {
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += wsum[w][threadIdx.x];
      v = v / (double)F;
      v = v < 0.0 ? 0.0 : v;
      psi[threadIdx.x] = apply_floor(sqrt(v), floor_kind, eps);
    }
  }
  __syncthreads();
  if (pscale && blockIdx.x == 0 && threadIdx.x < M) {
    const double p2 = psi[threadIdx.x] * psi[threadIdx.x];
    pscale[b * M + threadIdx.x] = (pscale_fresh ? 1.0 : pscale[b * M + threadIdx.x]) / p2;
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < nb * M * M) {
      const int m = (e / M) % M;
      Qb[e] = cmake(qv[u].x / psi[m], qv[u].y / psi[m]);
    }
  }
#pragma unroll
  for (int u = 0; u < DPT; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < nb * N * M) {
      const int m = e % M;
      Db[e] = dv[u] / (psi[m] * psi[m]);
    }
  }
}

// ================================================================= multichannel Wiener filter
// Qinv[b,i] = Q[b,i]^-1
template <int M>
__global__ __launch_bounds__(64) void k_mnmf_qinv(const c128 *__restrict__ Q, c128 *Qinv,
                                                  long long nbins, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<M> A, Inv;
  load_mat<M>(A, Q + idx * (M * M));
  const bool ok = invert<M>(A, Inv);
  store_mat<M>(Inv, Qinv + idx * (M * M));
  if (!ok && info) atomicAdd(info, 1);
}

// The closed form alone (see k_mnmf_separate below for the bound): when min_m R~_m clears the
// eigenvalue floor at every frame of a bin -- the normal case with the default floor of 1e-10 --
// the bin is finished here, by a kernel that carries no eigen-decomposition and so runs at full
// occupancy; otherwise redo[b, i] is set and the general kernel recomputes the bin.
// grid: (F, B); lanes along frames.
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_separate_closed(
    const c128 *__restrict__ X, const c128 *__restrict__ Q, const c128 *__restrict__ Qinv,
    const double *__restrict__ Dsp, const double *__restrict__ basis,
    const double *__restrict__ act, c128 *Y, Dims d, int ref, int floor_kind, double eps,
    int *__restrict__ redo) {
  const int i = blockIdx.x, b = blockIdx.y;
  const int F = d.F, T = d.T, K = d.K;
  __shared__ c128 qref[M];
  __shared__ c128 qsrc[M * M];
  __shared__ double dd[N * M];
  if (threadIdx.x < M) qref[threadIdx.x] = Qinv[((long long)b * F + i) * (M * M) + ref * M + threadIdx.x];
  if (threadIdx.x < M * M) qsrc[threadIdx.x] = Q[((long long)b * F + i) * (M * M) + threadIdx.x];
  if (threadIdx.x < N * M) dd[threadIdx.x] = Dsp[((long long)b * F + i) * (N * M) + threadIdx.x];
  __syncthreads();
  if (floor_kind == SSSPY_FLOOR_ADD) {  // add-flooring shifts every eigenvalue: general path
    if (threadIdx.x == 0) redo[(long long)b * F + i] = 1;
    return;
  }
  double qf2 = 0.0;
#pragma unroll
  for (int e = 0; e < M * M; ++e) qf2 += cabs2(qsrc[e]);
  bool bad = false;
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    double lam[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *tr = basis + (((long long)b * N + n) * F + i) * K;
      const double *Vn = act + ((long long)b * N + n) * K * T;
      double r = 0.0;
      for (int k = 0; k < K; ++k) r = fma(tr[k], Vn[(long long)k * T + j], r);
      lam[n] = r;
    }
    double rc[M];
    double rcmin = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double r = 0.0;
#pragma unroll
      for (int n = 0; n < N; ++n) r = fma(lam[n], dd[n * M + m], r);
      rc[m] = r;
      rcmin = m == 0 ? r : (r < rcmin ? r : rcmin);
    }
    if (!(rcmin > eps * qf2 * 1.0000001)) {
      bad = true;
      continue;
    }
    c128 x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + i) * T + j];
    // s_m = (Q x)_m / rc_m ;  Y_n = sum_m lam_n d_nm q~[ref][m] s_m  (the operations of the general
    // kernel's closed branch, in its order)
    c128 sm[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      c128 y = cmake(0.0, 0.0);
#pragma unroll
      for (int a = 0; a < M; ++a) cfma(y, qsrc[m * M + a], x[a]);
      const double g = 1.0 / rc[m];
      sm[m] = cmul(qref[m], cmake(y.x * g, y.y * g));
    }
#pragma unroll
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
  if (bad) redo[(long long)b * F + i] = 1;
}

// The same for n_basis <= 16 with a thread owning ONE frame of SEP_BINS consecutive bins: the
// activations of its frame stay in registers across the bins (the per-bin kernel above fetches
// N K values per point, 4 GB of L2 traffic at 32 mixtures), and everything per-bin -- basis rows, Q,
// row `ref` of Q^-1, D -- is block-uniform and arrives through scalar loads, so there is no LDS and
// no barrier.  grid: (ceil(F / SEP_BINS), ceil(T / 256), B).
constexpr int SEP_BINS = 16;
template <int M, int KMAX>
__global__ __launch_bounds__(256) void k_mnmf_separate_closed_rows(
    const c128 *__restrict__ X, const c128 *__restrict__ Q, const c128 *__restrict__ Qinv,
    const double *__restrict__ Dsp, const double *__restrict__ basis,
    const double *__restrict__ act, c128 *Y, Dims d, int ref, double eps,
    int *__restrict__ redo) {
  // KMAX (round 5): 8 for n_basis <= 8 -- half the activation registers of the 16-wide form (224 ->
  // 130 VGPRs: three waves per SIMD instead of two) and no FMAs on zero padding; the x samples of the
  // NEXT bin are requested before this bin's arithmetic (the walk waited for four dependent loads
  // per bin: 2.2 TB/s), and 1 / R~ is v_rcp_f64 + two Newton steps like everywhere else.
  const int b = blockIdx.z;
  const int F = d.F, T = d.T, K = d.K;
  const int j = blockIdx.y * 256 + threadIdx.x;
  const bool fvalid = j < T;
  const int jc = fvalid ? j : T - 1;
  double v[N][KMAX];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      v[n][k] = k < K ? act[(((long long)b * N + n) * K + k) * T + jc] : 0.0;
  const int i_begin = blockIdx.x * SEP_BINS;
  const int i_end = min(F, i_begin + SEP_BINS);
  c128 xnext[M];
#pragma unroll
  for (int m = 0; m < M; ++m) xnext[m] = X[(((long long)b * M + m) * F + i_begin) * T + jc];
  for (int i = i_begin; i < i_end; ++i) {
    c128 x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = xnext[m];
    {
      const int inx = min(i + 1, i_end - 1);
#pragma unroll
      for (int m = 0; m < M; ++m) xnext[m] = X[(((long long)b * M + m) * F + inx) * T + jc];
    }
    const c128 *__restrict__ qsrc = Q + ((long long)b * F + i) * (M * M);
    const c128 *__restrict__ qref = Qinv + ((long long)b * F + i) * (M * M) + ref * M;
    const double *__restrict__ dd = Dsp + ((long long)b * F + i) * (N * M);
    double qf2 = 0.0;
#pragma unroll
    for (int e = 0; e < M * M; ++e) qf2 += cabs2(qsrc[e]);
    double lam[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *__restrict__ tr = basis + (((long long)b * N + n) * F + i) * K;
      double r = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) r = fma(tr[k], v[n][k], r);
      lam[n] = r;
    }
    double rc[M];
    double rcmin = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double r = 0.0;
#pragma unroll
      for (int n = 0; n < N; ++n) r = fma(lam[n], dd[n * M + m], r);
      rc[m] = r;
      rcmin = m == 0 ? r : (r < rcmin ? r : rcmin);
    }
    if (!fvalid) continue;
    if (!(rcmin > eps * qf2 * 1.0000001)) {
      redo[(long long)b * F + i] = 1;
      continue;
    }
    c128 sm[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      c128 y = cmake(0.0, 0.0);
#pragma unroll
      for (int a = 0; a < M; ++a) cfma(y, qsrc[m * M + a], x[a]);
      const double g = rcp_nr(rc[m]);
      sm[m] = cmul(qref[m], cmake(y.x * g, y.y * g));
    }
#pragma unroll
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

// grid: (F, B); lanes along frames.  Y_n = sum_m lam_n d_nm q~[ref][m] s_m,
// s = Q~^H R^-1 x, R = to_psd(sum_m R~_m q~_m q~_m^H) (eigenvalues floored).
template <int M>
__global__ __launch_bounds__(256) void k_mnmf_separate(const c128 *__restrict__ X,
                                                       const c128 *__restrict__ Q,
                                                       const c128 *__restrict__ Qinv,
                                                       const double *__restrict__ Dsp,
                                                       const double *__restrict__ basis,
                                                       const double *__restrict__ act, c128 *Y,
                                                       Dims d, int ref, int floor_kind,
                                                       double eps, const int *__restrict__ redo) {
  const int i = blockIdx.x, b = blockIdx.y;
  const int F = d.F, T = d.T, K = d.K;
  // (redo: only the bins the closed-form kernel below could not finish)
  if (redo && redo[(long long)b * F + i] == 0) return;
  __shared__ c128 qt[M * M];
  __shared__ c128 qsrc[M * M];
  __shared__ double dd[N * M];
  if (threadIdx.x < M * M) {
    qt[threadIdx.x] = Qinv[((long long)b * F + i) * (M * M) + threadIdx.x];
    qsrc[threadIdx.x] = Q[((long long)b * F + i) * (M * M) + threadIdx.x];
  }
  if (threadIdx.x < N * M) dd[threadIdx.x] = Dsp[((long long)b * F + i) * (N * M) + threadIdx.x];
  __syncthreads();
  // ||Q||_F^2 of the bin: R = Q~ diag(rc) Q~^H with Q~ = Q^-1 has lambda_min(R) >= min_m rc_m / ||Q||_2^2
  // >= min_m rc_m / ||Q||_F^2, so when that bound clears the eigenvalue floor, to_psd leaves R
  // untouched and R^-1 = Q^H diag(1 / rc) Q in closed form -- no eigen-decomposition
  // (add-flooring shifts every eigenvalue and always takes the general path).
  double qf2 = 0.0;
#pragma unroll
  for (int e = 0; e < M * M; ++e) qf2 += cabs2(qsrc[e]);
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    double lam[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *tr = basis + (((long long)b * N + n) * F + i) * K;
      const double *Vn = act + ((long long)b * N + n) * K * T;
      double r = 0.0;
      for (int k = 0; k < K; ++k) r = fma(tr[k], Vn[(long long)k * T + j], r);
      lam[n] = r;
    }
    double rc[M];
    double rcmin = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double r = 0.0;
#pragma unroll
      for (int n = 0; n < N; ++n) r = fma(lam[n], dd[n * M + m], r);
      rc[m] = r;
      rcmin = m == 0 ? r : (r < rcmin ? r : rcmin);
    }
    c128 x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + i) * T + j];
    if (floor_kind != SSSPY_FLOOR_ADD && rcmin > eps * qf2 * 1.0000001) {
      // s_m = (Q x)_m / rc_m ;  Y_n = sum_m lam_n d_nm q~[ref][m] s_m
      c128 sm[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < M; ++a) cfma(y, qsrc[m * M + a], x[a]);
        const double g = 1.0 / rc[m];
        sm[m] = cmul(qt[ref * M + m], cmake(y.x * g, y.y * g));
      }
#pragma unroll
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
      continue;
    }
    // R = sum_m rc[m] q~_m q~_m^H  (q~_m = column m of Q^-1), Hermitian by construction
    c128 A[M][M], P[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int c2 = a; c2 < M; ++c2) {
        c128 s = cmake(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const c128 z = cmulc(qt[a * M + m], qt[c2 * M + m]);
          s.x = fma(rc[m], z.x, s.x);
          s.y = fma(rc[m], z.y, s.y);
        }
        if (a == c2) s.y = 0.0;
        A[a][c2] = s;
        A[c2][a] = cconj(s);
      }
    jacobi_eigh<M>(A, P);
    // z = P diag(1/floor(lam)) P^H x
    c128 z[M];
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
      const double ev = apply_floor(A[k][k].x, floor_kind, eps);
      proj = cmake(proj.x / ev, proj.y / ev);
#pragma unroll
      for (int a = 0; a < M; ++a) cfma(z[a], P[a][k], proj);
    }
    // s_m = sum_c conj(q~[c][m]) z_c ;  Y_n = sum_m lam_n d_nm q~[ref][m] s_m
    c128 sm[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int c2 = 0; c2 < M; ++c2) {
        const c128 qv = qt[c2 * M + m];
        s.x += qv.x * z[c2].x + qv.y * z[c2].y;
        s.y += qv.x * z[c2].y - qv.y * z[c2].x;
      }
      sm[m] = cmul(qt[ref * M + m], s);
    }
#pragma unroll
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

}  // namespace mnmf_n<N>
using namespace SSSPY_CAT(mnmf_n, SSSPY_N);

#define MNMF_DISPATCH_M(M_, CALL)                                                          \
  switch (M_) {                                                                            \
    case 2: { constexpr int MM = 2; CALL; } break;                                         \
    case 3: { constexpr int MM = 3; CALL; } break;                                         \
    case 4: { constexpr int MM = 4; CALL; } break;                                         \
    default: return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_channels must be in [2, 4]"); \
  }

static inline int kt_count(int K) { return (K + 15) / 16; }
template <int M>
constexpr int cov_lds_mm() {
  return cov_lds_doubles_per_wave<M, M>();
}

// The bin-split variants run on the two-level schedule of tail_plan.hpp: whole rounds of 512
// workgroups unsplit, the remainder (or a small batch) split along the frames.
static inline bool mnmf_fast_ok(int B, int F, int T, int K) {
  static const bool disabled = std::getenv("SSSPY_AMD_NO_FAST") != nullptr;
  // one mixture must fit a 32-bit buffer descriptor (up to 4 channels of complex128)
  return !disabled && K <= 16 && (long long)4 * F * T * 16 < (1ll << 32);
}
static inline TailPlan mnmf_plan(int B, int F, int T) {
  // these kernels hold one workgroup per CU (x prefetch in registers, > 256 VGPR + AGPR)
  return make_tail_plan(B, (F + 63) / 64, (T + 15) / 16, 256);
}
// private activation tiles per wave (n_basis <= 8; the barrier-free form of k_mnmf_binmajor_glds)
// Measured (benchmarks/tools/mnmf_steps.py, configs[3] shape): the covariance pass gains at every
// batch (32 mixtures: 267 -> 253 us, 128: 964 -> 908); the spatial pass, bound by its read + write
// stream, does not (401 -> 400, 1259 -> 1309) -- so only the covariance pass takes it.  (The A / B
// switches of round 5 -- private tiles in both passes, |Q x|^2 stores transposed through LDS,
// register-fed passes at tile-aligned frame counts -- went in round 6 with their instantiations.)
// the LDS-DMA form of the two x-reading passes (k_mnmf_binmajor_glds): whole tiles of frames
static inline bool mnmf_glds_ok(int B, int F, int T, int K) {
  return mnmf_fast_ok(B, F, T, K) && T % 16 == 0;
}

// whether the |Q x|^2 hand-over (P, pscale) is taken by the passes of this shape
int LAUNCHER(mnmf_handover_ok)(int B, int F, int T, int K) {
  return (mnmf_fast_ok(B, F, T, K) && T % 2 == 0) ? 1 : 0;
}

int LAUNCHER(mnmf_qx2)(const void *X, const void *Q, double *P, double *pscale, int B, int M,
                       int F, int T, hipStream_t st) {
  dim3 grid((T + 255) / 256, F, B);
  MNMF_DISPATCH_M(M, hipLaunchKernelGGL((k_mnmf_qx2<MM>), grid, dim3(256), 0, st, (const c128 *)X,
                                        (const c128 *)Q, P, pscale, F, T));
  return check_launch("k_mnmf_qx2");
}

int LAUNCHER(mnmf_basis)(const void *X, const void *Q, const double *Dsp, const double *basis,
                         double *basis_out, const double *act, int B, int M, int F, int T, int K,
                         int floor_kind, double eps, double *tailpart, const double *P,
                         const double *pscale, hipStream_t st) {
  Dims d{B, F, T, K};
  if (mnmf_fast_ok(B, F, T, K) && basis_out == basis) {
    // two workgroups per CU with the hand-over (no x tiles, no Q in registers)
    const TailPlan plan = P ? make_tail_plan(B, (F + 63) / 64, (T + 15) / 16, 256 * PBASIS_WAVES)
                            : mnmf_plan(B, F, T);
    dim3 fgrid(plan.full + plan.tail * plan.split);
    MNMF_DISPATCH_M(M, {
      if (P && K <= 8)
        hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_BASIS, P_READ, 2>), fgrid, dim3(256), 0,
                           st, (const c128 *)X, (const c128 *)Q, (double *)Dsp, basis_out, act,
                           (c128 *)nullptr, F, T, K, floor_kind, eps, plan, tailpart,
                           const_cast<double *>(P), pscale);
      else if (P)
        hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_BASIS, P_READ>), fgrid, dim3(256), 0, st,
                           (const c128 *)X, (const c128 *)Q, (double *)Dsp, basis_out, act,
                           (c128 *)nullptr, F, T, K, floor_kind, eps, plan, tailpart,
                           const_cast<double *>(P), pscale);
      else
      hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_BASIS>), fgrid, dim3(256), 0, st,
                         (const c128 *)X, (const c128 *)Q, (double *)Dsp, basis_out, act,
                         (c128 *)nullptr, F, T, K, floor_kind, eps, plan, tailpart,
                         (double *)nullptr, (const double *)nullptr);
      if (plan.tail > 0)
        hipLaunchKernelGGL((k_mnmf_basis_finalize<MM>), dim3(N * 64 * 16 / 256, plan.tail), dim3(256),
                           0, st, basis_out, tailpart, F, K, plan, floor_kind, eps);
    });
    return check_launch("k_mnmf_basis_fast");
  }
  dim3 grid((F + 15) / 16, kt_count(K), B), block(256);
  const size_t lds = (size_t)4 * N * 2 * 256 * sizeof(double);
  MNMF_DISPATCH_M(M, {
    if (K <= 16)
      hipLaunchKernelGGL((k_mnmf_basis<MM, true>), grid, block, lds, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, basis_out, act, d, floor_kind, eps);
    else
      hipLaunchKernelGGL((k_mnmf_basis<MM, false>), grid, block, lds, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, basis_out, act, d, floor_kind, eps);
  });
  return check_launch("k_mnmf_basis");
}

int LAUNCHER(mnmf_activation)(const void *X, const void *Q, const double *Dsp, const double *basis,
                              const double *act, double *part, int nchunks, int B, int M, int F,
                              int T, int K, const double *P, const double *pscale,
                              hipStream_t st) {
  Dims d{B, F, T, K};
  const int ntiles = (F + 15) / 16;
  const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
  const int ktiles = kt_count(K);
  dim3 grid((T + 63) / 64, nchunks, B * ktiles), block(256);
  if (mnmf_fast_ok(B, F, T, K)) {
    MNMF_DISPATCH_M(M, {
      if (P)
        hipLaunchKernelGGL((k_mnmf_activation_fast<MM, true>), grid, block, 0, st, (const c128 *)X,
                           (const c128 *)Q, Dsp, basis, act, part, F, T, K, tiles_per_chunk, nchunks,
                           P, pscale);
      else
        hipLaunchKernelGGL((k_mnmf_activation_fast<MM, false>), grid, block, 0, st,
                           (const c128 *)X, (const c128 *)Q, Dsp, basis, act, part, F, T, K,
                           tiles_per_chunk, nchunks, (const double *)nullptr,
                           (const double *)nullptr);
    });
    return check_launch("k_mnmf_activation_fast");
  }
  MNMF_DISPATCH_M(M, {
    if (K <= 16)
      hipLaunchKernelGGL((k_mnmf_activation<MM, true>), grid, block, 0, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, act, part, d, ktiles, tiles_per_chunk,
                         nchunks);
    else
      hipLaunchKernelGGL((k_mnmf_activation<MM, false>), grid, block, 0, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, act, part, d, ktiles, tiles_per_chunk,
                         nchunks);
  });
  return check_launch("k_mnmf_activation");
}

// split_out / rec_out (optional, a caller that can fold the partial records itself -- IP1's latency
// form): when every item was split the fold is skipped, *split_out = chunks per item (else 0) and
// *rec_out = c128 between records; the records sit in tailpart as [(b groups + group) split + ch],
// 64 bins x M^3 each.
int LAUNCHER(mnmf_wcov)(const void *X, const double *Dsp, const double *basis, const double *act,
                        void *U, int B, int M, int F, int T, int K, double *tailpart,
                        int *split_out, long long *rec_out, hipStream_t st) {
  Dims d{B, F, T, K};
  if (split_out) *split_out = 0;
  if (mnmf_fast_ok(B, F, T, K) && tailpart) {
    const TailPlan plan = mnmf_plan(B, F, T);
    dim3 fgrid(plan.full + plan.tail * plan.split);
    const bool records = split_out && rec_out && plan.full == 0 && plan.tail > 0;
    const bool glds = mnmf_glds_ok(B, F, T, K);
    MNMF_DISPATCH_M(M, {
      if (glds && K <= 8)
        hipLaunchKernelGGL((k_mnmf_binmajor_glds<MM, MODE_WCOV, 2, true>), fgrid, dim3(256), 0, st,
                           (const c128 *)X, (const c128 *)nullptr, (double *)Dsp, basis, act,
                           (c128 *)U, F, T, K, plan, tailpart, (double *)nullptr);
      else if (glds)
        hipLaunchKernelGGL((k_mnmf_binmajor_glds<MM, MODE_WCOV, 4>), fgrid, dim3(256), 0, st,
                           (const c128 *)X, (const c128 *)nullptr, (double *)Dsp, basis, act,
                           (c128 *)U, F, T, K, plan, tailpart, (double *)nullptr);
      else
      hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_WCOV>), fgrid, dim3(256), 0, st,
                         (const c128 *)X, (const c128 *)nullptr, (double *)Dsp, (double *)basis, act,
                         (c128 *)U, F, T, K, 0, 0.0, plan, tailpart, (double *)nullptr,
                         (const double *)nullptr);
      if (records) {
        *split_out = plan.split;
        *rec_out = mnmf_tail_doubles<MM>() / 2;
      } else if (plan.tail > 0) {
        hipLaunchKernelGGL((k_mnmf_wcov_fold<MM>), dim3((64 * MM * MM * MM + 255) / 256, plan.tail),
                           dim3(256), 0, st, (c128 *)U, tailpart, F, plan);
      }
    });
    return check_launch("k_mnmf_wcov_fast");
  }
  dim3 grid((F + 15) / 16, 1, B), block(256);
  MNMF_DISPATCH_M(M, {
    const size_t lds = (size_t)4 * cov_lds_mm<MM>() * sizeof(double);
    if (K <= 16)
      hipLaunchKernelGGL((k_mnmf_wcov<MM, true>), grid, block, lds, st, (const c128 *)X, Dsp, basis,
                         act, (c128 *)U, d);
    else
      hipLaunchKernelGGL((k_mnmf_wcov<MM, false>), grid, block, lds, st, (const c128 *)X, Dsp,
                         basis, act, (c128 *)U, d);
  });
  return check_launch("k_mnmf_wcov");
}

// P / pscale (optional): the pass also writes |Q x|^2 for the next basis and activation passes.
// scale_follows: the caller normalises next and lets k_mnmf_norm_scale store the scale
// (pscale_fresh), so it is not reset to 1 here.
// split_out (optional; only a caller that runs mnmf_norm_scale next may pass it): when every item
// was split, the fold of the partial records is left to that kernel and *split_out = chunks per
// item (else 0).
int LAUNCHER(mnmf_spatial)(const void *X, const void *Q, double *Dsp, const double *basis,
                           const double *act, int B, int M, int F, int T, int K, double *tailpart,
                           double *P, double *pscale, int scale_follows, int *split_out,
                           hipStream_t st) {
  if (split_out) *split_out = 0;
  Dims d{B, F, T, K};
  if (mnmf_fast_ok(B, F, T, K)) {
    const TailPlan plan = mnmf_plan(B, F, T);
    dim3 fgrid(plan.full + plan.tail * plan.split);
    if (P && !scale_follows)
      hipLaunchKernelGGL(k_mnmf_fill_ones, dim3((B * M + 255) / 256), dim3(256), 0, st, pscale,
                         B * M);
    const bool glds = P && mnmf_glds_ok(B, F, T, K);
    MNMF_DISPATCH_M(M, {
      if (glds && K <= 8)
        hipLaunchKernelGGL((k_mnmf_binmajor_glds<MM, MODE_SPATIAL, 2>), fgrid, dim3(256), 0, st,
                           (const c128 *)X, (const c128 *)Q, Dsp, basis, act, (c128 *)nullptr, F,
                           T, K, plan, tailpart, P);
      else if (glds)
        hipLaunchKernelGGL((k_mnmf_binmajor_glds<MM, MODE_SPATIAL, 4>), fgrid, dim3(256), 0, st,
                           (const c128 *)X, (const c128 *)Q, Dsp, basis, act, (c128 *)nullptr, F,
                           T, K, plan, tailpart, P);
      else if (P)
        hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_SPATIAL, P_WRITE>), fgrid, dim3(256), 0,
                           st, (const c128 *)X, (const c128 *)Q, Dsp, (double *)basis, act,
                           (c128 *)nullptr, F, T, K, 0, 0.0, plan, tailpart, P,
                           (const double *)nullptr);
      else
      hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_SPATIAL>), fgrid, dim3(256), 0, st,
                         (const c128 *)X, (const c128 *)Q, Dsp, (double *)basis, act,
                         (c128 *)nullptr, F, T, K, 0, 0.0, plan, tailpart, (double *)nullptr,
                         (const double *)nullptr);
      if (split_out && plan.full == 0 && plan.tail > 0)
        *split_out = plan.split;
      else if (plan.tail > 0)
        hipLaunchKernelGGL((k_mnmf_spatial_finalize<MM>), dim3((64 * N * MM + 255) / 256, plan.tail),
                           dim3(256), 0, st, Dsp, tailpart, F, plan);
    });
    return check_launch("k_mnmf_spatial_fast");
  }
  dim3 grid((F + 15) / 16, 1, B), block(256);
  MNMF_DISPATCH_M(M, {
    const size_t lds = (size_t)4 * 2 * N * MM * 16 * sizeof(double);
    if (K <= 16)
      hipLaunchKernelGGL((k_mnmf_spatial<MM, true>), grid, block, lds, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, act, d);
    else
      hipLaunchKernelGGL((k_mnmf_spatial<MM, false>), grid, block, lds, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, act, d);
  });
  return check_launch("k_mnmf_spatial");
}

// scratch of the deterministic loss sums (per-block / per-wave shares, folded in a fixed order)
size_t LAUNCHER(mnmf_loss_ws_bytes)(int B, int F) {
  const int a = (F + 15) / 16, b = ((F + 63) / 64) * 16 * 4;
  return scalar_slots_bytes(B, a > b ? a : b);
}

// out[b] = the data term of the loss; loss_ws: mnmf_loss_ws_bytes()
int LAUNCHER(mnmf_loss)(const void *X, const void *Q, const double *Dsp, const double *basis,
                        const double *act, double *out, void *loss_ws, int B, int M, int F, int T,
                        int K, hipStream_t st) {
  Dims d{B, F, T, K};
  dim3 grid((F + 15) / 16, 1, B), block(256);
  // (every block writes its slot: no reset needed)
  MNMF_DISPATCH_M(M, {
    if (K <= 16)
      hipLaunchKernelGGL((k_mnmf_loss<MM, true>), grid, block, 0, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, act, (double *)loss_ws, d);
    else
      hipLaunchKernelGGL((k_mnmf_loss<MM, false>), grid, block, 0, st, (const c128 *)X,
                         (const c128 *)Q, Dsp, basis, act, (double *)loss_ws, d);
  });
  const int rc = check_launch("k_mnmf_loss");
  return rc ? rc : scalar_slots_fold(loss_ws, B, (int)grid.x, out, 0, st);
}

// out[b] = the data term of the loss from the hand-over; loss_ws: mnmf_loss_ws_bytes()
int LAUNCHER(mnmf_loss_handover)(const double *Dsp, const double *basis, const double *act,
                                 const double *P, const double *pscale, double *out, void *loss_ws,
                                 int B, int M, int F, int T, int K, hipStream_t st) {
  if (!mnmf_fast_ok(B, F, T, K) || T % 2 != 0)
    return fail(SSSPY_ERR_UNSUPPORTED, "fastmnmf_loss_data_handover: no hand-over for this shape");
  const TailPlan plan = make_tail_plan(B, (F + 63) / 64, (T + 15) / 16, 256 * PBASIS_WAVES);
  dim3 fgrid(plan.full + plan.tail * plan.split);
  const int nslots = plan.groups * (plan.split > 1 ? plan.split : 1) * 4;
  int rc = scalar_slots_begin(loss_ws, B, nslots, st);
  if (rc) return rc;
  // (MODE_LOSS: `tailpart` = the slots, `floor_kind` = B, see the kernel)
  MNMF_DISPATCH_M(M, hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_LOSS, P_READ>), fgrid,
                                        dim3(256), 0, st, (const c128 *)nullptr,
                                        (const c128 *)nullptr, (double *)Dsp, (double *)basis, act,
                                        (c128 *)nullptr, F, T, K, B, 0.0, plan, (double *)loss_ws,
                                        const_cast<double *>(P), pscale));
  rc = check_launch("k_mnmf_loss_handover");
  return rc ? rc : scalar_slots_fold(loss_ws, B, nslots, out, 0, st);
}

// The same with the per-wave shares left RAW in the caller's array, share s of mixture b at
// slots[s * stride + b] (s < mnmf_loss_handover_slots): a record_loss run zeroes one array for all
// its iterations and folds it once, instead of two memsets and a fold launch per loss.
int LAUNCHER(mnmf_loss_handover_slots)(int B, int F, int T) {
  const TailPlan plan = make_tail_plan(B, (F + 63) / 64, (T + 15) / 16, 256 * PBASIS_WAVES);
  return plan.groups * (plan.split > 1 ? plan.split : 1) * 4;
}
int LAUNCHER(mnmf_loss_handover_raw)(const double *Dsp, const double *basis, const double *act,
                                     const double *P, const double *pscale, double *slots,
                                     long long stride, int B, int M, int F, int T, int K,
                                     hipStream_t st) {
  if (!mnmf_fast_ok(B, F, T, K) || T % 2 != 0)
    return fail(SSSPY_ERR_UNSUPPORTED, "fastmnmf_loss_data_handover: no hand-over for this shape");
  const TailPlan plan = make_tail_plan(B, (F + 63) / 64, (T + 15) / 16, 256 * PBASIS_WAVES);
  dim3 fgrid(plan.full + plan.tail * plan.split);
  // (MODE_LOSS: `tailpart` = the slots, `floor_kind` = their stride, see the kernel)
  MNMF_DISPATCH_M(M, hipLaunchKernelGGL((k_mnmf_binmajor_fast<MM, MODE_LOSS, P_READ>), fgrid,
                                        dim3(256), 0, st, (const c128 *)nullptr,
                                        (const c128 *)nullptr, (double *)Dsp, (double *)basis, act,
                                        (c128 *)nullptr, F, T, K, (int)stride, 0.0, plan, slots,
                                        const_cast<double *>(P), pscale));
  return check_launch("k_mnmf_loss_handover");
}

int LAUNCHER(mnmf_norm_scale)(void *Q, double *Dsp, const double *qbuf, int B, int M, int F,
                              int floor_kind, double eps, double *pscale, int pscale_fresh,
                              const double *spatial_records, int spatial_split, hipStream_t st) {
  // spatial_records / spatial_split: see mnmf_spatial's split_out
  dim3 grid((F + 63) / 64, B), block(256);
  MNMF_DISPATCH_M(M, hipLaunchKernelGGL((k_mnmf_norm_scale<MM>), grid, block, 0, st, (c128 *)Q, Dsp,
                                        qbuf, F, floor_kind, eps, pscale, pscale_fresh,
                                        spatial_split ? spatial_records : (const double *)nullptr,
                                        spatial_split));
  return check_launch("k_mnmf_norm_scale");
}

// redo: B F ints of scratch (bins the closed-form kernel hands to the general one)
int LAUNCHER(mnmf_separate)(const void *X, const void *Q, void *Qinv, const double *Dsp,
                            const double *basis, const double *act, void *Y, int B, int M, int F,
                            int T, int K, int ref, int floor_kind, double eps, int *info,
                            int *redo, hipStream_t st) {
  Dims d{B, F, T, K};
  const long long nbins = (long long)B * F;
  hipError_t e = hipMemsetAsync(redo, 0, (size_t)nbins * sizeof(int), st);
  if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  MNMF_DISPATCH_M(M, {
    hipLaunchKernelGGL((k_mnmf_qinv<MM>), dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                       (const c128 *)Q, (c128 *)Qinv, nbins, info);
    if (K <= 8 && floor_kind != SSSPY_FLOOR_ADD)
      hipLaunchKernelGGL((k_mnmf_separate_closed_rows<MM, 8>),
                         dim3((F + SEP_BINS - 1) / SEP_BINS, (T + 255) / 256, B), dim3(256), 0, st,
                         (const c128 *)X, (const c128 *)Q, (const c128 *)Qinv, Dsp, basis, act,
                         (c128 *)Y, d, ref, eps, redo);
    else if (K <= 16 && floor_kind != SSSPY_FLOOR_ADD)
      hipLaunchKernelGGL((k_mnmf_separate_closed_rows<MM, 16>),
                         dim3((F + SEP_BINS - 1) / SEP_BINS, (T + 255) / 256, B), dim3(256), 0, st,
                         (const c128 *)X, (const c128 *)Q, (const c128 *)Qinv, Dsp, basis, act,
                         (c128 *)Y, d, ref, eps, redo);
    else
      hipLaunchKernelGGL((k_mnmf_separate_closed<MM>), dim3(F, B), dim3(256), 0, st,
                         (const c128 *)X, (const c128 *)Q, (const c128 *)Qinv, Dsp, basis, act,
                         (c128 *)Y, d, ref, floor_kind, eps, redo);
    hipLaunchKernelGGL((k_mnmf_separate<MM>), dim3(F, B), dim3(256), 0, st, (const c128 *)X,
                       (const c128 *)Q, (const c128 *)Qinv, Dsp, basis, act, (c128 *)Y, d, ref,
                       floor_kind, eps, (const int *)redo);
  });
  return check_launch("k_mnmf_separate");
}

}  // namespace ssspy
