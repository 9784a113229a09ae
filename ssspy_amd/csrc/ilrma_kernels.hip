// Gauss-ILRMA (MM source model) kernels for gfx950.
//
// Data stay in the reference's host layout: X (B,N,F,T) c128, W (B,F,N,N) c128,
// basis T (B,N,F,K) f64, activation V (B,N,K,T) f64.  All arithmetic is fp64.
//
// The three passes over X per iteration (basis, activation, weighted covariance) all need
// R = T V for every (source, bin, frame) and a contraction of an element-wise function of
// (|y|^2, R) against V (over frames) or T (over bins).  Both are dense and go to the f64
// matrix core (v_mfma_f64_16x16x4_f64) on 16 x 16 (bin x frame) tiles:
//
//   "bin-major tile"  (basis, covariance, loss):  D columns = 16 bins, D rows = 16 frames.
//       lane = q*16 + c owns bin i0+c and frames j0 + 4q + r (r = D register).  R^T comes out
//       of GEMM1 (A = V^T, B = T^T) in exactly the layout GEMM2 wants for its A operand
//       (rows = bins, contraction = frames), so |y|^2/R^2 and 1/R feed the second MFMA
//       straight from registers: num[bin,k] += a[bin,frame] V[k,frame].
//   "frame-major tile" (activation):  D columns = 16 frames, D rows = 16 bins.
//       lane owns frame j0+c and bins i0 + q + 4r.  GEMM1 (A = T, B = V) gives R, and the
//       element-wise result is the B operand of GEMM2: numV[k,frame] += T[bin,k] a[bin,frame].
//
// Sources are processed SG at a time per wave (all of them when N <= 4) so x is read once.
// This file is compiled once per N (-DSSSPY_N=<n>) to keep build time parallel.
#include "common.hpp"
#include "cov_core.hpp"
#include "ilrma_params.hpp"
#include "nmf_tile.hpp"

#ifndef SSSPY_N
#error "compile with -DSSSPY_N=<n_sources>"
#endif

#define SSSPY_CAT_(a, b) a##b
#define SSSPY_CAT(a, b) SSSPY_CAT_(a, b)
#define LAUNCHER(name) SSSPY_CAT(SSSPY_CAT(name, _n), SSSPY_N)

namespace ssspy {
// every per-N translation unit defines the same kernel templates with a different NSRC:
// give each its own namespace so the symbols (host stubs and device code) stay distinct.
namespace SSSPY_CAT(ilrma_n, SSSPY_N) {

constexpr int NSRC = SSSPY_N;
constexpr int SGRP = NSRC <= 4 ? NSRC : 2;  // sources per wave pass
constexpr int NGROUPS = (NSRC + SGRP - 1) / SGRP;

// ======================================================================== pass 1: basis update
// grid: (bin tiles, k tiles, B * NGROUPS); block: NW waves, wave w takes frame tiles w, w+NW, ...
template <bool KSMALL>
__global__ __launch_bounds__(256) void k_ilrma_basis(const c128 *__restrict__ X,
                                                     const c128 *__restrict__ W,
                                                     const double *basis, double *basis_out,
                                                     const double *__restrict__ act,
                                                     IlrmaDims d) {
  constexpr int N = NSRC, SG = SGRP;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z / NGROUPS, g = blockIdx.z % NGROUPS;
  const int s0 = g * SG;
  const int kt = blockIdx.y;  // 16-wide tile of basis indices this block produces
  const int i0 = blockIdx.x * 16;
  const int bin = min(i0 + c, F - 1);

  // demixing rows of this lane's bin for the sources of the group (identity when W == NULL:
  // the ISS path passes the separated spectrogram as X)
  c128 w[SG][N];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int m = 0; m < N; ++m)
      w[s][m] = W ? W[(((long long)b * F + bin) * N + n) * N + m] : cmake(m == n ? 1.0 : 0.0, 0.0);
  }
  double tb[SG][4];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      double t = 0.0;
      if (KSMALL && kk < K) t = basis[(((long long)b * N + n) * F + bin) * K + kk];
      tb[s][ks] = t;
    }
  }
  double4_t num[SG], den[SG];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    num[s] = double4_t{0.0, 0.0, 0.0, 0.0};
    den[s] = double4_t{0.0, 0.0, 0.0, 0.0};
  }

  const int ntiles = (T + 15) >> 4;
  const int k2 = kt * 16 + c;  // basis index of this lane as GEMM2 B-operand column
  const bool k2valid = k2 < K;
  const int k2c = k2valid ? k2 : K - 1;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
    c128 x[N][4];
    bool fval[4];
    int jcl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j0 + 4 * q + r;
      fval[r] = jj < T;
      jcl[r] = fval[r] ? jj : T - 1;
    }
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[m][r] = X[(((long long)b * N + m) * F + bin) * T + jcl[r]];

#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int n = min(s0 + s, N - 1);
      const double *Vn = act + ((long long)b * N + n) * K * T;
      const double *Tn = basis + (((long long)b * N + n) * F + bin) * K;
      const double4_t R = nmf_rt_tile<KSMALL>(Vn, Tn, tb[s], K, T, j0, c, q);
      double a[4], bb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) cfma(y, w[s][m], x[m][r]);
        mm_weights(cabs2(y), R[r], d, fval[r], a[r], bb[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double vb = Vn[(long long)k2c * T + jcl[r]];
        vb = (k2valid && fval[r]) ? vb : 0.0;
        num[s] = mfma_f64(a[r], vb, num[s]);
        den[s] = mfma_f64(bb[r], vb, den[s]);
      }
    }
  }

  // cross-wave fold: lds[((w*SG + s)*2 + nd)*256 + r*64 + lane]
#pragma unroll
  for (int s = 0; s < SG; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lds[((wave * SG + s) * 2 + 0) * 256 + r * 64 + lane] = num[s][r];
      lds[((wave * SG + s) * 2 + 1) * 256 + r * 64 + lane] = den[s][r];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < SG * 256; e += blockDim.x) {
    const int s = e >> 8, r = (e >> 6) & 3, ln = e & 63;
    double sn = 0.0, sd = 0.0;
    for (int wv = 0; wv < nw; ++wv) {
      sn += lds[((wv * SG + s) * 2 + 0) * 256 + r * 64 + ln];
      sd += lds[((wv * SG + s) * 2 + 1) * 256 + r * 64 + ln];
    }
    const int ob = i0 + (ln >> 4) + 4 * r;  // D row -> bin
    const int ok = kt * 16 + (ln & 15);     // D col -> basis index
    const int n = s0 + s;
    if (ob < F && ok < K && n < N) {
      const long long o = (((long long)b * N + n) * F + ob) * K + ok;
      if (d.raw) {
        basis_out[2 * o] = sn;
        basis_out[2 * o + 1] = sd;
      } else {
        basis_out[o] = apply_floor(mm_ratio_pow(sn, sd, d) * basis[o], d.floor_kind, d.floor_eps);
      }
    }
  }
}

// =================================================================== pass 2: activation update
// grid: (frame groups of 16*NW, bin chunks, (B*NGROUPS)*KTILES + kt); wave w owns frame tile
// blockIdx.x*NW + w and walks the bin tiles of its chunk; partial sums go to
// part[b][chunk][n][nd][K][T].
template <bool KSMALL>
__global__ __launch_bounds__(256) void k_ilrma_activation(const c128 *__restrict__ X,
                                                          const c128 *__restrict__ W,
                                                          const double *__restrict__ basis,
                                                          const double *__restrict__ act,
                                                          double *__restrict__ part, IlrmaDims d,
                                                          int ktiles, int tiles_per_chunk,
                                                          int nchunks) {
  constexpr int N = NSRC, SG = SGRP;
  __shared__ c128 wl[16 * N * N];  // demixing matrices of the current 16-bin tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int kt = blockIdx.z % ktiles;
  const int bg = blockIdx.z / ktiles;
  const int b = bg / NGROUPS, g = bg % NGROUPS;
  const int s0 = g * SG;
  const int chunk = blockIdx.y;
  const int j0 = (blockIdx.x * nw + wave) * 16;
  const int jf = j0 + c;
  const bool fvalid = jf < T;
  const int jc = fvalid ? jf : T - 1;

  // GEMM1 B operand: V[n, 4ks+q, frame]; fixed for the whole walk over bins
  double vb[SG][4];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      double v = 0.0;
      if (KSMALL && kk < K && fvalid) v = act[(((long long)b * N + n) * K + kk) * T + jc];
      vb[s][ks] = v;
    }
  }
  double4_t numv[SG], denv[SG];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    numv[s] = double4_t{0.0, 0.0, 0.0, 0.0};
    denv[s] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
  const int ntiles = (F + 15) >> 4;
  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = min(ntiles, t_begin + tiles_per_chunk);
  const int k2 = kt * 16 + c;  // basis index as GEMM2 A-operand row
  const bool k2valid = k2 < K;
  const int k2c = k2valid ? k2 : K - 1;

  for (int it = t_begin; it < t_end; ++it) {
    const int i0 = it * 16;
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * N * N; e += blockDim.x) {
      const int bl = e / (N * N), rem = e % (N * N);
      const int bi = min(i0 + bl, F - 1);
      wl[e] = W ? W[((long long)b * F + bi) * (N * N) + rem]
                : cmake((rem / N) == (rem % N) ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();

    c128 x[N][4];
    bool bval[4];
    int bcl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bi = i0 + q + 4 * r;
      bval[r] = bi < F;
      bcl[r] = bval[r] ? bi : F - 1;
    }
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[m][r] = X[(((long long)b * N + m) * F + bcl[r]) * T + jc];

#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int n = min(s0 + s, N - 1);
      const double *Tn = basis + ((long long)b * N + n) * F * K;
      // GEMM1: R[bin i0+q+4r, frame j0+c];  A[row = c -> bin i0+c][kk = q] = T[n, i0+c, 4ks+q]
      double4_t R = {0.0, 0.0, 0.0, 0.0};
      const int ab = min(i0 + c, F - 1);
      if (KSMALL) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks * 4 < K) {
            const int kk = ks * 4 + q;
            double ta = Tn[(long long)ab * K + (kk < K ? kk : K - 1)];
            ta = kk < K ? ta : 0.0;
            R = mfma_f64(ta, vb[s][ks], R);
          }
        }
      } else {
        for (int k0 = 0; k0 < K; k0 += 4) {
          const int kk = k0 + q;
          const int kc = kk < K ? kk : K - 1;
          double ta = Tn[(long long)ab * K + kc];
          ta = kk < K ? ta : 0.0;
          double v = act[(((long long)b * N + n) * K + kc) * T + jc];
          v = (kk < K && fvalid) ? v : 0.0;
          R = mfma_f64(ta, v, R);
        }
      }
      double a[4], bb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const c128 *wr = wl + ((q + 4 * r) * N + n) * N;
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) cfma(y, wr[m], x[m][r]);
        mm_weights(cabs2(y), R[r], d, bval[r] && fvalid, a[r], bb[r]);
      }
      // GEMM2: numV[k = kt*16 + (q+4r'), frame] += T[n, bin i0+q+4r, k2] * a[r]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double ta = Tn[(long long)bcl[r] * K + k2c];
        ta = (k2valid && bval[r]) ? ta : 0.0;
        numv[s] = mfma_f64(ta, a[r], numv[s]);
        denv[s] = mfma_f64(ta, bb[r], denv[s]);
      }
    }
  }
  // D: col = frame c, row = q + 4r -> basis index kt*16 + q + 4r
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = s0 + s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ok = kt * 16 + q + 4 * r;
      if (n < N && ok < K && fvalid) {
        const long long base = ((((long long)b * nchunks + chunk) * N + n) * 2) * K;
        part[(base + ok) * T + jf] = numv[s][r];
        part[(base + K + ok) * T + jf] = denv[s][r];
      }
    }
  }
}

// ============================================================ pass 3: NMF-weighted covariance
// U[b,i,n] = (1/T) sum_j x x^H / R^(2/p).  grid: (bin tiles, 1, B*NGROUPS)
template <bool KSMALL>
// For the heavy-tailed models the weight also depends on |y|^2 = |w x|^2: W gives the demixing rows
// (NULL = X already holds the separated spectrogram, the ISS state).
__global__ __launch_bounds__(256) void k_ilrma_wcov(const c128 *__restrict__ X,
                                                    const c128 *__restrict__ W,
                                                    const double *__restrict__ basis,
                                                    const double *__restrict__ act,
                                                    c128 *__restrict__ U, IlrmaDims d) {
  constexpr int N = NSRC, SG = SGRP;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z / NGROUPS, g = blockIdx.z % NGROUPS;
  const int s0 = g * SG;
  const int i0 = blockIdx.x * 16;
  const int bin = min(i0 + c, F - 1);
  double tb[SG][4];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      double t = 0.0;
      if (KSMALL && kk < K) t = basis[(((long long)b * N + n) * F + bin) * K + kk];
      tb[s][ks] = t;
    }
  }
  const bool need_y = d.model != SSSPY_SOURCE_GAUSS;
  c128 w[SG][N];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int m = 0; m < N; ++m)
      w[s][m] = (need_y && W) ? W[(((long long)b * F + bin) * N + n) * N + m]
                              : cmake(m == n ? 1.0 : 0.0, 0.0);
  }
  CovAcc<N, SG> acc;
  acc.clear();
  const int ntiles = (T + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
    double4_t R[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int n = min(s0 + s, N - 1);
      const double *Vn = act + ((long long)b * N + n) * K * T;
      const double *Tn = basis + (((long long)b * N + n) * F + bin) * K;
      R[s] = nmf_rt_tile<KSMALL>(Vn, Tn, tb[s], K, T, j0, c, q);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = j0 + 4 * q + r;
      const bool valid = jj < T;
      const int jc = valid ? jj : T - 1;
      c128 x[N];
#pragma unroll
      for (int m = 0; m < N; ++m) x[m] = X[(((long long)b * N + m) * F + bin) * T + jc];
      double phi[SG];
#pragma unroll
      for (int s = 0; s < SG; ++s) {
        double P = 0.0;
        if (need_y) {
          c128 y = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < N; ++m) cfma(y, w[s][m], x[m]);
          P = cabs2(y);
        }
        const double ph = spatial_weight(P, R[s][r], d);
        phi[s] = (valid && s0 + s < N) ? ph : 0.0;
      }
      acc.add(x, phi);
    }
  }
  acc.fold_q();
  cov_reduce_store<N, SG>(acc, lds, U, (long long)b * F, i0, F, N, s0, min(SG, N - s0),
                          1.0 / (double)T);
}

// =============================================================================== loss (data term)
// out[b] += sum_{n,i} (1/T) sum_j ( |y|^2 / R^(2/p) + (2/p) log R )
template <bool KSMALL>
__global__ __launch_bounds__(256) void k_ilrma_loss(const c128 *__restrict__ X,
                                                    const c128 *__restrict__ W,
                                                    const double *__restrict__ basis,
                                                    const double *__restrict__ act,
                                                    double *slots, IlrmaDims d) {
  constexpr int N = NSRC, SG = SGRP;
  __shared__ double scratch[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int F = d.F, T = d.T, K = d.K;
  const int b = blockIdx.z / NGROUPS, g = blockIdx.z % NGROUPS;
  const int s0 = g * SG;
  const int i0 = blockIdx.x * 16;
  const bool binvalid = i0 + c < F;
  const int bin = min(i0 + c, F - 1);
  c128 w[SG][N];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int m = 0; m < N; ++m)
      w[s][m] = W ? W[(((long long)b * F + bin) * N + n) * N + m] : cmake(m == n ? 1.0 : 0.0, 0.0);
  }
  double tb[SG][4];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    const int n = min(s0 + s, N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + q;
      double t = 0.0;
      if (KSMALL && kk < K) t = basis[(((long long)b * N + n) * F + bin) * K + kk];
      tb[s][ks] = t;
    }
  }
  double local = 0.0;
  const int ntiles = (T + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += nw) {
    const int j0 = jt * 16;
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int n = min(s0 + s, N - 1);
      const double *Vn = act + ((long long)b * N + n) * K * T;
      const double *Tn = basis + (((long long)b * N + n) * F + bin) * K;
      const double4_t R = nmf_rt_tile<KSMALL>(Vn, Tn, tb[s], K, T, j0, c, q);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = j0 + 4 * q + r;
        const bool valid = binvalid && jj < T && (s0 + s < N);
        const int jc = jj < T ? jj : T - 1;
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < N; ++m)
          cfma(y, w[s][m], X[(((long long)b * N + m) * F + bin) * T + jc]);
        const double term = loss_term(cabs2(y), R[r], d);
        local += valid ? term : 0.0;
      }
    }
  }
  const double total = block_sum(local, scratch);
  // one slot per (bin tile, source group) of the mixture; the launcher adds them up in order
  if (threadIdx.x == 0)
    slots[(long long)(blockIdx.x * NGROUPS + g) * d.B + b] = total / (double)T;
}

// ----------------------------------------------------------------------- host-side launchers
static inline int kt_count(int K) { return (K + 15) / 16; }

}  // namespace ilrma_n<N>
using namespace SSSPY_CAT(ilrma_n, SSSPY_N);

// basis_out may alias basis only when K <= 16 (one k tile per bin: nobody else reads the rows a
// block rewrites); for K > 16 the caller passes a scratch buffer and copies back.
int LAUNCHER(ilrma_basis)(const void *X, const void *W, const double *basis, double *basis_out,
                          const double *act, IlrmaDims d, hipStream_t st) {
  dim3 grid((d.F + 15) / 16, kt_count(d.K), d.B * NGROUPS), block(256);
  const size_t lds = (size_t)4 * SGRP * 2 * 256 * sizeof(double);
  if (d.K <= 16)
    hipLaunchKernelGGL((k_ilrma_basis<true>), grid, block, lds, st, (const c128 *)X,
                       (const c128 *)W, basis, basis_out, act, d);
  else
    hipLaunchKernelGGL((k_ilrma_basis<false>), grid, block, lds, st, (const c128 *)X,
                       (const c128 *)W, basis, basis_out, act, d);
  return check_launch("k_ilrma_basis");
}

int LAUNCHER(ilrma_activation)(const void *X, const void *W, const double *basis,
                               const double *act, double *part, int nchunks, IlrmaDims d,
                               hipStream_t st) {
  const int ntiles = (d.F + 15) / 16;
  const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
  const int ktiles = kt_count(d.K);
  dim3 grid((d.T + 63) / 64, nchunks, d.B * NGROUPS * ktiles), block(256);
  if (d.K <= 16)
    hipLaunchKernelGGL((k_ilrma_activation<true>), grid, block, 0, st, (const c128 *)X,
                       (const c128 *)W, basis, act, part, d, ktiles, tiles_per_chunk, nchunks);
  else
    hipLaunchKernelGGL((k_ilrma_activation<false>), grid, block, 0, st, (const c128 *)X,
                       (const c128 *)W, basis, act, part, d, ktiles, tiles_per_chunk, nchunks);
  return check_launch("k_ilrma_activation");
}

int LAUNCHER(ilrma_wcov)(const void *X, const void *W, const double *basis, const double *act,
                         void *U, IlrmaDims d, hipStream_t st) {
  dim3 grid((d.F + 15) / 16, 1, d.B * NGROUPS), block(256);
  const size_t lds = (size_t)4 * cov_lds_doubles_per_wave<NSRC, SGRP>() * sizeof(double);
  if (d.K <= 16)
    hipLaunchKernelGGL((k_ilrma_wcov<true>), grid, block, lds, st, (const c128 *)X, (const c128 *)W,
                       basis, act, (c128 *)U, d);
  else
    hipLaunchKernelGGL((k_ilrma_wcov<false>), grid, block, lds, st, (const c128 *)X,
                       (const c128 *)W, basis, act, (c128 *)U, d);
  return check_launch("k_ilrma_wcov");
}

// out[b] = the data term of the loss; loss_ws: ilrma_loss_ws_bytes() of scratch (per-block shares,
// added up in a fixed order: no atomics)
size_t LAUNCHER(ilrma_loss_ws_bytes)(int B, int F) {
  return scalar_slots_bytes(B, ((F + 15) / 16) * NGROUPS);
}
int LAUNCHER(ilrma_loss)(const void *X, const void *W, const double *basis, const double *act,
                         double *out, void *loss_ws, IlrmaDims d, hipStream_t st) {
  dim3 grid((d.F + 15) / 16, 1, d.B * NGROUPS), block(256);
  const int nslots = (int)grid.x * NGROUPS;
  int rc = scalar_slots_begin(loss_ws, d.B, nslots, st);
  if (rc) return rc;
  if (d.K <= 16)
    hipLaunchKernelGGL((k_ilrma_loss<true>), grid, block, 0, st, (const c128 *)X, (const c128 *)W,
                       basis, act, (double *)loss_ws, d);
  else
    hipLaunchKernelGGL((k_ilrma_loss<false>), grid, block, 0, st, (const c128 *)X,
                       (const c128 *)W, basis, act, (double *)loss_ws, d);
  rc = check_launch("k_ilrma_loss");
  return rc ? rc : scalar_slots_fold(loss_ws, d.B, nslots, out, 0, st);
}

}  // namespace ssspy
