// More than 8 sources / channels (up to SSSPY_RT_MAX_SOURCES = 16): the reference takes n_sources
// from input.shape with no limit (ssspy/bss/ilrma.py:180, iva.py:152); the kernels of the other
// units are compiled per source count with everything in registers, which stops at 8.  These are the
// same operators with the source count at run time -- loops instead of unrolled code, the per-bin
// matrices in the lane's private memory, the spectrogram passes staged through LDS a frame tile at a
// time: correct and simple, not tuned (round-3 verdict item 7b).  The entry points of
// spatial_kernels.hip / iva_kernels.hip / ilrma_api.hip call in here when N > SSSPY_MAX_SOURCES.
//
// Arithmetic follows the per-N kernels step for step (same pivot rule, same np.maximum(., 0) and
// flooring places, ref: ssspy/bss/_update_spatial_model.py:17-78, :146-194;
// ssspy/algorithm/projection_back.py:6-121).
#include "common.hpp"
#include "rt_dense.hpp"
#include "ssspy_amd.h"

namespace ssspy {

// ---- separate: Y = W X (in place allowed), or its power ------------------------------------------
// grid (F, B), 256 threads along frames
template <bool POWER>
__global__ __launch_bounds__(256) void k_separate_rt(const c128 *X, const c128 *__restrict__ W,
                                                     c128 *Y, int N, int F, int T) {
  const int i = blockIdx.x, b = blockIdx.y;
  const c128 *__restrict__ w = W + ((long long)b * F + i) * (N * N);
  const long long row0 = ((long long)b * N) * F + i;
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    c128 x[RTN], y[RTN];
    for (int m = 0; m < N; ++m) x[m] = X[(row0 + (long long)m * F) * T + j];
    for (int n = 0; n < N; ++n) {
      c128 acc = cmake(0.0, 0.0);
      for (int m = 0; m < N; ++m) cfma(acc, w[n * N + m], x[m]);
      y[n] = acc;
    }
    for (int n = 0; n < N; ++n) {
      if (POWER) reinterpret_cast<double *>(Y)[(row0 + (long long)n * F) * T + j] = cabs2(y[n]);
      else Y[(row0 + (long long)n * F) * T + j] = y[n];
    }
  }
}

// ---- (weighted / cross) covariance: C[s][a][c] = (1/T) sum_j w_sj A_a conj(B_c) ------------------
// grid (F, S, B), 256 threads; a tile of 32 frames of both operands sits in LDS, every thread owns
// the outputs e = tid, tid + 256, ... < N^2 and adds the tile's frames in order (deterministic).
constexpr int RT_TJ = 32;
__global__ __launch_bounds__(256) void k_cov_rt(const c128 *__restrict__ A,
                                                const c128 *__restrict__ Bm,
                                                const double *__restrict__ weight, int kind,
                                                c128 *__restrict__ C, int S, int N, int F, int T) {
  __shared__ c128 ta[RTN * RT_TJ], tb[RTN * RT_TJ];
  __shared__ double tw[RT_TJ];
  const int i = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const bool same = A == Bm;
  c128 acc[(RTN * RTN + 255) / 256];
  for (int u = 0; u < (RTN * RTN + 255) / 256; ++u) acc[u] = cmake(0.0, 0.0);
  for (int j0 = 0; j0 < T; j0 += RT_TJ) {
    const int nj = min(RT_TJ, T - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < N * RT_TJ; e += blockDim.x) {
      const int m = e / RT_TJ, jj = e % RT_TJ;
      const bool v = jj < nj;
      const long long off = (((long long)b * N + m) * F + i) * T + j0 + (v ? jj : 0);
      ta[e] = v ? A[off] : cmake(0.0, 0.0);
      if (!same) tb[e] = v ? Bm[off] : cmake(0.0, 0.0);
    }
    if (threadIdx.x < RT_TJ) {
      const int jj = threadIdx.x;
      double wv = 0.0;
      if (jj < nj) {
        if (kind == SSSPY_WEIGHT_UNIT) wv = 1.0;
        if (kind == SSSPY_WEIGHT_FRAME) wv = weight[((long long)b * S + s) * T + j0 + jj];
        if (kind == SSSPY_WEIGHT_BIN_FRAME)
          wv = weight[(((long long)b * S + s) * F + i) * T + j0 + jj];
      }
      tw[jj] = wv;
    }
    __syncthreads();
    const c128 *tbb = same ? ta : tb;
    for (int u = 0, e = threadIdx.x; e < N * N; e += blockDim.x, ++u) {
      const int a = e / N, c = e % N;
      c128 sum = acc[u];
      for (int jj = 0; jj < nj; ++jj) {
        const c128 z = cmulc(ta[a * RT_TJ + jj], tbb[c * RT_TJ + jj]);
        sum.x = fma(tw[jj], z.x, sum.x);
        sum.y = fma(tw[jj], z.y, sum.y);
      }
      acc[u] = sum;
    }
  }
  const double scale = 1.0 / (double)T;
  for (int u = 0, e = threadIdx.x; e < N * N; e += blockDim.x, ++u)
    C[(((long long)b * F + i) * S + s) * (long long)(N * N) + e] =
        cmake(acc[u].x * scale, acc[u].y * scale);
}

// ---- IP1: one lane per (mixture, bin) ------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ip1_rt(c128 *W, const c128 *__restrict__ U, long long nbins,
                                               int N, int floor_kind, double eps, int *info,
                                               const c128 *__restrict__ C, double *qbuf) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 Wm[RTN * RTN], A[RTN * RTN], w[RTN];
  for (int e = 0; e < N * N; ++e) Wm[e] = W[idx * (N * N) + e];
  bool ok = true;
  for (int n = 0; n < N; ++n) {
    const c128 *__restrict__ Un = U + (idx * N + n) * (long long)(N * N);
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) {
        c128 acc = cmake(0.0, 0.0);
        for (int k = 0; k < N; ++k) cfma(acc, Wm[r * N + k], Un[k * N + c]);
        A[r * N + c] = acc;
      }
    for (int r = 0; r < N; ++r) w[r] = cmake(r == n ? 1.0 : 0.0, 0.0);
    ok = rt_lu_solve(A, w, N, 1) && ok;
    double qf = rt_quad(w, Un, N);
    qf = qf < 0.0 ? 0.0 : qf;  // np.maximum(., 0): NaN propagates
    const double d = apply_floor(sqrt(qf), floor_kind, eps);
    for (int c = 0; c < N; ++c) Wm[n * N + c] = cmake(w[c].x / d, -w[c].y / d);
  }
  for (int e = 0; e < N * N; ++e) W[idx * (N * N) + e] = Wm[e];
  if (!ok && info) atomicAdd(info, 1);
  if (C && qbuf) {
    const c128 *__restrict__ Cm = C + idx * (long long)(N * N);
    for (int n = 0; n < N; ++n) {
      for (int m = 0; m < N; ++m) w[m] = cconj(Wm[n * N + m]);
      qbuf[idx * N + n] = rt_quad(w, Cm, N);
    }
  }
}

__global__ __launch_bounds__(64) void k_row_power_rt(const c128 *__restrict__ W,
                                                     const c128 *__restrict__ C, double *qbuf,
                                                     long long nbins, int N) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 v[RTN];
  for (int n = 0; n < N; ++n) {
    for (int m = 0; m < N; ++m) v[m] = cconj(W[idx * (N * N) + n * N + m]);
    qbuf[idx * N + n] = rt_quad(v, C + idx * (long long)(N * N), N);
  }
}

// ---- ISS1 on per-bin statistics (Vc[s] = mean_j varphi_s y y^H of the current Y) -> transform G ---
__global__ __launch_bounds__(64) void k_iss1_transform_rt(const c128 *__restrict__ Vc, c128 *G,
                                                          long long nbins, int N, int floor_kind,
                                                          double eps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 Gm[RTN * RTN], gn[RTN], v[RTN], t[RTN];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) Gm[r * N + c] = cmake(r == c ? 1.0 : 0.0, 0.0);
  const c128 *__restrict__ V0 = Vc + idx * (long long)(N * N) * N;
  for (int n = 0; n < N; ++n) {
    for (int c = 0; c < N; ++c) gn[c] = Gm[n * N + c];
    for (int s = 0; s < N; ++s) {
      for (int a = 0; a < N; ++a) {  // t = V0_s gn^H
        c128 acc = cmake(0.0, 0.0);
        for (int d = 0; d < N; ++d) {
          const c128 u = V0[((long long)s * N + a) * N + d];
          acc.x = fma(u.x, gn[d].x, acc.x);
          acc.x = fma(u.y, gn[d].y, acc.x);
          acc.y = fma(u.y, gn[d].x, acc.y);
          acc.y = fma(-u.x, gn[d].y, acc.y);
        }
        t[a] = acc;
      }
      c128 num = cmake(0.0, 0.0), dn = cmake(0.0, 0.0);
      for (int a = 0; a < N; ++a) cfma(num, Gm[s * N + a], t[a]);
      for (int a = 0; a < N; ++a) cfma(dn, gn[a], t[a]);
      const double den = apply_floor(dn.x, floor_kind, eps);
      const double inv = 1.0 / den;
      v[s] = s == n ? cmake(1.0 - 1.0 / sqrt(den), 0.0) : cmake(num.x * inv, num.y * inv);
    }
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) cfms(Gm[r * N + c], v[r], gn[c]);
  }
  for (int e = 0; e < N * N; ++e) G[idx * (N * N) + e] = Gm[e];
}

// ---- projection back / demixing filter from covariances / log-determinants -----------------------
// W[n, :] *= (W^-1)[ref, n]; G (optional) = the diagonal scale
__global__ __launch_bounds__(64) void k_pb_filter_rt(c128 *W, c128 *G, long long nbins, int N,
                                                     int ref, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  // row `ref` of W^-1 = the solution z of W^T z = e_ref
  c128 A[RTN * RTN], z[RTN];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) A[r * N + c] = W[idx * (N * N) + c * N + r];
  for (int r = 0; r < N; ++r) z[r] = cmake(r == ref ? 1.0 : 0.0, 0.0);
  const bool ok = rt_lu_solve(A, z, N, 1);
  for (int n = 0; n < N; ++n) {
    for (int c = 0; c < N; ++c) {
      c128 *p = W + idx * (N * N) + n * N + c;
      *p = cmul(*p, z[n]);
      if (G) G[idx * (N * N) + n * N + c] = (c == n) ? z[n] : cmake(0.0, 0.0);
    }
  }
  if (!ok && info) atomicAdd(info, 1);
}

// G = diag(s), s_n = sum_c XY[ref][c] (YY^-1)[c][n]: the row XY[ref, :] YY^-1 = solution of YY^T s = XY[ref]^T
__global__ __launch_bounds__(64) void k_pb_scale_rt(const c128 *__restrict__ XY,
                                                    const c128 *__restrict__ YY, c128 *G,
                                                    long long nbins, int N, int ref, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 A[RTN * RTN], s[RTN];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) A[r * N + c] = YY[idx * (N * N) + c * N + r];
  for (int c = 0; c < N; ++c) s[c] = XY[idx * (N * N) + ref * N + c];
  const bool ok = rt_lu_solve(A, s, N, 1);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < N; ++c) G[idx * (N * N) + n * N + c] = (c == n) ? s[n] : cmake(0.0, 0.0);
  if (!ok && info) atomicAdd(info, 1);
}

// W = YX XX^-1: row r of W solves XX^T w = YX[r]^T; all rows at once as N right-hand sides
__global__ __launch_bounds__(64) void k_demix_from_cov_rt(const c128 *__restrict__ YX,
                                                          const c128 *__restrict__ XX, c128 *W,
                                                          long long nbins, int N, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 A[RTN * RTN], R[RTN * RTN];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      A[r * N + c] = XX[idx * (N * N) + c * N + r];
      R[r * N + c] = YX[idx * (N * N) + c * N + r];  // column c of R = row c of YX
    }
  const bool ok = rt_lu_solve(A, R, N, N);
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) W[idx * (N * N) + r * N + c] = R[c * N + r];
  if (!ok && info) atomicAdd(info, 1);
}

// one block per mixture
__global__ __launch_bounds__(256) void k_sum_logdet_rt(const c128 *__restrict__ W, double *out,
                                                       int F, int N) {
  __shared__ double scratch[4];
  const int b = blockIdx.x;
  double s = 0.0;
  c128 A[RTN * RTN];
  for (int i = threadIdx.x; i < F; i += blockDim.x) {
    for (int e = 0; e < N * N; ++e) A[e] = W[((long long)b * F + i) * (N * N) + e];
    s += rt_logabsdet(A, N);
  }
  const double total = block_sum(s, scratch);
  if (threadIdx.x == 0) out[b] = total;
}

// ---- frame powers r2[b, n, j] = sum_i |y_nij|^2 (y = W x, or x itself when W is NULL) ------------
// grid (ceil(T / 256), bin chunks, B), a thread per frame; slab layout of ssspy_iva_frame_power
__global__ __launch_bounds__(256) void k_frame_power_rt(const c128 *__restrict__ X,
                                                        const c128 *__restrict__ W, double *r2,
                                                        int N, int F, int T, int bins_per_chunk) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z;
  const int i_begin = blockIdx.y * bins_per_chunk;
  const int i_end = min(F, i_begin + bins_per_chunk);
  if (j >= T) return;
  double acc[RTN];
  c128 x[RTN];
  for (int n = 0; n < N; ++n) acc[n] = 0.0;
  for (int i = i_begin; i < i_end; ++i) {
    for (int m = 0; m < N; ++m) x[m] = X[(((long long)b * N + m) * F + i) * T + j];
    if (W) {
      const c128 *__restrict__ Wi = W + ((long long)b * F + i) * (N * N);
      for (int n = 0; n < N; ++n) {
        c128 y = cmake(0.0, 0.0);
        for (int m = 0; m < N; ++m) cfma(y, Wi[n * N + m], x[m]);
        acc[n] += cabs2(y);
      }
    } else {
      for (int n = 0; n < N; ++n) acc[n] += cabs2(x[n]);
    }
  }
  double *dst = r2 + (long long)blockIdx.y * gridDim.z * N * T;
  for (int n = 0; n < N; ++n) dst[((long long)b * N + n) * T + j] = acc[n];
}

// ---- ILRMA data term of the loss, Gauss model: sum_{n,i} mean_j ( |y|^2 / R^(2/p) + (2/p) log R ) ----
// R = (T V)_nij.  ref: ssspy/bss/ilrma.py:1946-1965.  grid (F, N, B), 256 threads along frames; one
// value per (mixture, source, bin) goes to `terms`, k_sum_terms_rt adds a mixture's N F values.
__global__ __launch_bounds__(256) void k_ilrma_loss_rt(const c128 *__restrict__ X,
                                                       const c128 *__restrict__ W,
                                                       const double *__restrict__ basis,
                                                       const double *__restrict__ act,
                                                       double *__restrict__ terms, int N, int F,
                                                       int T, int K, double p) {
  __shared__ double scratch[4];
  const int i = blockIdx.x, n = blockIdx.y, b = blockIdx.z;
  const double *__restrict__ tb = basis + (((long long)b * N + n) * F + i) * K;
  const double *__restrict__ vb = act + ((long long)b * N + n) * K * (long long)T;
  const c128 *__restrict__ w = W ? W + ((long long)b * F + i) * (N * N) + n * N : nullptr;
  double acc = 0.0;
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    c128 y;
    if (w) {
      y = cmake(0.0, 0.0);
      for (int m = 0; m < N; ++m) cfma(y, w[m], X[(((long long)b * N + m) * F + i) * T + j]);
    } else {
      y = X[(((long long)b * N + n) * F + i) * T + j];
    }
    double R = 0.0;
    for (int k = 0; k < K; ++k) R = fma(tb[k], vb[(long long)k * T + j], R);
    const double P = cabs2(y);
    acc += (p == 2.0 ? P / R : P / pow(R, 2.0 / p)) + (2.0 / p) * log(R);
  }
  const double total = block_sum(acc, scratch);
  if (threadIdx.x == 0) terms[((long long)b * N + n) * F + i] = total / (double)T;
}

// out[b] = sum of the mixture's `count` terms (one block per mixture, fixed order)
__global__ __launch_bounds__(256) void k_sum_terms_rt(const double *__restrict__ terms, double *out,
                                                      long long count) {
  __shared__ double scratch[4];
  const int b = blockIdx.x;
  double s = 0.0;
  for (long long e = threadIdx.x; e < count; e += blockDim.x) s += terms[b * count + e];
  const double total = block_sum(s, scratch);
  if (threadIdx.x == 0) out[b] = total;
}

// ---- launchers (called by the entry points of the other units when N > SSSPY_MAX_SOURCES) --------
bool rt_sources_ok(int N) { return N > SSSPY_MAX_SOURCES && N <= SSSPY_RT_MAX_SOURCES; }

int rt_separate(const void *X, const void *W, void *Y, int B, int N, int F, int T, bool power,
                hipStream_t st) {
  dim3 grid(F, B), block(256);
  if (power)
    hipLaunchKernelGGL(k_separate_rt<true>, grid, block, 0, st, (const c128 *)X, (const c128 *)W,
                       (c128 *)Y, N, F, T);
  else
    hipLaunchKernelGGL(k_separate_rt<false>, grid, block, 0, st, (const c128 *)X, (const c128 *)W,
                       (c128 *)Y, N, F, T);
  return check_launch("k_separate_rt");
}

int rt_covariance(const void *A, const void *Bm, const double *weight, int kind, void *C, int B,
                  int N, int S, int F, int T, hipStream_t st) {
  hipLaunchKernelGGL(k_cov_rt, dim3(F, S, B), dim3(256), 0, st, (const c128 *)A, (const c128 *)Bm,
                     weight, kind, (c128 *)C, S, N, F, T);
  return check_launch("k_cov_rt");
}

int rt_ip1(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N, int floor_kind,
           double floor_eps, int *info, hipStream_t st) {
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_ip1_rt, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st, (c128 *)W,
                     (const c128 *)U, nbins, N, floor_kind, floor_eps, info, (const c128 *)C, qbuf);
  return check_launch("k_ip1_rt");
}

int rt_row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st) {
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_row_power_rt, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                     (const c128 *)W, (const c128 *)C, qbuf, nbins, N);
  return check_launch("k_row_power_rt");
}

int rt_iss1_transform(const void *Vc, void *G, int B, int F, int N, int floor_kind, double floor_eps,
                      hipStream_t st) {
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_iss1_transform_rt, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                     (const c128 *)Vc, (c128 *)G, nbins, N, floor_kind, floor_eps);
  return check_launch("k_iss1_transform_rt");
}

int rt_pb_filter(void *W, void *G, int B, int F, int N, int ref, int *info, hipStream_t st) {
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_pb_filter_rt, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                     (c128 *)W, (c128 *)G, nbins, N, ref, info);
  return check_launch("k_pb_filter_rt");
}

int rt_pb_scale(const void *XY, const void *YY, void *G, int B, int F, int N, int ref, int *info,
                hipStream_t st) {
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_pb_scale_rt, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                     (const c128 *)XY, (const c128 *)YY, (c128 *)G, nbins, N, ref, info);
  return check_launch("k_pb_scale_rt");
}

int rt_demix_from_cov(const void *YX, const void *XX, void *W, int B, int F, int N, int *info,
                      hipStream_t st) {
  const long long nbins = (long long)B * F;
  hipLaunchKernelGGL(k_demix_from_cov_rt, dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                     (const c128 *)YX, (const c128 *)XX, (c128 *)W, nbins, N, info);
  return check_launch("k_demix_from_cov_rt");
}

int rt_sum_logdet(const void *W, double *out, int B, int F, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_logdet_rt, dim3(B), dim3(256), 0, st, (const c128 *)W, out, F, N);
  return check_launch("k_sum_logdet_rt");
}

size_t rt_ilrma_loss_ws_bytes(int B, int N, int F) { return (size_t)B * N * F * sizeof(double); }

int rt_ilrma_loss(const void *X, const void *W, const double *basis, const double *act, double *out,
                  void *ws, int B, int N, int F, int T, int K, double domain, hipStream_t st) {
  hipLaunchKernelGGL(k_ilrma_loss_rt, dim3(F, N, B), dim3(256), 0, st, (const c128 *)X,
                     (const c128 *)W, basis, act, (double *)ws, N, F, T, K, domain);
  int rc = check_launch("k_ilrma_loss_rt");
  if (rc) return rc;
  hipLaunchKernelGGL(k_sum_terms_rt, dim3(B), dim3(256), 0, st, (const double *)ws, out,
                     (long long)N * F);
  return check_launch("k_sum_terms_rt");
}

int rt_frame_power(const void *X, const void *W, double *dst, int B, int N, int F, int T,
                   int bins_per_chunk, int chunks, hipStream_t st) {
  hipLaunchKernelGGL(k_frame_power_rt, dim3((T + 255) / 256, chunks, B), dim3(256), 0, st,
                     (const c128 *)X, (const c128 *)W, dst, N, F, T, bins_per_chunk);
  return check_launch("k_frame_power_rt");
}

}  // namespace ssspy
