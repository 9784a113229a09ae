// The Hermitian operators and `solve` with the matrix size at run time (9 <= M <= SSSPY_RT_MAX_SOURCES
// = 16): the reference's ssspy.linalg functions take any size (they are numpy.linalg calls), the
// templated kernels stop at 8 x 8 (linalg_kernels.hip, hermitian_ops.hip, hermitian_rows.hip).
// Correct, not tuned: one lane owns one matrix and keeps its M x M working set in private (scratch)
// memory with run-time loops (rt_hermitian.hpp); the statements are those of the templated kernels,
// one by one -- the same cyclic Jacobi, the same eigenvalue order, the same floors.
//
// replaces (for M > 8): ssspy/linalg/_solve.py:9-21, ssspy/linalg/eigh.py:8-81, :164-207,
//   ssspy/linalg/sqrtm.py:8-64, ssspy/linalg/mean.py:6-83, ssspy/special/psd.py:11-71.
#include "common.hpp"
#include "rt_hermitian.hpp"
#include "ssspy_amd.h"

namespace ssspy {

namespace {

constexpr int RM = SSSPY_RT_MAX_SOURCES;

// a lane past the end works on the last matrix (every lane is in the wave votes of rt_jacobi) and
// writes nothing
struct Lane {
  long long idx;
  bool live;
};
__device__ __forceinline__ Lane lane_of(long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  return Lane{i < n ? i : n - 1, i < n};
}

__device__ inline void rt_load(c128 *dst, const c128 *__restrict__ src, long long idx, int M) {
  for (int e = 0; e < M * M; ++e) dst[e] = src[idx * M * M + e];
}
__device__ inline void rt_store(c128 *dst, const c128 *src, long long idx, int M) {
  for (int e = 0; e < M * M; ++e) dst[idx * M * M + e] = src[e];
}

// ascending eigenvalues by rank (ties by index) and the eigenvector columns in that order
__device__ inline void rt_store_sorted(const c128 *D, const c128 *P, double *lamb, c128 *V,
                                       long long idx, int M) {
  for (int k = 0; k < M; ++k) {
    const double lk = D[k * M + k].x;
    int rank = 0;
    for (int j = 0; j < M; ++j) {
      const double lj = D[j * M + j].x;
      rank += (lj < lk || (lj == lk && j < k)) ? 1 : 0;
    }
    lamb[idx * M + rank] = lk;
    for (int r = 0; r < M; ++r) V[(idx * M + r) * M + rank] = P[r * M + k];
  }
}

// lower Cholesky factor in place, upper part zeroed (cholesky_lower, hermitian_ops.hip)
__device__ inline bool rt_cholesky_lower(c128 *A, int M) {
  bool ok = true;
  for (int c = 0; c < M; ++c) {
    double d = A[c * M + c].x;
    for (int k = 0; k < c; ++k) d -= cabs2(A[c * M + k]);
    ok = ok && (d > 0.0);
    const double l = sqrt(d > 0.0 ? d : 1.0), il = 1.0 / l;
    A[c * M + c] = cmake(l, 0.0);
    for (int r = c + 1; r < M; ++r) {
      c128 s = A[r * M + c];
      for (int k = 0; k < c; ++k) cfms(s, A[r * M + k], cconj(A[c * M + k]));
      A[r * M + c] = cscale(s, il);
    }
    for (int r = 0; r < c; ++r) A[r * M + c] = cmake(0.0, 0.0);
  }
  return ok;
}

// inverse of a lower triangular matrix (lower_inverse, hermitian_ops.hip)
__device__ inline void rt_lower_inverse(const c128 *L, c128 *Li, int M) {
  for (int c = 0; c < M; ++c) {
    for (int r = 0; r < M; ++r) Li[r * M + c] = cmake(0.0, 0.0);
    Li[c * M + c] = crecip(L[c * M + c]);
    for (int r = c + 1; r < M; ++r) {
      c128 s = cmake(0.0, 0.0);
      for (int k = c; k < r; ++k) cfms(s, L[r * M + k], Li[k * M + c]);
      Li[r * M + c] = cmul(s, crecip(L[r * M + r]));
    }
  }
}

// X = A^-1 B by LU with partial pivoting; the right-hand sides pass through in chunks of M columns
__global__ __launch_bounds__(64) void k_solve_rt(const c128 *__restrict__ A,
                                                 const c128 *__restrict__ Bm, c128 *X, long long n,
                                                 int M, int nrhs, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  c128 Am[RM * RM], Inv[RM * RM];
  rt_load(Am, A, idx, M);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < M; ++c) Inv[r * M + c] = cmake(r == c ? 1.0 : 0.0, 0.0);
  const bool ok = rt_lu_solve(Am, Inv, M, M);
  for (int c = 0; c < nrhs; ++c)
    for (int r = 0; r < M; ++r) {
      c128 acc = cmake(0.0, 0.0);
      for (int k = 0; k < M; ++k) cfma(acc, Inv[r * M + k], Bm[(idx * M + k) * nrhs + c]);
      X[(idx * M + r) * nrhs + c] = acc;
    }
  if (!ok && info) atomicAdd(info, 1);
}

// mode 0: eigh (lamb, V);  mode 1: to_psd into V  (k_eigh, linalg_kernels.hip)
__global__ __launch_bounds__(64) void k_eigh_rt(const c128 *__restrict__ A, double *lamb, c128 *V,
                                                long long n, int M, int mode, int floor_kind,
                                                double eps) {
  const Lane ln = lane_of(n);
  c128 Am[RM * RM], P[RM * RM];
  rt_load(Am, A, ln.idx, M);
  rt_hermitize(Am, M);
  rt_jacobi(Am, P, M);
  if (!ln.live) return;
  if (mode == 0) {
    rt_store_sorted(Am, P, lamb, V, ln.idx, M);
  } else {
    double w[RM];
    for (int k = 0; k < M; ++k) w[k] = apply_floor(Am[k * M + k].x, floor_kind, eps);
    rt_rebuild(P, w, Am, M);
    rt_store(V, Am, ln.idx, M);
  }
}

// generalised problem through the Cholesky factor of B (k_eigh_general, hermitian_ops.hip)
__global__ __launch_bounds__(64) void k_eigh_general_rt(const c128 *__restrict__ A,
                                                        const c128 *__restrict__ Bm, double *lamb,
                                                        c128 *Z, long long n, int M, int type,
                                                        int *info) {
  const Lane ln = lane_of(n);
  c128 Am[RM * RM], L[RM * RM], Li[RM * RM], T[RM * RM], P[RM * RM];
  rt_load(Am, A, ln.idx, M);
  rt_load(L, Bm, ln.idx, M);
  const bool ok = rt_cholesky_lower(L, M);
  if (!ok && info && ln.live) atomicAdd(info, 1);
  rt_lower_inverse(L, Li, M);
  if (type == 1) {
    rt_matmul(Li, Am, T, M);      // L^-1 A
    rt_matmul_h(T, Li, Am, M);    // (L^-1 A) L^-H
  } else {
    rt_matmul_hl(L, Am, T, M);    // L^H A
    rt_matmul(T, L, Am, M);       // (L^H A) L
  }
  rt_hermitize(Am, M);
  rt_jacobi(Am, P, M);
  if (type == 3)
    rt_matmul(L, P, T, M);
  else
    rt_matmul_hl(Li, P, T, M);    // L^-H P
  if (ln.live) rt_store_sorted(Am, T, lamb, Z, ln.idx, M);
}

// mode 0: X^(1/2); mode 1: P diag(1 / floor(sqrt(lam))) P^H  (k_sqrtmh, hermitian_ops.hip)
__global__ __launch_bounds__(64) void k_sqrtmh_rt(const c128 *__restrict__ X, c128 *out, long long n,
                                                  int M, int mode, int floor_kind, double eps) {
  const Lane ln = lane_of(n);
  c128 Am[RM * RM], P[RM * RM];
  rt_load(Am, X, ln.idx, M);
  rt_hermitize(Am, M);
  rt_jacobi(Am, P, M);
  double w[RM];
  for (int k = 0; k < M; ++k) {
    const double s = sqrt(Am[k * M + k].x);  // NaN for a negative eigenvalue, as numpy.sqrt
    w[k] = mode == 0 ? s : 1.0 / apply_floor(s, floor_kind, eps);
  }
  rt_rebuild(P, w, Am, M);
  if (ln.live) rt_store(out, Am, ln.idx, M);
}

// geometric mean through Hermitian square roots (k_gmeanmh, hermitian_ops.hip)
__global__ __launch_bounds__(64) void k_gmeanmh_rt(const c128 *__restrict__ A,
                                                   const c128 *__restrict__ Bm, c128 *G, long long n,
                                                   int M, int type) {
  const Lane ln = lane_of(n);
  c128 Xm[RM * RM], Ym[RM * RM], P[RM * RM], Out[RM * RM], In[RM * RM];
  // the matrix whose square roots frame the mean, and the one in the middle
  rt_load(Xm, type == 3 ? Bm : A, ln.idx, M);
  rt_load(Ym, type == 3 ? A : Bm, ln.idx, M);
  rt_hermitize(Xm, M);
  rt_hermitize(Ym, M);
  rt_jacobi(Xm, P, M);
  double wo[RM], wi[RM];
  for (int k = 0; k < M; ++k) {
    const double s = sqrt(Xm[k * M + k].x);
    wo[k] = type == 1 ? s : 1.0 / s;
    wi[k] = type == 1 ? 1.0 / s : s;
  }
  rt_rebuild(P, wo, Out, M);
  rt_rebuild(P, wi, In, M);
  rt_matmul(In, Ym, Xm, M);
  rt_matmul(Xm, In, Ym, M);
  rt_hermitize(Ym, M);
  rt_jacobi(Ym, P, M);
  for (int k = 0; k < M; ++k) wo[k] = sqrt(fmax(Ym[k * M + k].x, 0.0));
  rt_rebuild(P, wo, In, M);
  rt_matmul(Out, In, Xm, M);
  rt_matmul(Xm, Out, Ym, M);
  rt_hermitize(Ym, M);
  if (ln.live) rt_store(G, Ym, ln.idx, M);
}

inline dim3 lanes(long long n) { return dim3((unsigned)((n + 63) / 64)); }

}  // namespace

bool hermitian_rt_wanted(int M) { return M > SSSPY_MAX_SOURCES && M <= SSSPY_RT_MAX_SOURCES; }

int solve_rt(const void *A, const void *Bm, void *X, long long n, int M, int nrhs, int *info,
             hipStream_t st) {
  hipLaunchKernelGGL(k_solve_rt, lanes(n), dim3(64), 0, st, (const c128 *)A, (const c128 *)Bm,
                     (c128 *)X, n, M, nrhs, info);
  return check_launch("k_solve_rt");
}

int eigh_rt(const void *A, double *lamb, void *V, long long n, int M, int mode, int floor_kind,
            double eps, hipStream_t st) {
  hipLaunchKernelGGL(k_eigh_rt, lanes(n), dim3(64), 0, st, (const c128 *)A, lamb, (c128 *)V, n, M,
                     mode, floor_kind, eps);
  return check_launch("k_eigh_rt");
}

int eigh_general_rt(const void *A, const void *Bm, double *lamb, void *Z, long long n, int M,
                    int type, int *info, hipStream_t st) {
  hipLaunchKernelGGL(k_eigh_general_rt, lanes(n), dim3(64), 0, st, (const c128 *)A,
                     (const c128 *)Bm, lamb, (c128 *)Z, n, M, type, info);
  return check_launch("k_eigh_general_rt");
}

int sqrtmh_rt(const void *X, void *out, long long n, int M, int mode, int floor_kind, double eps,
              hipStream_t st) {
  hipLaunchKernelGGL(k_sqrtmh_rt, lanes(n), dim3(64), 0, st, (const c128 *)X, (c128 *)out, n, M,
                     mode, floor_kind, eps);
  return check_launch("k_sqrtmh_rt");
}

int gmeanmh_rt(const void *A, const void *Bm, void *G, long long n, int M, int type,
               hipStream_t st) {
  hipLaunchKernelGGL(k_gmeanmh_rt, lanes(n), dim3(64), 0, st, (const c128 *)A, (const c128 *)Bm,
                     (c128 *)G, n, M, type);
  return check_launch("k_gmeanmh_rt");
}

}  // namespace ssspy
