// Hermitian matrix functions of ssspy.linalg on the device, one matrix per lane (M <= 8):
// generalised eigen-decomposition (types 1-3), square root / inverse square root, geometric mean.
//
// replaces: ssspy/linalg/eigh.py:8-81, :164-207 (eigh with B, _eigh), ssspy/linalg/sqrtm.py:8-64
//           (sqrtmh, invsqrtmh), ssspy/linalg/mean.py:6-83 (gmeanmh).
#include "common.hpp"
#include "hermitian.hpp"
#include "ssspy_amd.h"

namespace ssspy {

// hermitian_rows.hip: the same operators with a matrix on 8 lanes (6 x 6 .. 8 x 8)
bool hermitian_rows_wanted(int M, int always_from);
int sqrtmh_rows(const void *X, void *out, long long n, int M, int mode, int floor_kind, double eps,
                hipStream_t st);
int gmeanmh_rows(const void *A, const void *Bm, void *G, long long n, int M, int type,
                 hipStream_t st);
int eigh_general_rows(const void *A, const void *Bm, double *lamb, void *Z, long long n, int M,
                      int type, int *info, hipStream_t st);

// hermitian_rt.hip: 9 x 9 .. 16 x 16, the size at run time
bool hermitian_rt_wanted(int M);
int eigh_general_rt(const void *A, const void *Bm, double *lamb, void *Z, long long n, int M,
                    int type, int *info, hipStream_t st);
int sqrtmh_rt(const void *X, void *out, long long n, int M, int mode, int floor_kind, double eps,
              hipStream_t st);
int gmeanmh_rt(const void *A, const void *Bm, void *G, long long n, int M, int type,
               hipStream_t st);

// lower Cholesky factor of a Hermitian positive definite matrix, in place (upper part zeroed)
template <int M>
__device__ __forceinline__ bool cholesky_lower(c128 (&A)[M][M]) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < M; ++c) {
    double d = A[c][c].x;
#pragma unroll
    for (int k = 0; k < c; ++k) d -= cabs2(A[c][k]);
    ok = ok && (d > 0.0);
    const double l = sqrt(d > 0.0 ? d : 1.0), il = 1.0 / l;
    A[c][c] = cmake(l, 0.0);
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      c128 s = A[r][c];
#pragma unroll
      for (int k = 0; k < c; ++k) cfms(s, A[r][k], cconj(A[c][k]));
      A[r][c] = cscale(s, il);
    }
#pragma unroll
    for (int r = 0; r < c; ++r) A[r][c] = cmake(0.0, 0.0);
  }
  return ok;
}

// inverse of a lower triangular matrix (lower triangular)
template <int M>
__device__ __forceinline__ void lower_inverse(const c128 (&L)[M][M], c128 (&Li)[M][M]) {
#pragma unroll
  for (int c = 0; c < M; ++c) {
#pragma unroll
    for (int r = 0; r < M; ++r) Li[r][c] = cmake(0.0, 0.0);
    Li[c][c] = crecip(L[c][c]);
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int k = c; k < r; ++k) cfms(s, L[r][k], Li[k][c]);
      Li[r][c] = cmul(s, crecip(L[r][r]));
    }
  }
}

template <int M>
__device__ __forceinline__ void conj_transpose(const c128 (&A)[M][M], c128 (&At)[M][M]) {
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) At[r][c] = cconj(A[c][r]);
}

// generalised Hermitian eigenproblem through the Cholesky factor of B (ref: eigh.py:164-207):
//   type 1: A z = lamb B z   (C = L^-1 A L^-H, z = L^-H y)
//   type 2: A B z = lamb z   (C = L^H A L,    z = L^-H y)
//   type 3: B A z = lamb z   (C = L^H A L,    z = L y)
// eigenvalues ascending; eigenvectors carry the decomposition's phase (as with LAPACK, arbitrary).
template <int M>
__global__ __launch_bounds__(64) void k_eigh_general(const c128 *__restrict__ A,
                                                     const c128 *__restrict__ Bm, double *lamb,
                                                     c128 *Z, long long n, int type, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  c128 Am[M][M], L[M][M], T1[M][M], T2[M][M], P[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      Am[r][c] = A[(idx * M + r) * M + c];
      L[r][c] = Bm[(idx * M + r) * M + c];
    }
  const bool ok = cholesky_lower<M>(L);
  if (!ok && info) atomicAdd(info, 1);
  c128 Li[M][M], LiH[M][M], LH[M][M];
  lower_inverse<M>(L, Li);
  conj_transpose<M>(Li, LiH);
  conj_transpose<M>(L, LH);
  if (type == 1) {
    matmul<M>(Li, Am, T1);
    matmul<M>(T1, LiH, T2);
  } else {
    matmul<M>(LH, Am, T1);
    matmul<M>(T1, L, T2);
  }
  hermitize<M>(T2);
  jacobi_eigh<M>(T2, P);
  c128 Zm[M][M];
  if (type == 3)
    matmul<M>(L, P, Zm);
  else
    matmul<M>(LiH, P, Zm);
#pragma unroll
  for (int k = 0; k < M; ++k) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < M; ++j)
      rank += (T2[j][j].x < T2[k][k].x || (T2[j][j].x == T2[k][k].x && j < k)) ? 1 : 0;
    lamb[idx * M + rank] = T2[k][k].x;
#pragma unroll
    for (int r = 0; r < M; ++r) Z[(idx * M + r) * M + rank] = Zm[r][k];
  }
}

// mode 0: X^(1/2); mode 1: P diag(1 / floor(sqrt(lam))) P^H  (ref: sqrtm.py:8-64)
template <int M>
__global__ __launch_bounds__(64) void k_sqrtmh(const c128 *__restrict__ X, c128 *out, long long n,
                                               int mode, int floor_kind, double eps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  c128 Am[M][M], P[M][M], R[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) Am[r][c] = X[(idx * M + r) * M + c];
  hermitize<M>(Am);
  jacobi_eigh<M>(Am, P);
  double w[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const double s = sqrt(Am[k][k].x);  // NaN for a negative eigenvalue, as numpy.sqrt
    w[k] = mode == 0 ? s : 1.0 / apply_floor(s, floor_kind, eps);
  }
  herm_rebuild<M>(P, w, R);
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) out[(idx * M + r) * M + c] = R[r][c];
}

// geometric mean through Hermitian square roots: X # Y = X^1/2 (X^-1/2 Y X^-1/2)^1/2 X^1/2 with
//   type 1: A # B;  type 2: A^-1 # B;  type 3: A # B^-1   (ref: mean.py:6-83)
template <int M>
__global__ __launch_bounds__(64) void k_gmeanmh(const c128 *__restrict__ A,
                                                const c128 *__restrict__ Bm, c128 *G, long long n,
                                                int type) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  c128 Xm[M][M], Ym[M][M], P[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) {
      const c128 a = A[(idx * M + r) * M + c], b = Bm[(idx * M + r) * M + c];
      // the matrix whose square roots frame the mean, and the one in the middle
      Xm[r][c] = type == 3 ? b : a;
      Ym[r][c] = type == 3 ? a : b;
    }
  hermitize<M>(Xm);
  hermitize<M>(Ym);
  jacobi_eigh<M>(Xm, P);
  // type 1: outer = X^1/2, inner = X^-1/2;  types 2, 3 (mean with an inverse): the roles swap
  double wo[M], wi[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const double s = sqrt(Xm[k][k].x);
    wo[k] = type == 1 ? s : 1.0 / s;
    wi[k] = type == 1 ? 1.0 / s : s;
  }
  c128 Out[M][M], In[M][M], T1[M][M], C[M][M];
  herm_rebuild<M>(P, wo, Out);
  herm_rebuild<M>(P, wi, In);
  matmul<M>(In, Ym, T1);
  matmul<M>(T1, In, C);
  hermitize<M>(C);
  jacobi_eigh<M>(C, P);
  double w[M];
#pragma unroll
  for (int k = 0; k < M; ++k) w[k] = sqrt(fmax(C[k][k].x, 0.0));
  herm_rebuild<M>(P, w, Ym);
  matmul<M>(Out, Ym, T1);
  matmul<M>(T1, Out, C);
  hermitize<M>(C);
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) G[(idx * M + r) * M + c] = C[r][c];
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

int ssspy_eigh_general(const void *A, const void *Bm, double *lamb, void *Z, long long n, int M,
                       int type, int *info, void *stream) {
  SSSPY_REQUIRE(A && Bm && lamb && Z && n > 0, "eigh_general: bad argument");
  SSSPY_REQUIRE(type >= 1 && type <= 3, "eigh_general: type must be 1, 2 or 3");
  if (hermitian_rt_wanted(M))
    return eigh_general_rt(A, Bm, lamb, Z, n, M, type, info, as_stream(stream));
  if (hermitian_rows_wanted(M, 7))
    return eigh_general_rows(A, Bm, lamb, Z, n, M, type, info, as_stream(stream));
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  DISPATCH_N6(M, hipLaunchKernelGGL((k_eigh_general<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)A, (const c128 *)Bm, lamb, (c128 *)Z, n, type,
                                   info));
  return check_launch("k_eigh_general");
}

int ssspy_sqrtmh(const void *X, void *out, long long n, int M, int inverse, int floor_kind,
                 double floor_eps, void *stream) {
  SSSPY_REQUIRE(X && out && n > 0, "sqrtmh: bad argument");
  if (hermitian_rt_wanted(M))
    return sqrtmh_rt(X, out, n, M, inverse ? 1 : 0, floor_kind, floor_eps, as_stream(stream));
  if (hermitian_rows_wanted(M, 7))
    return sqrtmh_rows(X, out, n, M, inverse ? 1 : 0, floor_kind, floor_eps, as_stream(stream));
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  DISPATCH_N6(M, hipLaunchKernelGGL((k_sqrtmh<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)X, (c128 *)out, n, inverse ? 1 : 0, floor_kind,
                                   floor_eps));
  return check_launch("k_sqrtmh");
}

int ssspy_gmeanmh(const void *A, const void *Bm, void *G, long long n, int M, int type,
                  void *stream) {
  SSSPY_REQUIRE(A && Bm && G && n > 0, "gmeanmh: bad argument");
  SSSPY_REQUIRE(type >= 1 && type <= 3, "gmeanmh: type must be 1, 2 or 3");
  if (hermitian_rt_wanted(M)) return gmeanmh_rt(A, Bm, G, n, M, type, as_stream(stream));
  if (hermitian_rows_wanted(M, 7)) return gmeanmh_rows(A, Bm, G, n, M, type, as_stream(stream));
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  DISPATCH_N6(M, hipLaunchKernelGGL((k_gmeanmh<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)A, (const c128 *)Bm, (c128 *)G, n, type));
  return check_launch("k_gmeanmh");
}

}  // extern "C"
