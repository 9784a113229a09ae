// Hermitian matrix functions of ssspy.linalg for 6 x 6 .. 8 x 8 with a matrix on 8 lanes (round 5):
// eigh / to_psd, the generalised eigenproblem (types 1-3), sqrtmh / invsqrtmh, gmeanmh.  The
// lane-per-matrix kernels of linalg_kernels.hip / hermitian_ops.hip spill 1 000-3 800 VGPRs at
// these sizes (profiles/r04_kernel_resources.txt); here a row per lane (herm_rows8.hpp), no scratch.
//
// replaces: ssspy/linalg/eigh.py:8-81, :164-207, ssspy/linalg/sqrtm.py:8-64, ssspy/linalg/mean.py:6-83,
//           ssspy/special/psd.py:11-71.
// Same definitions as the lane-per-matrix kernels (Hermitian part of the input, eigenvalues ascending
// with ties by index, eigenvectors with the decomposition's own phase); the rotations come in the
// round-robin order instead of row by row, so results agree to rounding, not bit for bit.
#include <cstdlib>

#include "herm_rows8.hpp"
#include "ssspy_amd.h"

namespace ssspy {

using namespace rows8;

namespace {

constexpr int MATS = 32;  // matrices per 256-thread block

struct Where {
  int r;          // my row
  c128 *X;        // my matrix's exchange slot
  long long idx;  // my matrix
  bool live;
};

__device__ __forceinline__ Where where(c128 *slots, long long n) {
  Where w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  w.r = lane & 7;
  const int g = wave * 8 + (lane >> 3);
  w.X = slots + g * SLOT;
  const long long idx = (long long)blockIdx.x * MATS + g;
  w.live = idx < n;
  w.idx = w.live ? idx : n - 1;  // idle groups shadow the last matrix and never store
  return w;
}

// row r of the Hermitian part of the M x M matrix at src, padded with zeros
__device__ __forceinline__ void load_herm_row(const c128 *__restrict__ src, int r, int M,
                                              c128 (&row)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    c128 v = cmake(0.0, 0.0);
    if (r < M && c < M) {
      const c128 u = src[r * M + c], l = src[c * M + r];
      v = cmake(0.5 * (u.x + l.x), 0.5 * (u.y - l.y));
    }
    row[c] = v;
  }
}

// row <- row r of (Z + Z^H) / 2 for the matrix whose rows the lanes hold (slot used)
__device__ __forceinline__ void hermitize_rows(c128 (&row)[8], c128 *X, int r) {
  wsync();
  store_row(X, r, row);
  wsync();
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const c128 t = X[c * LD + r];
    row[c] = cmake(0.5 * (row[c].x + t.x), 0.5 * (row[c].y - t.y));
  }
  wsync();
}

// eigen-decomposition of the Hermitian matrix with row r in `a`: lam_r in dg, row r of the
// eigenvector matrix (or of W0 J) in w
__device__ __forceinline__ void eig_rows(c128 (&a)[8], double &dg, c128 (&w)[8], int r) {
  dg = sel(a, r).x;
  jacobi<true>(a, dg, w, r);
}

__device__ __forceinline__ void identity_row(c128 (&w)[8], int r) {
#pragma unroll
  for (int c = 0; c < 8; ++c) w[c] = cmake(c == r ? 1.0 : 0.0, 0.0);
}

// position of eigenvalue r in ascending order (ties by index) among the first M
__device__ __forceinline__ int ascending_rank(double lam, int r, int M) {
  int rank = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double lj = shfl8(lam, j);
    rank += (j < M && (lj < lam || (lj == lam && j < r))) ? 1 : 0;
  }
  return rank;
}

__device__ __forceinline__ void store_sorted(double *lamb, c128 *V, long long idx, int M, double lam,
                                             const c128 (&w)[8], int r, bool live) {
  const int rank = ascending_rank(lam, r, M);
  if (live && r < M) lamb[idx * M + rank] = lam;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int rk = __shfl(rank, k, 8);
    if (live && r < M && k < M) V[(idx * M + r) * M + rk] = w[k];
  }
}

__device__ __forceinline__ void store_rows(c128 *out, long long idx, int M, const c128 (&row)[8],
                                           int r, bool live) {
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (live && r < M && c < M) out[(idx * M + r) * M + c] = row[c];
}

// mode 0: eigenvalues (n, M) ascending and eigenvectors (n, M, M); mode 1 (to_psd): floor, rebuild
__global__ __launch_bounds__(256, 2) void k_eigh_rows(const c128 *__restrict__ A, double *lamb, c128 *V,
                                                   long long n, int M, int mode, int floor_kind,
                                                   double eps) {
  __shared__ c128 slots[MATS * SLOT];
  const Where q = where(slots, n);
  c128 a[8], w[8];
  load_herm_row(A + q.idx * M * M, q.r, M, a);
  identity_row(w, q.r);
  double lam;
  eig_rows(a, lam, w, q.r);
  if (mode == 0) {
    store_sorted(lamb, V, q.idx, M, lam, w, q.r, q.live);
  } else {
    c128 out[8];
    rebuild(w, q.r < M ? apply_floor(lam, floor_kind, eps) : 0.0, q.X, q.r, out);
    store_rows(V, q.idx, M, out, q.r, q.live);
  }
}

// mode 0: X^(1/2); mode 1: P diag(1 / floor(sqrt(lam))) P^H
__global__ __launch_bounds__(256, 2) void k_sqrtmh_rows(const c128 *__restrict__ Xin, c128 *out,
                                                     long long n, int M, int mode, int floor_kind,
                                                     double eps) {
  __shared__ c128 slots[MATS * SLOT];
  const Where q = where(slots, n);
  c128 a[8], w[8];
  load_herm_row(Xin + q.idx * M * M, q.r, M, a);
  identity_row(w, q.r);
  double lam;
  eig_rows(a, lam, w, q.r);
  const double s = sqrt(lam);  // NaN for a negative eigenvalue, as numpy.sqrt
  const double f = mode == 0 ? s : 1.0 / apply_floor(s, floor_kind, eps);
  c128 res[8];
  rebuild(w, q.r < M ? f : 0.0, q.X, q.r, res);
  store_rows(out, q.idx, M, res, q.r, q.live);
}

// row r of L R for the matrices whose rows the lanes hold: `left` my row of L, R goes through the slot
__device__ __forceinline__ void matmul_rows(const c128 (&left)[8], const c128 (&right)[8], c128 *X,
                                            int r, c128 (&out)[8]) {
  wsync();
  store_row(X, r, right);
  wsync();
  mul_rows(left, X, out);
  wsync();
}

// X # Y = X^1/2 (X^-1/2 Y X^-1/2)^1/2 X^1/2;  type 1: A # B, type 2: A^-1 # B, type 3: A # B^-1
__global__ __launch_bounds__(256, 2) void k_gmeanmh_rows(const c128 *__restrict__ A,
                                                      const c128 *__restrict__ Bm, c128 *G,
                                                      long long n, int M, int type) {
  __shared__ c128 slots[MATS * SLOT];
  const Where q = where(slots, n);
  const int r = q.r;
  c128 x[8], y[8], p[8];
  load_herm_row((type == 3 ? Bm : A) + q.idx * M * M, r, M, x);
  load_herm_row((type == 3 ? A : Bm) + q.idx * M * M, r, M, y);
  identity_row(p, r);
  double lam;
  eig_rows(x, lam, p, r);
  const double s = sqrt(lam);
  // type 1: outer = X^1/2, inner = X^-1/2;  types 2, 3 (mean with an inverse): the roles swap
  const double wo = r < M ? (type == 1 ? s : 1.0 / s) : 0.0;
  const double wi = r < M ? (type == 1 ? 1.0 / s : s) : 0.0;
  c128 outer[8], inner[8], t1[8], c[8];
  rebuild(p, wo, q.X, r, outer);
  rebuild(p, wi, q.X, r, inner);
  matmul_rows(inner, y, q.X, r, t1);
  matmul_rows(t1, inner, q.X, r, c);
  hermitize_rows(c, q.X, r);
  identity_row(p, r);
  eig_rows(c, lam, p, r);
  rebuild(p, r < M ? sqrt(fmax(lam, 0.0)) : 0.0, q.X, r, y);
  matmul_rows(outer, y, q.X, r, t1);
  matmul_rows(t1, outer, q.X, r, c);
  hermitize_rows(c, q.X, r);
  put(c, r, cmake(sel(c, r).x, 0.0));
  store_rows(G, q.idx, M, c, r, q.live);
}

// B = U^H U (L = U^H);  type 1: A z = lamb B z (C = U^-H A U^-1, z = U^-1 y);  type 2: A B z = lamb z
// (C = U A U^H, z = U^-1 y);  type 3: B A z = lamb z (C = U A U^H, z = U^H y)
__global__ __launch_bounds__(256, 2) void k_eigh_general_rows(const c128 *__restrict__ A,
                                                           const c128 *__restrict__ Bm, double *lamb,
                                                           c128 *Z, long long n, int M, int type,
                                                           int *info) {
  __shared__ c128 slots[MATS * SLOT];
  const Where q = where(slots, n);
  const int r = q.r;
  c128 a[8], u[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {  // (A as given, B as given: the lane-per-matrix kernel does the same)
    a[c] = (r < M && c < M) ? A[(q.idx * M + r) * M + c] : cmake(0.0, 0.0);
    // the factorisation reads the lower triangle of B (numpy.linalg.cholesky): row r of the
    // Hermitian matrix it stands for
    c128 b = cmake(c == r ? 1.0 : 0.0, 0.0);
    if (r < M && c < M) {
      const c128 lo = Bm[(q.idx * M + (r > c ? r : c)) * M + (r > c ? c : r)];
      b = r >= c ? lo : cconj(lo);
      if (r == c) b = cmake(lo.x, 0.0);
    }
    u[c] = b;
  }
  const bool ok = chol_upper(u, r);
  if (!ok && info && q.live && r == 0) atomicAdd(info, 1);
  wsync();
  store_row(q.X, r, u);
  wsync();
  c128 vcol[8], vrow[8];
  trtri_col(q.X, r, vcol);  // column r of V = U^-1
  wsync();
#pragma unroll
  for (int k = 0; k < 8; ++k) q.X[k * LD + r] = vcol[k];
  wsync();
#pragma unroll
  for (int c = 0; c < 8; ++c) vrow[c] = q.X[r * LD + c];
  wsync();
  c128 cm[8], w0[8];
  if (type == 1) {
    c128 t[8];
    matmul_rows(a, vrow, q.X, r, t);  // A V
    wsync();
    store_row(q.X, r, t);
    wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // (V^H T)[r][c] = sum_k conj(V[k][r]) T[k][c]
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const c128 tk = q.X[k * LD + c];
        s.x = fma(vcol[k].x, tk.x, s.x);
        s.x = fma(vcol[k].y, tk.y, s.x);
        s.y = fma(vcol[k].x, tk.y, s.y);
        s.y = fma(-vcol[k].y, tk.x, s.y);
      }
      cm[c] = s;
    }
    wsync();
  } else {
    c128 t[8];
    matmul_rows(u, a, q.X, r, t);  // U A
    wsync();
    store_row(q.X, r, u);
    wsync();
    mul_rows_adj(t, q.X, cm);  // (U A) U^H
    wsync();
  }
  if (type == 3) {  // rows of U^H: conj of column r of U
    wsync();
    store_row(q.X, r, u);
    wsync();
#pragma unroll
    for (int c = 0; c < 8; ++c) w0[c] = cconj(q.X[c * LD + r]);
    wsync();
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) w0[c] = vrow[c];
  }
  hermitize_rows(cm, q.X, r);
  double lam;
  eig_rows(cm, lam, w0, r);  // w0 <- (V or U^H) J = Z
  store_sorted(lamb, Z, q.idx, M, lam, w0, r, q.live);
}

}  // namespace

// 32 800 matrices, us per launch, 8 lanes | a lane per matrix at 6 / 7 / 8: eigh and to_psd 260 |
// 76 / 214 / 693; sqrtmh 262 | 83 / 341 / 680; gmeanmh 433 | 263 / 841 / 1 749; generalised eigh
// 232 | 106 / 382 / 872 (the padded 8 x 8 costs the same at every size).  So: always from
// `always_from` on (the lane-per-matrix instantiations above that are gone: 600-3 800 spilled
// VGPRs each).
bool hermitian_rows_wanted(int M, int always_from) { return M >= always_from && M <= 8; }

static dim3 rows_grid(long long n) { return dim3((unsigned)((n + MATS - 1) / MATS)); }

int eigh_rows(const void *A, double *lamb, void *V, long long n, int M, int mode, int floor_kind,
              double eps, hipStream_t st) {
  hipLaunchKernelGGL(k_eigh_rows, rows_grid(n), dim3(256), 0, st, (const c128 *)A, lamb, (c128 *)V, n,
                     M, mode, floor_kind, eps);
  return check_launch("k_eigh_rows");
}

int sqrtmh_rows(const void *X, void *out, long long n, int M, int mode, int floor_kind, double eps,
                hipStream_t st) {
  hipLaunchKernelGGL(k_sqrtmh_rows, rows_grid(n), dim3(256), 0, st, (const c128 *)X, (c128 *)out, n, M,
                     mode, floor_kind, eps);
  return check_launch("k_sqrtmh_rows");
}

int gmeanmh_rows(const void *A, const void *Bm, void *G, long long n, int M, int type,
                 hipStream_t st) {
  hipLaunchKernelGGL(k_gmeanmh_rows, rows_grid(n), dim3(256), 0, st, (const c128 *)A,
                     (const c128 *)Bm, (c128 *)G, n, M, type);
  return check_launch("k_gmeanmh_rows");
}

int eigh_general_rows(const void *A, const void *Bm, double *lamb, void *Z, long long n, int M,
                      int type, int *info, hipStream_t st) {
  hipLaunchKernelGGL(k_eigh_general_rows, rows_grid(n), dim3(256), 0, st, (const c128 *)A,
                     (const c128 *)Bm, lamb, (c128 *)Z, n, M, type, info);
  return check_launch("k_eigh_general_rows");
}

}  // namespace ssspy
