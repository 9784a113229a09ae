// Per-thread N x N complex128 linear algebra, fully unrolled (N <= 8).
//
// One lane owns one matrix: the per-bin problems of the hot path (N x N demixing
// matrices, N <= 8) are far too small to spread over a wavefront, so the batch
// axis (bins x mixtures) is what the 64 lanes cover.  Everything is statically
// indexed so the matrices live in VGPRs; partial pivoting is done with
// compare-and-swap of whole rows (v_cndmask), never with dynamic indexing.
#pragma once

#include "common.hpp"

namespace ssspy {

template <int N>
struct Mat {
  c128 a[N][N];
};

template <int N>
__device__ __forceinline__ void load_mat(Mat<N> &m, const c128 *__restrict__ p) {
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) m.a[r][c] = p[r * N + c];
}

template <int N>
__device__ __forceinline__ void store_mat(const Mat<N> &m, c128 *__restrict__ p) {
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) p[r * N + c] = m.a[r][c];
}

template <int N>
__device__ __forceinline__ void set_identity(Mat<N> &m) {
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) m.a[r][c] = cmake(r == c ? 1.0 : 0.0, 0.0);
}

template <int N>
__device__ __forceinline__ void matmul(Mat<N> &out, const Mat<N> &x, const Mat<N> &y) {
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) cfma(acc, x.a[r][k], y.a[k][c]);
      out.a[r][c] = acc;
    }
}

__device__ __forceinline__ void cswap_if(bool p, c128 &x, c128 &y) {
  c128 tx = x, ty = y;
  x = p ? ty : tx;
  y = p ? tx : ty;
}

// In-place LU with partial pivoting (pivot = largest |re|+|im|, as LAPACK izamax).
// `rhs` (NRHS columns) receives the same row operations.  Returns false when a pivot is 0.
// `det_sign_flips` counts row swaps (parity only matters for real determinants; we return
// log|det| so it is unused by callers but kept for completeness).
template <int N, int NRHS>
__device__ __forceinline__ bool lu_forward(Mat<N> &A, c128 (&rhs)[N][NRHS]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int r = k + 1; r < N; ++r) {
      const bool sw = cabs1(A.a[r][k]) > cabs1(A.a[k][k]);
#pragma unroll
      for (int c = k; c < N; ++c) cswap_if(sw, A.a[k][c], A.a[r][c]);
#pragma unroll
      for (int c = 0; c < NRHS; ++c) cswap_if(sw, rhs[k][c], rhs[r][c]);
    }
    const c128 piv = A.a[k][k];
    ok = ok && (piv.x != 0.0 || piv.y != 0.0);
    const c128 inv = crecip(piv);
#pragma unroll
    for (int r = k + 1; r < N; ++r) {
      const c128 f = cmul(A.a[r][k], inv);
#pragma unroll
      for (int c = k + 1; c < N; ++c) cfms(A.a[r][c], f, A.a[k][c]);
#pragma unroll
      for (int c = 0; c < NRHS; ++c) cfms(rhs[r][c], f, rhs[k][c]);
    }
  }
  return ok;
}

// back substitution on the upper triangle left by lu_forward
template <int N, int NRHS>
__device__ __forceinline__ void lu_backward(const Mat<N> &A, c128 (&rhs)[N][NRHS]) {
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const c128 inv = crecip(A.a[k][k]);
#pragma unroll
    for (int c = 0; c < NRHS; ++c) {
      c128 acc = rhs[k][c];
#pragma unroll
      for (int j = k + 1; j < N; ++j) cfms(acc, A.a[k][j], rhs[j][c]);
      rhs[k][c] = cmul(acc, inv);
    }
  }
}

// x = A^-1 e_col  (A is destroyed).  Returns false if singular.
template <int N>
__device__ __forceinline__ bool solve_unit(Mat<N> &A, int col, c128 (&x)[N]) {
  c128 rhs[N][1];
#pragma unroll
  for (int r = 0; r < N; ++r) rhs[r][0] = cmake(r == col ? 1.0 : 0.0, 0.0);
  const bool ok = lu_forward<N, 1>(A, rhs);
  lu_backward<N, 1>(A, rhs);
#pragma unroll
  for (int r = 0; r < N; ++r) x[r] = rhs[r][0];
  return ok;
}

// Ainv = A^-1 (A is destroyed).  Returns false if singular.
template <int N>
__device__ __forceinline__ bool invert(Mat<N> &A, Mat<N> &Ainv) {
  c128 rhs[N][N];
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) rhs[r][c] = cmake(r == c ? 1.0 : 0.0, 0.0);
  const bool ok = lu_forward<N, N>(A, rhs);
  lu_backward<N, N>(A, rhs);
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) Ainv.a[r][c] = rhs[r][c];
  return ok;
}

// log|det A| (A is destroyed); -inf when singular, like numpy.linalg.slogdet.
template <int N>
__device__ __forceinline__ double logabsdet(Mat<N> &A) {
  c128 dummy[N][1];
#pragma unroll
  for (int r = 0; r < N; ++r) dummy[r][0] = cmake(0.0, 0.0);
  lu_forward<N, 1>(A, dummy);
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) s += 0.5 * log(cabs2(A.a[k][k]));
  return s;
}

// Re(w^H U w) for Hermitian-ish U
template <int N>
__device__ __forceinline__ double quad_form(const c128 (&w)[N], const Mat<N> &U) {
  double q = 0.0;
#pragma unroll
  for (int a = 0; a < N; ++a) {
    c128 t = cmake(0.0, 0.0);
#pragma unroll
    for (int b = 0; b < N; ++b) cfma(t, U.a[a][b], w[b]);
    q += w[a].x * t.x + w[a].y * t.y;  // Re(conj(w_a) t)
  }
  return q;
}

}  // namespace ssspy
