// Per-lane dense complex linear algebra with the dimension at run time (matrices in the lane's
// private memory, row-major, leading dimension N <= SSSPY_RT_MAX_SOURCES): the building blocks of the
// run-time-N kernels (wide_n.hip, pairwise_kernels.hip).
#pragma once

#include "common.hpp"
#include "ssspy_amd.h"

namespace ssspy {

constexpr int RTN = SSSPY_RT_MAX_SOURCES;

// ---- per-lane dense helpers (row-major, leading dimension N) -------------------------------------
// LU with partial pivoting (largest |re| + |im| of the column, the first one on ties), the right-hand
// sides (N x R, leading dimension R) take the same row operations.  false: a zero pivot.
__device__ inline bool rt_lu_solve(c128 *A, c128 *rhs, int N, int R) {
  bool ok = true;
  for (int k = 0; k < N; ++k) {
    int p = k;
    double best = cabs1(A[k * N + k]);
    for (int r = k + 1; r < N; ++r) {
      const double v = cabs1(A[r * N + k]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
    if (p != k) {
      for (int c = k; c < N; ++c) {
        const c128 t = A[k * N + c];
        A[k * N + c] = A[p * N + c];
        A[p * N + c] = t;
      }
      for (int c = 0; c < R; ++c) {
        const c128 t = rhs[k * R + c];
        rhs[k * R + c] = rhs[p * R + c];
        rhs[p * R + c] = t;
      }
    }
    const c128 piv = A[k * N + k];
    ok = ok && (piv.x != 0.0 || piv.y != 0.0);
    const c128 inv = crecip(piv);
    for (int r = k + 1; r < N; ++r) {
      const c128 f = cmul(A[r * N + k], inv);
      for (int c = k + 1; c < N; ++c) cfms(A[r * N + c], f, A[k * N + c]);
      for (int c = 0; c < R; ++c) cfms(rhs[r * R + c], f, rhs[k * R + c]);
    }
  }
  for (int k = N - 1; k >= 0; --k) {
    const c128 inv = crecip(A[k * N + k]);
    for (int c = 0; c < R; ++c) {
      c128 acc = rhs[k * R + c];
      for (int j = k + 1; j < N; ++j) cfms(acc, A[k * N + j], rhs[j * R + c]);
      rhs[k * R + c] = cmul(acc, inv);
    }
  }
  return ok;
}

// log|det A| by the same elimination (A is destroyed); -inf when singular
__device__ inline double rt_logabsdet(c128 *A, int N) {
  double s = 0.0;
  for (int k = 0; k < N; ++k) {
    int p = k;
    double best = cabs1(A[k * N + k]);
    for (int r = k + 1; r < N; ++r) {
      const double v = cabs1(A[r * N + k]);
      if (v > best) {
        best = v;
        p = r;
      }
    }
    if (p != k)
      for (int c = k; c < N; ++c) {
        const c128 t = A[k * N + c];
        A[k * N + c] = A[p * N + c];
        A[p * N + c] = t;
      }
    const c128 piv = A[k * N + k];
    s += 0.5 * log(cabs2(piv));
    const c128 inv = crecip(piv);
    for (int r = k + 1; r < N; ++r) {
      const c128 f = cmul(A[r * N + k], inv);
      for (int c = k + 1; c < N; ++c) cfms(A[r * N + c], f, A[k * N + c]);
    }
  }
  return s;
}

// Re(v^H M v) with M read from global memory (leading dimension N), rows in order like quad_form
__device__ inline double rt_quad(const c128 *v, const c128 *__restrict__ M, int N) {
  double q = 0.0;
  for (int a = 0; a < N; ++a) {
    c128 t = cmake(0.0, 0.0);
    for (int b = 0; b < N; ++b) cfma(t, M[a * N + b], v[b]);
    q += v[a].x * t.x + v[a].y * t.y;
  }
  return q;
}

}  // namespace ssspy
