// Per-lane Hermitian linear algebra with the dimension at run time (matrices in the lane's private
// memory, row-major, leading dimension N <= SSSPY_RT_MAX_SOURCES): the run-time-N statements of
// hermitian.hpp's templates, shared by the IPA sweep at 9-16 sources (ipa_rt.hip) and the Hermitian
// operators at 9-16 (hermitian_rt.hip).  Every lane of the wave must be in a call of rt_jacobi (the
// sweep loop ends on a wave vote).
#pragma once

#include "common.hpp"
#include "hermitian.hpp"
#include "rt_dense.hpp"

namespace ssspy {

// (A + A^H) / 2 in place
__device__ inline void rt_hermitize(c128 *A, int N) {
  for (int a = 0; a < N; ++a) {
    A[a * N + a] = cmake(A[a * N + a].x, 0.0);
    for (int b = a + 1; b < N; ++b) {
      const c128 z = cmake(0.5 * (A[a * N + b].x + A[b * N + a].x),
                           0.5 * (A[a * N + b].y - A[b * N + a].y));
      A[a * N + b] = z;
      A[b * N + a] = cconj(z);
    }
  }
}

// lam_min(A) > shift by the pivots of the Cholesky factorisation of A - shift I (W: working copy)
__device__ inline bool rt_shifted_pd(const c128 *A, c128 *W, int N, double shift) {
  bool ok = true;
  for (int c = 0; c < N; ++c) {
    double d = A[c * N + c].x - shift;
    for (int k = 0; k < c; ++k) d -= cabs2(W[c * N + k]);
    ok = ok && (d > 0.0);
    const double il = 1.0 / sqrt(d > 0.0 ? d : 1.0);
    for (int r = c + 1; r < N; ++r) {
      c128 sum = A[r * N + c];
      for (int k = 0; k < c; ++k) cfms(sum, W[r * N + k], cconj(W[c * N + k]));
      W[r * N + c] = cscale(sum, il);
    }
  }
  return ok;
}

// cyclic complex Jacobi (the sweeps of jacobi_eigh, hermitian.hpp): A = P diag(lam) P^H, lam on the
// diagonal of A.  (The sweep loop ends when every lane that is in the call has converged.)
__device__ inline void rt_jacobi(c128 *A, c128 *P, int N) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) P[r * N + c] = cmake(r == c ? 1.0 : 0.0, 0.0);
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < N; ++p) {
      diag = fma(A[p * N + p].x, A[p * N + p].x, diag);
      for (int q = p + 1; q < N; ++q) off += cabs2(A[p * N + q]);
    }
    if (__all(off <= 1e-34 * diag)) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double app = A[p * N + p].x, aqq = A[q * N + q].x;
        const JacobiRot rot = jacobi_rot(A[p * N + q], app, aqq);
        const double cs = rot.cs;
        const c128 su = rot.su, sub = cconj(rot.su);
        for (int k = 0; k < N; ++k) {
          if (k != p && k != q) {
            const c128 akp = A[k * N + p], akq = A[k * N + q];
            c128 nkp = cmake(cs * akp.x, cs * akp.y);
            cfms(nkp, sub, akq);
            c128 nkq = cmake(cs * akq.x, cs * akq.y);
            cfma(nkq, su, akp);
            A[k * N + p] = nkp;
            A[p * N + k] = cconj(nkp);
            A[k * N + q] = nkq;
            A[q * N + k] = cconj(nkq);
          }
        }
        A[p * N + p] = cmake(app - rot.tm, 0.0);
        A[q * N + q] = cmake(aqq + rot.tm, 0.0);
        A[p * N + q] = cmake(0.0, 0.0);
        A[q * N + p] = cmake(0.0, 0.0);
        for (int k = 0; k < N; ++k) {
          const c128 vkp = P[k * N + p], vkq = P[k * N + q];
          c128 nkp = cmake(cs * vkp.x, cs * vkp.y);
          cfms(nkp, sub, vkq);
          c128 nkq = cmake(cs * vkq.x, cs * vkq.y);
          cfma(nkq, su, vkp);
          P[k * N + p] = nkp;
          P[k * N + q] = nkq;
        }
      }
  }
}

// to_psd: Hermitise, eigen-decompose, floor the eigenvalues (psd_eigen, hermitian.hpp)
__device__ inline void rt_psd_eigen(c128 *A, c128 *P, double *lam, int N, int floor_kind, double eps) {
  rt_hermitize(A, N);
  rt_jacobi(A, P, N);
  for (int k = 0; k < N; ++k) lam[k] = apply_floor(A[k * N + k].x, floor_kind, eps);
}

// Out = P diag(w) P^H (exactly Hermitian)
__device__ inline void rt_rebuild(const c128 *P, const double *w, c128 *Out, int N) {
  for (int a = 0; a < N; ++a)
    for (int b = a; b < N; ++b) {
      c128 s = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) {
        const c128 t = cmulc(P[a * N + k], P[b * N + k]);
        s.x = fma(w[k], t.x, s.x);
        s.y = fma(w[k], t.y, s.y);
      }
      if (a == b) s.y = 0.0;
      Out[a * N + b] = s;
      Out[b * N + a] = cconj(s);
    }
}

// Inverse of a Hermitian positive definite matrix by Cholesky (chol_inverse, hermitian.hpp): A is
// destroyed (its lower triangle becomes L), Li is scratch (L^-1).  False: a pivot was not positive.
__device__ inline bool rt_chol_inverse(c128 *A, c128 *Inv, c128 *Li, int N) {
  bool ok = true;
  for (int c = 0; c < N; ++c) {
    double d = A[c * N + c].x;
    for (int k = 0; k < c; ++k) d -= cabs2(A[c * N + k]);
    ok = ok && (d > 0.0);
    const double dd = d > 0.0 ? d : 1.0;
    const double l = sqrt(dd), il = 1.0 / l;
    A[c * N + c] = cmake(l, 0.0);
    for (int r = c + 1; r < N; ++r) {
      c128 s = A[r * N + c];
      for (int k = 0; k < c; ++k) cfms(s, A[r * N + k], cconj(A[c * N + k]));
      A[r * N + c] = cscale(s, il);
    }
  }
  for (int c = 0; c < N; ++c) {
    for (int r = 0; r < N; ++r) Li[r * N + c] = cmake(0.0, 0.0);
    Li[c * N + c] = cmake(1.0 / A[c * N + c].x, 0.0);
    for (int r = c + 1; r < N; ++r) {
      c128 s = cmake(0.0, 0.0);
      for (int k = c; k < r; ++k) cfms(s, A[r * N + k], Li[k * N + c]);
      Li[r * N + c] = cscale(s, 1.0 / A[r * N + r].x);
    }
  }
  for (int a = 0; a < N; ++a)
    for (int b = a; b < N; ++b) {
      c128 s = cmake(0.0, 0.0);
      for (int k = b; k < N; ++k) {
        const c128 t = cmulc(Li[k * N + b], Li[k * N + a]);  // conj(Li[k][a]) Li[k][b]
        s.x += t.x;
        s.y += t.y;
      }
      if (a == b) s.y = 0.0;
      Inv[a * N + b] = s;
      Inv[b * N + a] = cconj(s);
    }
  return ok;
}

// C = A B
__device__ inline void rt_matmul(const c128 *A, const c128 *B, c128 *C, int N) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      c128 s = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(s, A[r * N + k], B[k * N + c]);
      C[r * N + c] = s;
    }
}

// C = A B^H
__device__ inline void rt_matmul_h(const c128 *A, const c128 *B, c128 *C, int N) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      c128 s = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(s, A[r * N + k], cconj(B[c * N + k]));
      C[r * N + c] = s;
    }
}

// C = A^H B
__device__ inline void rt_matmul_hl(const c128 *A, const c128 *B, c128 *C, int N) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      c128 s = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(s, cconj(A[k * N + r]), B[k * N + c]);
      C[r * N + c] = s;
    }
}

}  // namespace ssspy
