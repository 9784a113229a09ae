// Small Hermitian matrices held by one lane: Jacobi eigen-decomposition and functions of a
// matrix through it (to_psd, inverse, square roots).  Used by the MNMF kernels, where every
// (bin, frame) point owns an M x M covariance.
#pragma once

#include "common.hpp"

namespace ssspy {

// Hermitian eigen-decomposition by cyclic complex Jacobi rotations: A = P diag(lam) P^H, with lam
// left on the diagonal of A.  Straight-line sweeps (no data-dependent branches inside a sweep);

// The rotation that annihilates A[p][q] of a Hermitian matrix: cs, su = s u (u = apq / |apq|) and
// tm = t |apq| (A_pp -= tm, A_qq += tm).  Round 6: reciprocals and square roots by v_rcp_f64 /
// v_rsq_f64 + two Newton steps (~1 ulp; the 8-lane form herm_rows8.hpp has used them since round 5)
// instead of three IEEE divides and two square roots -- ~170 of a rotation's dependent
// instructions in kernels that are chains of rotations (IPA's LQPQM, the Hermitian operators).
// Range: |apq|^2 is brought to [1/4, 2) by its exponent before the reciprocal square root and
// scaled back, tau is capped at 1e150 (beyond it t < 5e-151: the rotation is the identity to any
// rounding) -- no slow branch; |apq|^2 below 1e-300 (or NaN) leaves the pair alone.
struct JacobiRot {
  double cs, tm;
  c128 su;
};
__device__ __forceinline__ double jr_rsq(double x) {
  double r = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x * r;
  double e = fma(-h, r, 0.5);
  r = fma(r, e, r);
  h = 0.5 * x * r;
  e = fma(-h, r, 0.5);
  return fma(r, e, r);
}
__device__ __forceinline__ double jr_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ JacobiRot jacobi_rot(c128 apq, double app, double aqq) {
  const double mag2 = cabs2(apq);
  const bool tiny = !(mag2 >= 1e-300);
  const int half = __builtin_amdgcn_frexp_exp(tiny ? 1.0 : mag2) >> 1;  // mag2 = s 4^half, s in [1/4, 2)
  const double inv = tiny ? 0.0 : ldexp(jr_rsq(ldexp(mag2, -2 * half)), -half);  // 1 / |apq|
  const double mag = tiny ? 0.0 : mag2 * inv;
  const c128 u = tiny ? cmake(1.0, 0.0) : cmake(apq.x * inv, apq.y * inv);
  const double tau = tiny ? 0.0 : (aqq - app) * 0.5 * inv;
  const double atau = fmin(fabs(tau), 1e150);
  const double w = fma(atau, atau, 1.0);
  const double tabs = jr_rcp(fma(w, jr_rsq(w), atau));  // 1 / (|tau| + sqrt(1 + tau^2))
  const double t = tiny ? 0.0 : (tau >= 0.0 ? tabs : -tabs);
  JacobiRot r;
  r.cs = jr_rsq(fma(t, t, 1.0));
  const double sn = t * r.cs;
  r.su = cmake(sn * u.x, sn * u.y);
  r.tm = t * mag;
  return r;
}
// the sweep loop ends when every lane of the wave has a negligible off-diagonal, at most 12 sweeps.
template <int M>
__device__ __forceinline__ void jacobi_eigh(c128 (&A)[M][M], c128 (&P)[M][M]) {
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int cc = 0; cc < M; ++cc) P[r][cc] = cmake(r == cc ? 1.0 : 0.0, 0.0);
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int p = 0; p < M; ++p) {
      diag = fma(A[p][p].x, A[p][p].x, diag);
#pragma unroll
      for (int qq = p + 1; qq < M; ++qq) off += cabs2(A[p][qq]);
    }
    if (__all(off <= 1e-34 * diag)) break;
#pragma unroll
    for (int p = 0; p < M - 1; ++p)
#pragma unroll
      for (int qq = p + 1; qq < M; ++qq) {
        const double app = A[p][p].x, aqq = A[qq][qq].x;
        const JacobiRot rot = jacobi_rot(A[p][qq], app, aqq);
        const double cs = rot.cs;
        const c128 su = rot.su;           // s u
        const c128 sub = cconj(rot.su);   // s conj(u)
#pragma unroll
        for (int k = 0; k < M; ++k) {
          if (k != p && k != qq) {
            const c128 akp = A[k][p], akq = A[k][qq];
            // A'_kp = c A_kp - s conj(u) A_kq ; A'_kq = s u A_kp + c A_kq
            c128 nkp = cmake(cs * akp.x, cs * akp.y);
            cfms(nkp, sub, akq);
            c128 nkq = cmake(cs * akq.x, cs * akq.y);
            cfma(nkq, su, akp);
            A[k][p] = nkp;
            A[p][k] = cconj(nkp);
            A[k][qq] = nkq;
            A[qq][k] = cconj(nkq);
          }
        }
        A[p][p] = cmake(app - rot.tm, 0.0);
        A[qq][qq] = cmake(aqq + rot.tm, 0.0);
        A[p][qq] = cmake(0.0, 0.0);
        A[qq][p] = cmake(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < M; ++k) {
          const c128 vkp = P[k][p], vkq = P[k][qq];
          c128 nkp = cmake(cs * vkp.x, cs * vkp.y);
          cfms(nkp, sub, vkq);
          c128 nkq = cmake(cs * vkq.x, cs * vkq.y);
          cfma(nkq, su, vkp);
          P[k][p] = nkp;
          P[k][qq] = nkq;
        }
      }
  }
}

// The same sweeps with the pair loop ROLLED: A and P are indexed at run time and therefore live in
// the lane's scratch memory instead of registers -- for the repair kernels that hold an M x M
// working set per lane from 5 channels on (unrolled they spill 500-2 400 VGPRs and their code is
// megabytes; they run only where an eigenvalue floor may act).  Same pair order, same formulas.
template <int M>
__device__ __noinline__ void jacobi_eigh_rolled(c128 (&A)[M][M], c128 (&P)[M][M]) {
#pragma unroll 1
  for (int r = 0; r < M; ++r)
#pragma unroll 1
    for (int cc = 0; cc < M; ++cc) P[r][cc] = cmake(r == cc ? 1.0 : 0.0, 0.0);
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll 1
    for (int p = 0; p < M; ++p) {
      diag = fma(A[p][p].x, A[p][p].x, diag);
#pragma unroll 1
      for (int qq = p + 1; qq < M; ++qq) off += cabs2(A[p][qq]);
    }
    if (__all(off <= 1e-34 * diag)) break;
#pragma unroll 1
    for (int p = 0; p < M - 1; ++p)
#pragma unroll 1
      for (int qq = p + 1; qq < M; ++qq) {
        const double app = A[p][p].x, aqq = A[qq][qq].x;
        const JacobiRot rot = jacobi_rot(A[p][qq], app, aqq);
        const double cs = rot.cs;
        const c128 su = rot.su;           // s u
        const c128 sub = cconj(rot.su);   // s conj(u)
#pragma unroll 1
        for (int k = 0; k < M; ++k) {
          if (k != p && k != qq) {
            const c128 akp = A[k][p], akq = A[k][qq];
            c128 nkp = cmake(cs * akp.x, cs * akp.y);
            cfms(nkp, sub, akq);
            c128 nkq = cmake(cs * akq.x, cs * akq.y);
            cfma(nkq, su, akp);
            A[k][p] = nkp;
            A[p][k] = cconj(nkp);
            A[k][qq] = nkq;
            A[qq][k] = cconj(nkq);
          }
        }
        A[p][p] = cmake(app - rot.tm, 0.0);
        A[qq][qq] = cmake(aqq + rot.tm, 0.0);
        A[p][qq] = cmake(0.0, 0.0);
        A[qq][p] = cmake(0.0, 0.0);
#pragma unroll 1
        for (int k = 0; k < M; ++k) {
          const c128 vkp = P[k][p], vkq = P[k][qq];
          c128 nkp = cmake(cs * vkp.x, cs * vkp.y);
          cfms(nkp, sub, vkq);
          c128 nkq = cmake(cs * vkq.x, cs * vkq.y);
          cfma(nkq, su, vkp);
          P[k][p] = nkp;
          P[k][qq] = nkq;
        }
      }
  }
}

// (A + A^H) / 2 in place
template <int M>
__device__ __forceinline__ void hermitize(c128 (&A)[M][M]) {
#pragma unroll
  for (int a = 0; a < M; ++a) {
    A[a][a] = cmake(A[a][a].x, 0.0);
#pragma unroll
    for (int b = a + 1; b < M; ++b) {
      const c128 z = cmake(0.5 * (A[a][b].x + A[b][a].x), 0.5 * (A[a][b].y - A[b][a].y));
      A[a][b] = z;
      A[b][a] = cconj(z);
    }
  }
}

// Out = P diag(w) P^H (exactly Hermitian)
template <int M>
__device__ __forceinline__ void herm_rebuild(const c128 (&P)[M][M], const double (&w)[M],
                                             c128 (&Out)[M][M]) {
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int b = a; b < M; ++b) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const c128 t = cmulc(P[a][k], P[b][k]);  // P_ak conj(P_bk)
        s.x = fma(w[k], t.x, s.x);
        s.y = fma(w[k], t.y, s.y);
      }
      if (a == b) s.y = 0.0;
      Out[a][b] = s;
      Out[b][a] = cconj(s);
    }
}

// C = A B (general complex M x M)
template <int M>
__device__ __forceinline__ void matmul(const c128 (&A)[M][M], const c128 (&B)[M][M],
                                       c128 (&C)[M][M]) {
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int b = 0; b < M; ++b) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) cfma(s, A[a][k], B[k][b]);
      C[a][b] = s;
    }
}

// Memory-resident forms of herm_rebuild and matmul (same order of the sums) for the literal repair
// kernels whose M x M working set does not fit the register file (see jacobi_eigh_rolled)
template <int M>
__device__ __noinline__ void herm_rebuild_rolled(const c128 (&P)[M][M], const double (&w)[M],
                                                 c128 (&Out)[M][M]) {
#pragma unroll 1
  for (int a = 0; a < M; ++a)
#pragma unroll 1
    for (int b = a; b < M; ++b) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll 1
      for (int k = 0; k < M; ++k) {
        const c128 t = cmulc(P[a][k], P[b][k]);  // P_ak conj(P_bk)
        s.x = fma(w[k], t.x, s.x);
        s.y = fma(w[k], t.y, s.y);
      }
      if (a == b) s.y = 0.0;
      Out[a][b] = s;
      Out[b][a] = cconj(s);
    }
}

template <int M>
__device__ __noinline__ void matmul_rolled(const c128 (&A)[M][M], const c128 (&B)[M][M],
                                           c128 (&C)[M][M]) {
#pragma unroll 1
  for (int a = 0; a < M; ++a)
#pragma unroll 1
    for (int b = 0; b < M; ++b) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll 1
      for (int k = 0; k < M; ++k) cfma(s, A[a][k], B[k][b]);
      C[a][b] = s;
    }
}

// to_psd: Hermitise, eigen-decompose, floor the eigenvalues; returns the floored eigenvalues in
// lam and the eigenvectors in P (A is destroyed).  ref: ssspy/special/psd.py:11-71.
// ROLLED: the memory-resident sweeps (jacobi_eigh_rolled) -- for repair kernels from 6 channels on
template <int M, bool ROLLED = false>
__device__ __forceinline__ void psd_eigen(c128 (&A)[M][M], c128 (&P)[M][M], double (&lam)[M],
                                          int floor_kind, double eps) {
  hermitize<M>(A);
  if constexpr (ROLLED) jacobi_eigh_rolled<M>(A, P);
  else jacobi_eigh<M>(A, P);
#pragma unroll
  for (int k = 0; k < M; ++k) lam[k] = apply_floor(A[k][k].x, floor_kind, eps);
}

// Inverse and log-determinant of a Hermitian positive definite matrix by Cholesky (A = L L^H,
// A^-1 = L^-H L^-1).  Returns false when a pivot is not positive (A is destroyed either way).
template <int M>
__device__ __forceinline__ bool chol_inverse(c128 (&A)[M][M], c128 (&Inv)[M][M], double &logdet) {
  bool ok = true;
  double ld = 0.0;
  // in-place lower Cholesky: A[r][c], r >= c, becomes L
#pragma unroll
  for (int c = 0; c < M; ++c) {
    double d = A[c][c].x;
#pragma unroll
    for (int k = 0; k < c; ++k) d -= cabs2(A[c][k]);
    ok = ok && (d > 0.0);
    const double dd = d > 0.0 ? d : 1.0;
    const double l = sqrt(dd), il = 1.0 / l;
    ld += log(dd);
    A[c][c] = cmake(l, 0.0);
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      c128 s = A[r][c];
#pragma unroll
      for (int k = 0; k < c; ++k) cfms(s, A[r][k], cconj(A[c][k]));
      A[r][c] = cscale(s, il);
    }
  }
  logdet = ld;
  // Linv (lower): forward substitution on the identity
  c128 Li[M][M];
#pragma unroll
  for (int c = 0; c < M; ++c) {
#pragma unroll
    for (int r = 0; r < M; ++r) Li[r][c] = cmake(0.0, 0.0);
    Li[c][c] = cmake(1.0 / A[c][c].x, 0.0);
#pragma unroll
    for (int r = c + 1; r < M; ++r) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int k = c; k < r; ++k) cfms(s, A[r][k], Li[k][c]);
      Li[r][c] = cscale(s, 1.0 / A[r][r].x);
    }
  }
  // Inv = Linv^H Linv
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int b = a; b < M; ++b) {
      c128 s = cmake(0.0, 0.0);
#pragma unroll
      for (int k = b; k < M; ++k) {
        const c128 t = cmulc(Li[k][b], Li[k][a]);  // conj(Li[k][a]) Li[k][b]
        s.x += t.x;
        s.y += t.y;
      }
      if (a == b) s.y = 0.0;
      Inv[a][b] = s;
      Inv[b][a] = cconj(s);
    }
  return ok;
}

}  // namespace ssspy
