// Hermitian M x M matrices in packed form -- the real diagonal and the upper triangle, M^2 doubles --
// and the in-place inverse the per-point kernels of GaussMNMF run on them.  One lane owns one
// matrix; every index below is a compile-time constant after unrolling, so the matrix lives in
// M^2 registers (64 for 8 channels, where the full-storage routines of hermitian.hpp keep four
// 128-double matrices alive and spill thousands of registers).
#pragma once

#include "common.hpp"
#include "hermitian.hpp"

namespace ssspy {

template <int M>
struct HermP {
  double d[M];                               // A[a][a]
  c128 o[M > 1 ? (M * (M - 1)) / 2 : 1];     // A[a][c], a < c, row-major
};

// slot of the off-diagonal (a, c), a < c
template <int M>
__host__ __device__ constexpr int tri(int a, int c) {
  return a * M - (a * (a + 1)) / 2 + (c - a - 1);
}

// A[a][c] for any a, c
template <int M>
__device__ __forceinline__ c128 hp_get(const HermP<M> &A, int a, int c) {
  if (a == c) return cmake(A.d[a], 0.0);
  if (a < c) return A.o[tri<M>(a, c)];
  return cconj(A.o[tri<M>(c, a)]);
}

template <int M>
__device__ __forceinline__ void hp_clear(HermP<M> &A) {
#pragma unroll
  for (int a = 0; a < M; ++a) A.d[a] = 0.0;
#pragma unroll
  for (int e = 0; e < (M * (M - 1)) / 2; ++e) A.o[e] = cmake(0.0, 0.0);
}

// A += l * H with H packed in memory as [M diagonal][re, im of the upper triangle]
template <int M>
__device__ __forceinline__ void hp_axpy(HermP<M> &A, double l, const double *__restrict__ Hp) {
#pragma unroll
  for (int a = 0; a < M; ++a) A.d[a] = fma(l, Hp[a], A.d[a]);
#pragma unroll
  for (int e = 0; e < (M * (M - 1)) / 2; ++e) {
    A.o[e].x = fma(l, Hp[M + 2 * e], A.o[e].x);
    A.o[e].y = fma(l, Hp[M + 2 * e + 1], A.o[e].y);
  }
}

// squared Frobenius norm
template <int M>
__device__ __forceinline__ double hp_fro2(const HermP<M> &A) {
  double s = 0.0, t = 0.0;
#pragma unroll
  for (int a = 0; a < M; ++a) s = fma(A.d[a], A.d[a], s);
#pragma unroll
  for (int e = 0; e < (M * (M - 1)) / 2; ++e) t += cabs2(A.o[e]);
  return fma(2.0, t, s);
}

// u = A x
template <int M>
__device__ __forceinline__ void hp_matvec(const HermP<M> &A, const c128 (&x)[M], c128 (&u)[M]) {
#pragma unroll
  for (int a = 0; a < M; ++a) {
    c128 s = cscale(x[a], A.d[a]);
#pragma unroll
    for (int c = 0; c < M; ++c)
      if (c != a) cfma(s, hp_get<M>(A, a, c), x[c]);
    u[a] = s;
  }
}

// (A A)[a][c], a <= c
template <int M>
__device__ __forceinline__ c128 hp_square_entry(const HermP<M> &A, int a, int c) {
  c128 s = cmake(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < M; ++k) cfma(s, hp_get<M>(A, a, k), hp_get<M>(A, k, c));
  return s;
}

template <int M>
__device__ __forceinline__ void hp_set(HermP<M> &A, int a, int c, c128 v) {
  if (a == c) A.d[a] = v.x;
  else if (a < c) A.o[tri<M>(a, c)] = v;
  else A.o[tri<M>(c, a)] = cconj(v);
}

// The three sweeps of the in-place inverse, separately (the spatial update of GaussMNMF needs the
// triangular factor and its inverse themselves).
// A = U^H U: U (upper, real positive diagonal) replaces A; dinv = 1 / diag U; returns false on a
// non-positive pivot (NaN included)
template <int M>
__device__ __forceinline__ bool hp_chol_upper(HermP<M> &A, double (&dinv)[M], double &logdet) {
  bool ok = true;
  double ld = 0.0;
#pragma unroll
  for (int k = 0; k < M; ++k) {
    double s = A.d[k];
#pragma unroll
    for (int p = 0; p < k; ++p) s -= cabs2(A.o[tri<M>(p, k)]);
    ok = ok && (s > 0.0);
    const double ss = s > 0.0 ? s : 1.0;
    const double ukk = sqrt(ss), inv = 1.0 / ukk;
    ld += log(ss);
    A.d[k] = ukk;
    dinv[k] = inv;
#pragma unroll
    for (int c = k + 1; c < M; ++c) {
      c128 t = A.o[tri<M>(k, c)];
#pragma unroll
      for (int p = 0; p < k; ++p) {  // t -= conj(U[p][k]) U[p][c]
        const c128 upk = A.o[tri<M>(p, k)], upc = A.o[tri<M>(p, c)];
        t.x = fma(-upk.x, upc.x, t.x);
        t.x = fma(-upk.y, upc.y, t.x);
        t.y = fma(-upk.x, upc.y, t.y);
        t.y = fma(upk.y, upc.x, t.y);
      }
      A.o[tri<M>(k, c)] = cscale(t, inv);
    }
  }
  logdet = ld;
  return ok;
}

// U (upper, as left by hp_chol_upper) -> V = U^-1 in place; the diagonal of V is dinv (A.d is set
// to it as well)
template <int M>
__device__ __forceinline__ void hp_trtri_upper(HermP<M> &A, const double (&dinv)[M]) {
#pragma unroll
  for (int k = 0; k < M; ++k)
#pragma unroll
    for (int c = k + 1; c < M; ++c) {
      c128 s = cscale(A.o[tri<M>(k, c)], dinv[k]);
#pragma unroll
      for (int p = k + 1; p < c; ++p) cfma(s, A.o[tri<M>(k, p)], A.o[tri<M>(p, c)]);
      A.o[tri<M>(k, c)] = cscale(s, -dinv[c]);
    }
#pragma unroll
  for (int k = 0; k < M; ++k) A.d[k] = dinv[k];
}

// squared Frobenius norm of an upper TRIANGULAR matrix in the packed slots (diagonal A.d)
template <int M>
__device__ __forceinline__ double hp_fro2_upper(const HermP<M> &A) {
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < M; ++a) s = fma(A.d[a], A.d[a], s);
#pragma unroll
  for (int e = 0; e < (M * (M - 1)) / 2; ++e) s += cabs2(A.o[e]);
  return s;
}

// Out = U B U^H for an upper triangular U (packed slots, real diagonal) and a Hermitian B
template <int M>
__device__ __forceinline__ void hp_congruence_upper(const HermP<M> &U, const HermP<M> &B,
                                                    HermP<M> &Out) {
#pragma unroll
  for (int a = 0; a < M; ++a) {
    c128 w[M];  // row a of U B
#pragma unroll
    for (int c = 0; c < M; ++c) {
      c128 s = cscale(hp_get<M>(B, a, c), U.d[a]);
#pragma unroll
      for (int k = a + 1; k < M; ++k) cfma(s, U.o[tri<M>(a, k)], hp_get<M>(B, k, c));
      w[c] = s;
    }
#pragma unroll
    for (int c = a; c < M; ++c) {  // sum_{k >= c} w[k] conj(U[c][k])
      c128 s = cscale(w[c], U.d[c]);
#pragma unroll
      for (int k = c + 1; k < M; ++k) {
        const c128 u = U.o[tri<M>(c, k)];
        s.x = fma(w[k].x, u.x, s.x);
        s.x = fma(w[k].y, u.y, s.x);
        s.y = fma(w[k].y, u.x, s.y);
        s.y = fma(-w[k].x, u.y, s.y);
      }
      if (c == a) Out.d[a] = s.x;
      else Out.o[tri<M>(a, c)] = s;
    }
  }
}

// Cyclic complex Jacobi on a packed Hermitian matrix (the rotations of jacobi_eigh, hermitian.hpp):
// A = P diag(lam) P^H, lam left in A.d.
template <int M>
__device__ __forceinline__ void hp_jacobi_eigh(HermP<M> &A, c128 (&P)[M][M]) {
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int cc = 0; cc < M; ++cc) P[r][cc] = cmake(r == cc ? 1.0 : 0.0, 0.0);
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int p = 0; p < M; ++p) diag = fma(A.d[p], A.d[p], diag);
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) off += cabs2(A.o[e]);
    if (__all(off <= 1e-34 * diag)) break;
#pragma unroll
    for (int p = 0; p < M - 1; ++p)
#pragma unroll
      for (int qq = p + 1; qq < M; ++qq) {
        const double app = A.d[p], aqq = A.d[qq];
        const JacobiRot rot = jacobi_rot(A.o[tri<M>(p, qq)], app, aqq);
        const double cs = rot.cs;
        const c128 su = rot.su;           // s u
        const c128 sub = cconj(rot.su);   // s conj(u)
#pragma unroll
        for (int k = 0; k < M; ++k) {
          if (k != p && k != qq) {
            const c128 akp = hp_get<M>(A, k, p), akq = hp_get<M>(A, k, qq);
            c128 nkp = cmake(cs * akp.x, cs * akp.y);
            cfms(nkp, sub, akq);
            c128 nkq = cmake(cs * akq.x, cs * akq.y);
            cfma(nkq, su, akp);
            hp_set<M>(A, k, p, nkp);
            hp_set<M>(A, k, qq, nkq);
          }
        }
        A.d[p] = app - rot.tm;
        A.d[qq] = aqq + rot.tm;
        A.o[tri<M>(p, qq)] = cmake(0.0, 0.0);
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

// The same sweeps with the rotations applied to NR rows of a given matrix instead of the identity:
// W <- W J for the J that diagonalises A (A = J diag(lam) J^H; lam left in A.d).  Rows transform
// independently, so a matrix too large for the register file beside A takes its rows in turns --
// every turn repeats the (deterministic) rotation sequence on a fresh copy of A.
template <int M, int NR>
__device__ __forceinline__ void hp_jacobi_rows(HermP<M> &A, c128 (&W)[NR][M]) {
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int p = 0; p < M; ++p) diag = fma(A.d[p], A.d[p], diag);
#pragma unroll
    for (int e = 0; e < (M * (M - 1)) / 2; ++e) off += cabs2(A.o[e]);
    if (__all(off <= 1e-34 * diag)) break;
#pragma unroll
    for (int p = 0; p < M - 1; ++p)
#pragma unroll
      for (int qq = p + 1; qq < M; ++qq) {
        const double app = A.d[p], aqq = A.d[qq];
        const JacobiRot rot = jacobi_rot(A.o[tri<M>(p, qq)], app, aqq);
        const double cs = rot.cs;
        const c128 su = rot.su;           // s u
        const c128 sub = cconj(rot.su);   // s conj(u)
#pragma unroll
        for (int k = 0; k < M; ++k) {
          if (k != p && k != qq) {
            const c128 akp = hp_get<M>(A, k, p), akq = hp_get<M>(A, k, qq);
            c128 nkp = cmake(cs * akp.x, cs * akp.y);
            cfms(nkp, sub, akq);
            c128 nkq = cmake(cs * akq.x, cs * akq.y);
            cfma(nkq, su, akp);
            hp_set<M>(A, k, p, nkp);
            hp_set<M>(A, k, qq, nkq);
          }
        }
        A.d[p] = app - rot.tm;
        A.d[qq] = aqq + rot.tm;
        A.o[tri<M>(p, qq)] = cmake(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const c128 vkp = W[r][p], vkq = W[r][qq];
          c128 nkp = cmake(cs * vkp.x, cs * vkp.y);
          cfms(nkp, sub, vkq);
          c128 nkq = cmake(cs * vkq.x, cs * vkq.y);
          cfma(nkq, su, vkp);
          W[r][p] = nkp;
          W[r][qq] = nkq;
        }
      }
  }
}

// Out = P diag(w) P^H, packed
template <int M>
__device__ __forceinline__ void hp_rebuild(const c128 (&P)[M][M], const double (&w)[M],
                                           HermP<M> &Out) {
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
      if (a == b) Out.d[a] = s.x;
      else Out.o[tri<M>(a, b)] = s;
    }
}

// In-place inverse of a Hermitian positive definite matrix, A = U^H U (U upper, real positive
// diagonal), U^-1 in place, A^-1 = U^-1 U^-H in place; logdet = log det A.  Returns false when a
// pivot is not positive (NaN included); the content is then unspecified.
// Order of the three sweeps (each overwrites only what no later step of the sweep reads):
//   Cholesky, row k: reads rows p < k (final) and row k of A;
//   inverse of U, row k left to right: V[k][c] reads V[k][p], p < c (new), U[p][c], k < p < c (rows
//     below, untouched) and U[k][c] itself before it is replaced;
//   V V^H, row a left to right: entry (a, c) reads V[a][p], p >= c (right of it, untouched) and
//     rows c > a (untouched).
template <int M>
__device__ __forceinline__ bool hp_chol_inverse(HermP<M> &A, double &logdet) {
  bool ok = true;
  double ld = 0.0;
  double dinv[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    double s = A.d[k];
#pragma unroll
    for (int p = 0; p < k; ++p) s -= cabs2(A.o[tri<M>(p, k)]);
    ok = ok && (s > 0.0);
    const double ss = s > 0.0 ? s : 1.0;
    const double ukk = sqrt(ss), inv = 1.0 / ukk;
    ld += log(ss);
    A.d[k] = ukk;
    dinv[k] = inv;
#pragma unroll
    for (int c = k + 1; c < M; ++c) {
      c128 t = A.o[tri<M>(k, c)];
#pragma unroll
      for (int p = 0; p < k; ++p) {  // t -= conj(U[p][k]) U[p][c]
        const c128 upk = A.o[tri<M>(p, k)], upc = A.o[tri<M>(p, c)];
        t.x = fma(-upk.x, upc.x, t.x);
        t.x = fma(-upk.y, upc.y, t.x);
        t.y = fma(-upk.x, upc.y, t.y);
        t.y = fma(upk.y, upc.x, t.y);
      }
      A.o[tri<M>(k, c)] = cscale(t, inv);
    }
  }
  logdet = ld;
  // V = U^-1 (upper), its diagonal in dinv
#pragma unroll
  for (int k = 0; k < M; ++k)
#pragma unroll
    for (int c = k + 1; c < M; ++c) {
      c128 s = cscale(A.o[tri<M>(k, c)], dinv[k]);
#pragma unroll
      for (int p = k + 1; p < c; ++p) cfma(s, A.o[tri<M>(k, p)], A.o[tri<M>(p, c)]);
      A.o[tri<M>(k, c)] = cscale(s, -dinv[c]);
    }
  // A^-1 = V V^H
#pragma unroll
  for (int a = 0; a < M; ++a) {
    double dd = dinv[a] * dinv[a];
#pragma unroll
    for (int p = a + 1; p < M; ++p) dd += cabs2(A.o[tri<M>(a, p)]);
#pragma unroll
    for (int c = a + 1; c < M; ++c) {
      c128 s = cscale(A.o[tri<M>(a, c)], dinv[c]);
#pragma unroll
      for (int p = c + 1; p < M; ++p) {  // s += V[a][p] conj(V[c][p])
        const c128 vap = A.o[tri<M>(a, p)], vcp = A.o[tri<M>(c, p)];
        s.x = fma(vap.x, vcp.x, s.x);
        s.x = fma(vap.y, vcp.y, s.x);
        s.y = fma(vap.y, vcp.x, s.y);
        s.y = fma(-vap.x, vcp.y, s.y);
      }
      A.o[tri<M>(a, c)] = s;
    }
    A.d[a] = dd;
  }
  return ok;
}

}  // namespace ssspy
