// Hermitian matrices of up to 8 x 8 with a ROW PER LANE: 8 lanes per matrix, 8 matrices per wave
// (round 5).  The lane-per-matrix kernels keep a whole matrix, its factor and its eigenvectors in
// one lane's registers -- from 7 x 7 on that is 3 500 spilled VGPRs and one wave per SIMD walking
// a chain of dependent scratch round trips.  Here lane r of an 8-lane group holds row r of every
// operand (16 VGPRs per matrix row), rows travel by ds_bpermute shuffles or through a 1 KB exchange
// slot per matrix in LDS, and the Jacobi sweeps run in the round-robin order: 7 rounds of 4
// disjoint rotations, every lane computing the rotation of its own pair.
//
// Matrices smaller than 8 x 8 are padded: zero rows / columns (unit diagonal where a factorisation
// needs it).  Rotations that would mix a padded index see a zero off-diagonal and are skipped, so
// the padding never reaches the leading block.
#pragma once

#include "common.hpp"

namespace ssspy {
namespace rows8 {

constexpr int LD = 8;             // row stride of an exchange slot (c128)
constexpr int SLOT = 8 * LD + 1;  // c128 per slot: the 8 slots of a wave start 16 bytes apart in the banks

__device__ __forceinline__ double shfl8(double x, int src) { return __shfl(x, src, 8); }
__device__ __forceinline__ c128 shfl8(c128 z, int src) {
  return cmake(__shfl(z.x, src, 8), __shfl(z.y, src, 8));
}
__device__ __forceinline__ double sum8(double x) {
  x += __shfl_xor(x, 1, 8);
  x += __shfl_xor(x, 2, 8);
  x += __shfl_xor(x, 4, 8);
  return x;
}

// v_rcp_f64 / v_rsq_f64 + two Newton steps (~1 ulp, benchmarks/micro/rcp_precision.hip) for finite
// positive arguments away from the ends of the exponent range -- the callers guard those
__device__ __forceinline__ double rcp2(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double rsq2(double x) {
  double r = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x * r;
  double e = fma(-h, r, 0.5);
  r = fma(r, e, r);
  h = 0.5 * x * r;
  e = fma(-h, r, 0.5);
  return fma(r, e, r);
}

// a[i] / a[i] = v for a lane-dependent index (a chain of selects: a register array has no
// run-time index)
__device__ __forceinline__ c128 sel(const c128 (&a)[8], int i) {
  c128 r = a[0];
#pragma unroll
  for (int c = 1; c < 8; ++c) {
    r.x = (i == c) ? a[c].x : r.x;
    r.y = (i == c) ? a[c].y : r.y;
  }
  return r;
}
__device__ __forceinline__ void put(c128 (&a)[8], int i, c128 v) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    a[c].x = (i == c) ? v.x : a[c].x;
    a[c].y = (i == c) ? v.y : a[c].y;
  }
}

// LDS traffic of one wave is in order; this keeps the compiler from moving accesses across the
// hand-over between the lane that wrote a row and the lanes that read it
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void store_row(c128 *X, int r, const c128 (&row)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) X[r * LD + c] = row[c];
}

// out[c] = sum_k a[k] X[k][c]   (row r of A X, X in the slot)
__device__ __forceinline__ void mul_rows(const c128 (&a)[8], const c128 *X, c128 (&out)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) out[c] = cmake(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int c = 0; c < 8; ++c) cfma(out[c], a[k], X[k * LD + c]);
    // (one row of the slot in flight at a time: hoisting all 64 reads costs 128 VGPRs)
    asm volatile("" ::: "memory");
  }
}

// out[c] = sum_k a[k] conj(X[c][k])   (row r of A X^H)
__device__ __forceinline__ void mul_rows_adj(const c128 (&a)[8], const c128 *X, c128 (&out)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    c128 s = cmake(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const c128 x = X[c * LD + k];
      s.x = fma(a[k].x, x.x, s.x);
      s.x = fma(a[k].y, x.y, s.x);
      s.y = fma(a[k].y, x.x, s.y);
      s.y = fma(-a[k].x, x.y, s.y);
    }
    out[c] = s;
    asm volatile("" ::: "memory");
  }
}

// P = U^H U (U upper triangular, real positive diagonal).  In: row r of the full Hermitian P.
// Out: row r of U (zero left of the diagonal).  False when a pivot is not positive (NaN included);
// the same verdict in all lanes of the group.
__device__ __forceinline__ bool chol_upper(c128 (&row)[8], int r) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double s = shfl8(row[k].x, k);  // the pivot, from lane k
    // (its reciprocal square root heads every step's dependency chain: v_rsq_f64 + two Newton steps.
    //  A pivot outside [1e-300, 1e300] counts as "not positive definite": the callers' repair /
    //  literal routes take such matrices, as they take indefinite ones)
    const bool pos = s > 1e-300 && s < 1e300;
    ok = ok && pos;
    const double sp = pos ? s : 1.0;
    const double dinv = rsq2(sp);
    const double d = sp * dinv;
    c128 uk[8];
#pragma unroll
    for (int c = k + 1; c < 8; ++c) uk[c] = shfl8(cscale(row[c], dinv), k);  // row k of U
    // the trailing block keeps both triangles: conj(U[k][r]) = P[r][k] / d is lane r's own entry
    const c128 f = cscale(row[k], dinv);
#pragma unroll
    for (int c = k + 1; c < 8; ++c) {
      c128 upd = row[c];
      cfms(upd, f, uk[c]);
      row[c].x = (r > k) ? upd.x : ((r == k) ? uk[c].x : row[c].x);
      row[c].y = (r > k) ? upd.y : ((r == k) ? uk[c].y : row[c].y);
    }
    row[k].x = (r > k) ? 0.0 : ((r == k) ? d : row[k].x);
    row[k].y = (r >= k) ? 0.0 : row[k].y;
  }
  return ok;
}

// Column c of V = U^-1 for the upper triangular U in the slot (lane c solves U v = e_c); entries
// below the diagonal are zero.
__device__ __forceinline__ void trtri_col(const c128 *X, int c, c128 (&v)[8]) {
#pragma unroll
  for (int k = 7; k >= 0; --k) {
    c128 acc = cmake(0.0, 0.0);
#pragma unroll
    for (int j = k + 1; j < 8; ++j) cfma(acc, X[k * LD + j], v[j]);  // (v[j] = 0 beyond c)
    const double inv = rcp2(X[k * LD + k].x);  // (a diagonal entry of a factor chol_upper accepted)
    asm volatile("" ::: "memory");
    v[k].x = (k == c) ? inv : ((k < c) ? -acc.x * inv : 0.0);
    v[k].y = (k < c) ? -acc.y * inv : 0.0;
  }
}

// --- Jacobi in the round-robin order, by POSITION (Brent-Luk): every round rotates the pairs of
// positions (0,1) (2,3) (4,5) (6,7) -- partner lane = lane ^ 1, a DPP quad permutation, and the
// same straight-line code in every round -- and then moves the indices one step round the
// tournament cycle 1 -> 2 -> 4 -> 6 -> 7 -> 5 -> 3 -> 1 (position 0 stays): rows change lanes
// (one ds_bpermute pattern), columns change registers.  Seven rounds meet all 28 pairs and bring
// every index back to where it started, so eigenvalue k and column k of W J stay at position k
// from sweep to sweep.  (A first version kept the indices in place and let the partner vary per
// round: 56 comparison masks per sweep lived in SGPRs across the loop and spilled 250 of them.)
__device__ __forceinline__ double xor1(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ c128 xor1(c128 z) { return cmake(xor1(z.x), xor1(z.y)); }

// where the content of position m comes from when the cycle advances: m <- RR_SRC[m]
constexpr unsigned RR_SRC = 0x0u | (3u << 4) | (1u << 8) | (5u << 12) | (2u << 16) | (7u << 20) |
                            (4u << 24) | (6u << 28);
__host__ __device__ constexpr int rr_src(int m) { return (int)((RR_SRC >> (4 * m)) & 7u); }

// z[p], z[q] <- (cs z[p] - conj(su) z[q], cs z[q] + su z[p]): columns p, q of Z J for the row z
__device__ __forceinline__ void col_rot(c128 (&z)[8], int p, int q, double cs, c128 su) {
  const c128 zp = z[p], zq = z[q];
  c128 np = cscale(zp, cs);
  cfms(np, cconj(su), zq);
  c128 nq = cscale(zq, cs);
  cfma(nq, su, zp);
  z[p] = np;
  z[q] = nq;
}

__device__ __forceinline__ void advance_columns(c128 (&z)[8]) {
  c128 t[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) t[m] = z[rr_src(m)];
#pragma unroll
  for (int m = 0; m < 8; ++m) z[m] = t[m];
}

template <bool WITHW>
__device__ __forceinline__ void jacobi_round(c128 (&a)[8], double &dg, c128 (&w)[8], int r) {
  const bool lo = (r & 1) == 0;
  const double dpar = xor1(dg);
  // A[p][q], p < q: the even lane's copy, in both lanes of the pair (the odd lane's A[q][p] is its
  // conjugate only up to rounding, and the two must agree on the rotation to the last bit)
  const c128 e = sel(a, r ^ 1);
  const c128 ex = xor1(e);
  const c128 apq = lo ? e : ex;
  const double app = lo ? dg : dpar, aqq = lo ? dpar : dg;
  // the rotation of herm_packed.hpp's hp_jacobi_rows
  // (reciprocals and square roots by v_rcp / v_rsq + Newton: the IEEE sequences were 2/3 of a
  //  round's instructions; |apq|^2 outside [1e-300, 1e300] is left alone / takes the slow forms)
  const double mag2 = cabs2(apq);
  const bool tiny = !(mag2 >= 1e-300);
  const bool huge = mag2 > 1e300;
  const double inv = tiny ? 0.0 : (huge ? 1.0 / sqrt(mag2) : rsq2(mag2));
  const double mag = mag2 * inv;
  const c128 u = tiny ? cmake(1.0, 0.0) : cmake(apq.x * inv, apq.y * inv);
  const double tau = tiny ? 0.0 : (aqq - app) * 0.5 * inv;
  const double atau = fabs(tau);
  const bool far = !(atau < 1e100);  // (1 + tau^2 would overflow: t = 1 / (2 tau))
  const double root = far ? atau : (1.0 + tau * tau) * rsq2(1.0 + tau * tau);
  const double tabs = far ? (atau < 1.7e308 ? 1.0 / (atau + root) : 0.0) : rcp2(atau + root);
  const double t = tiny ? 0.0 : (tau >= 0.0 ? tabs : -tabs);
  const double cs = rsq2(1.0 + t * t);
  const double sn = t * cs;
  const c128 su = cmake(sn * u.x, sn * u.y);
  const double tm = t * mag;
  // A J (and W J): every row takes the column rotations of all four pairs
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double csk = shfl8(cs, 2 * k);
    const c128 suk = shfl8(su, 2 * k);
    col_rot(a, 2 * k, 2 * k + 1, csk, suk);
    if (WITHW) col_rot(w, 2 * k, 2 * k + 1, csk, suk);
  }
  dg = lo ? app - tm : aqq + tm;
  // J^H (A J): rows p and q mix; the entries in columns p, q of those rows are the rotated 2 x 2
  // block, known in closed form (dg, and zero off the diagonal)
  const c128 sul = lo ? cmake(-su.x, -su.y) : cconj(su);  // row p: cs a - su o; row q: cs a + conj(su) o
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const c128 o = xor1(a[c]);
    c128 n = cscale(a[c], cs);
    cfma(n, sul, o);
    a[c] = n;
  }
  put(a, r ^ 1, cmake(0.0, 0.0));
  // the cycle advances: rows to their new lanes, columns to their new registers
  const int from = (int)((RR_SRC >> (4 * r)) & 7u);
  dg = shfl8(dg, from);
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = shfl8(a[c], from);
  advance_columns(a);
  if (WITHW) advance_columns(w);
}

// A = J diag(lam) J^H.  In: row r of A in `a` (a[r] is ignored), its diagonal entry in `dg`, row r
// of W.  Out: dg = lam_r (the eigenvalue whose eigenvector is column r of J), W <- W J.
template <bool WITHW>
__device__ __forceinline__ void jacobi(c128 (&a)[8], double &dg, c128 (&w)[8], int r) {
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    put(a, r, cmake(0.0, 0.0));
    double off = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) off += cabs2(a[c]);
    off = sum8(off);
    const double diag = sum8(dg * dg);
    // (off-diagonal norm below 1e-15 of the diagonal's: the next sweep would square that; the
    //  lane-per-matrix kernels ask for 1e-17 and pay a sweep for it)
    if (__all(off <= 1e-30 * diag)) break;
    // (not unrolled: the seven copies push the kernel 177 registers over the file; the column
    //  moves cost 56 of a round's ~330 instructions)
#pragma unroll 1
    for (int round = 0; round < 7; ++round) {
      jacobi_round<WITHW>(a, dg, w, r);
    }
  }
}

// row r of W diag(f) W^H; the slot is used for the exchange (callers wsync() around other uses)
__device__ __forceinline__ void rebuild(const c128 (&w)[8], double f_own, c128 *X, int r,
                                        c128 (&out)[8]) {
  c128 ra[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ra[k] = cscale(w[k], shfl8(f_own, k));
  wsync();
  store_row(X, r, w);
  wsync();
  mul_rows_adj(ra, X, out);
  put(out, r, cmake(sel(out, r).x, 0.0));
}

}  // namespace rows8
}  // namespace ssspy
