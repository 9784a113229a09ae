// Generalised 2x2 Hermitian eigenproblem A z = lamb B z (type 1) for one lane.
// ref: ssspy/linalg/eigh.py:164-207 with inv = inv2: L = chol(B), C = L^-1 A L^-H, eigh(C),
// z = L^-H y.
#pragma once

#include "common.hpp"

namespace ssspy {

// Eigen-decomposition of the 2 x 2 Hermitian matrix [[c00, conj(b)], [b, c11]] WITH THE PHASES
// np.linalg.eigh GIVES ITS EIGENVECTORS.  The reference solves its pairwise problems with
// np.linalg.eigh on the 2 x 2 matrix C (ssspy/linalg/eigh.py:198), and without scale restoration
// the phase of those eigenvectors stays in every (source, bin) row of the separated output, so
// "the same results" includes LAPACK's convention.  What zheevd does at n = 2 (reference LAPACK, the
// one NumPy's OpenBLAS carries), restated:
//   zhetd2, lower triangle: the Householder "reflector" of the one subdiagonal entry b is the phase
//     q = b / beta with beta = -sign(|b|, Re b) (zlarfg; b real: none, beta = b), leaving the real
//     tridiagonal matrix [[c00, beta], [beta, c11]];
//   zsteqr: a subdiagonal below eps sqrt(|c00| |c11|) is dropped (vectors = identity); otherwise
//     dlaev2 gives (rt1, rt2) and the rotation (cs1, sn1), applied to the identity: columns
//     (cs1, sn1) for rt1 and (-sn1, cs1) for rt2; then the eigenvalues are sorted ascending;
//   zunmtr: row 2 of the vectors times q.
// lamb ascending; y[r][k] = component r of eigenvector k.
__device__ __forceinline__ void eigh2_lapack(double c00, double c11, c128 b, double (&lamb)[2],
                                             c128 (&y)[2][2]) {
  // zlarfg on the single entry b
  double e = b.x;
  c128 q = cmake(1.0, 0.0);
  if (b.y != 0.0) {
    const double beta = -copysign(hypot(b.x, b.y), b.x);
    q = cmake(b.x / beta, b.y / beta);
    e = beta;
  }
  double rt1 = c00, rt2 = c11, cs1 = 1.0, sn1 = 0.0;
  const double eps = 1.1102230246251565e-16;  // dlamch('E')
  if (fabs(e) > sqrt(fabs(c00)) * sqrt(fabs(c11)) * eps) {
    // dlaev2(a = c00, b = e, c = c11)
    const double a = c00, c = c11;
    const double sm = a + c, df = a - c, adf = fabs(df), tb = e + e, ab = fabs(tb);
    const double acmx = fabs(a) > fabs(c) ? a : c, acmn = fabs(a) > fabs(c) ? c : a;
    double rt;
    if (adf > ab) {
      const double r = ab / adf;
      rt = adf * sqrt(1.0 + r * r);
    } else if (adf < ab) {
      const double r = adf / ab;
      rt = ab * sqrt(1.0 + r * r);
    } else {
      rt = ab * 1.4142135623730951;
    }
    int sgn1;
    if (sm < 0.0) {
      rt1 = 0.5 * (sm - rt);
      sgn1 = -1;
      rt2 = (acmx / rt1) * acmn - (e / rt1) * e;
    } else if (sm > 0.0) {
      rt1 = 0.5 * (sm + rt);
      sgn1 = 1;
      rt2 = (acmx / rt1) * acmn - (e / rt1) * e;
    } else {
      rt1 = 0.5 * rt;
      rt2 = -0.5 * rt;
      sgn1 = 1;
    }
    double cs;
    int sgn2;
    if (df >= 0.0) {
      cs = df + rt;
      sgn2 = 1;
    } else {
      cs = df - rt;
      sgn2 = -1;
    }
    if (fabs(cs) > ab) {
      const double ct = -tb / cs;
      sn1 = 1.0 / sqrt(1.0 + ct * ct);
      cs1 = ct * sn1;
    } else if (ab == 0.0) {
      cs1 = 1.0;
      sn1 = 0.0;
    } else {
      const double tn = -cs / tb;
      cs1 = 1.0 / sqrt(1.0 + tn * tn);
      sn1 = tn * cs1;
    }
    if (sgn1 == sgn2) {
      const double tn = cs1;
      cs1 = -sn1;
      sn1 = tn;
    }
  }
  // columns (cs1, sn1) | (-sn1, cs1), sorted ascending (one swap), row 2 times q
  const bool swap = rt2 < rt1;
  lamb[0] = swap ? rt2 : rt1;
  lamb[1] = swap ? rt1 : rt2;
  const double v00 = swap ? -sn1 : cs1, v10 = swap ? cs1 : sn1;
  const double v01 = swap ? cs1 : -sn1, v11 = swap ? sn1 : cs1;
  y[0][0] = cmake(v00, 0.0);
  y[0][1] = cmake(v01, 0.0);
  y[1][0] = cmake(q.x * v10, q.y * v10);
  y[1][1] = cmake(q.x * v11, q.y * v11);
}

// lamb ascending; z[r][k] = component r of eigenvector k.  Returns false if B is not PD.
__device__ __forceinline__ bool eigh2_type1(const c128 (&A)[2][2], const c128 (&Bm)[2][2],
                                            double (&lamb)[2], c128 (&z)[2][2]) {
  const double b00 = Bm[0][0].x, b11 = Bm[1][1].x;
  const c128 b10 = Bm[1][0];
  const double l00 = sqrt(b00);
  const c128 l10 = cmake(b10.x / l00, b10.y / l00);
  const double d = b11 - cabs2(l10);
  const double l11 = sqrt(d);
  const bool ok = (b00 > 0.0) && (d > 0.0);
  const double i00 = 1.0 / l00, i11 = 1.0 / l11;
  const c128 i10 = cmake(-l10.x * i00 * i11, -l10.y * i00 * i11);  // (L^-1)[1][0]
  // M1 = L^-1 A ;  C = M1 L^-H
  const c128 m00 = cscale(A[0][0], i00);
  const c128 m10 = cadd(cmul(i10, A[0][0]), cscale(A[1][0], i11));
  const c128 m11 = cadd(cmul(i10, A[0][1]), cscale(A[1][1], i11));
  const double c00 = m00.x * i00;
  const c128 c10 = cscale(m10, i00);  // (numpy.linalg.eigh reads the lower triangle)
  const double c11 = cadd(cmulc(m10, i10), cscale(m11, i11)).x;
  c128 y[2][2];
  eigh2_lapack(c00, c11, c10, lamb, y);
  // z = L^-H y,  L^-H = [[i00, conj(i10)], [0, i11]]
  const c128 i01 = cconj(i10);
  z[0][0] = cadd(cscale(y[0][0], i00), cmul(i01, y[1][0]));
  z[1][0] = cscale(y[1][0], i11);
  z[0][1] = cadd(cscale(y[0][1], i00), cmul(i01, y[1][1]));
  z[1][1] = cscale(y[1][1], i11);
  return ok;
}

// Re(h^H G h) for a 2-vector
__device__ __forceinline__ double quad2(const c128 (&h)[2], const c128 (&G)[2][2]) {
  double q = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    c128 t = cmake(0.0, 0.0);
#pragma unroll
    for (int b = 0; b < 2; ++b) cfma(t, G[a][b], h[b]);
    q += h[a].x * t.x + h[a].y * t.y;
  }
  return q;
}

}  // namespace ssspy
