// Generalised 2x2 Hermitian eigenproblem A z = lamb B z (type 1) for one lane.
// ref: ssspy/linalg/eigh.py:164-207 with inv = inv2: L = chol(B), C = L^-1 A L^-H, eigh(C),
// z = L^-H y.  A 2x2 Hermitian C is diagonalised exactly by one complex Jacobi rotation.
#pragma once

#include "common.hpp"

namespace ssspy {

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
  const c128 m00 = cscale(A[0][0], i00), m01 = cscale(A[0][1], i00);
  const c128 m10 = cadd(cmul(i10, A[0][0]), cscale(A[1][0], i11));
  const c128 m11 = cadd(cmul(i10, A[0][1]), cscale(A[1][1], i11));
  const double c00 = m00.x * i00;
  const c128 c01a = cadd(cmulc(m00, i10), cscale(m01, i11));
  const c128 c10a = cscale(m10, i00);
  const double c11 = cadd(cmulc(m10, i10), cscale(m11, i11)).x;
  // Hermitise (numpy.linalg.eigh reads one triangle; the two agree to rounding)
  const c128 c01 = cmake(0.5 * (c01a.x + c10a.x), 0.5 * (c01a.y - c10a.y));
  // one Jacobi rotation: tan(2 theta) from (c11 - c00) / (2 |c01|)
  const double mag2 = cabs2(c01);
  const double mag = sqrt(mag2);
  const bool tiny = mag2 < 1e-300;
  const double inv = tiny ? 0.0 : 1.0 / mag;
  const c128 u = tiny ? cmake(1.0, 0.0) : cmake(c01.x * inv, c01.y * inv);
  const double tau = tiny ? 0.0 : (c11 - c00) * 0.5 * inv;
  const double t = tiny ? 0.0 : ((tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau)));
  const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
  const double e0 = c00 - t * mag, e1 = c11 + t * mag;
  // eigenvectors (columns of J): y_a = (cs, -sn conj(u)) for e0, y_b = (sn u, cs) for e1
  const c128 ya[2] = {cmake(cs, 0.0), cmake(-sn * u.x, sn * u.y)};
  const c128 yb[2] = {cmake(sn * u.x, sn * u.y), cmake(cs, 0.0)};
  const bool swap = e1 < e0;
  lamb[0] = swap ? e1 : e0;
  lamb[1] = swap ? e0 : e1;
  const c128 y0[2] = {swap ? yb[0] : ya[0], swap ? yb[1] : ya[1]};
  const c128 y1[2] = {swap ? ya[0] : yb[0], swap ? ya[1] : yb[1]};
  // z = L^-H y,  L^-H = [[i00, conj(i10)], [0, i11]]
  const c128 i01 = cconj(i10);
  z[0][0] = cadd(cscale(y0[0], i00), cmul(i01, y0[1]));
  z[1][0] = cscale(y0[1], i11);
  z[0][1] = cadd(cscale(y1[0], i00), cmul(i01, y1[1]));
  z[1][1] = cscale(y1[1], i11);
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
