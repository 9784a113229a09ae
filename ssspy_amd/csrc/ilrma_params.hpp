// Parameters shared by the ILRMA launchers (ilrma_api.hip) and kernels (ilrma_kernels.hip).
#pragma once

#include "common.hpp"

namespace ssspy {

struct IlrmaDims {
  int B, F, T, K;
  double p;       // domain
  int model;      // SSSPY_SOURCE_GAUSS / _T / _GGD
  int me;         // 1: "ME" source updates (same sums, exponent 1; ref: ssspy/bss/ilrma.py:1249-1401)
  double mparam;  // dof (t) or beta (GGD)
  int floor_kind;
  double floor_eps;
  int raw;  // basis kernel: 1 = store the (num, den) sums as pairs instead of applying the update
};

// a = numerator factor of the MM update, b = 1/R          (R = (T V)_nij, P = |y_nij|^2)
//   Gauss: a = P / R^((p+2)/p)                            ref: ssspy/bss/ilrma.py:1116-1125
//   t    : a = P / (R~ R), R~ = nu/(nu+2) R^(2/p) + 2/(nu+2) P        ref: :2505-2518
//   GGD  : a = (beta/2) P^(beta/2) / R^((beta+p)/p)                    ref: :3810-3821
__device__ __forceinline__ void mm_weights(double P, double R, const IlrmaDims &d, bool valid,
                                           double &a, double &b) {
  const double rinv = 1.0 / R;
  double aa;
  if (d.model == SSSPY_SOURCE_GAUSS) {
    aa = (d.p == 2.0) ? P * rinv * rinv : P / pow(R, (d.p + 2.0) / d.p);
  } else if (d.model == SSSPY_SOURCE_T) {
    const double w = d.mparam / (d.mparam + 2.0);
    const double r2p = (d.p == 2.0) ? R : pow(R, 2.0 / d.p);
    aa = P / ((w * r2p + (1.0 - w) * P) * R);
  } else {
    aa = 0.5 * d.mparam * pow(P, 0.5 * d.mparam) / pow(R, (d.mparam + d.p) / d.p);
  }
  a = valid ? aa : 0.0;
  b = valid ? rinv : 0.0;
}

__device__ __forceinline__ double mm_ratio_pow(double num, double den, const IlrmaDims &d) {
  const double ratio = num / den;
  if (d.me) return ratio;
  if (d.model == SSSPY_SOURCE_GGD) return pow(ratio, d.p / (d.mparam + d.p));
  return (d.p == 2.0) ? sqrt(ratio) : pow(ratio, d.p / (d.p + 2.0));
}

// 1 / x by v_rcp_f64 + two Newton steps (~1 ulp: the recipe of fast_tiles.hpp's rcp_nr) where the
// weight pass of the wide covariance forms one reciprocal per (source, bin, frame) -- the IEEE
// divide is ~40 instructions, 68 of that pass's 278 us at 16 mixtures of 8 sources; zero, Inf, NaN
// and the ends of the range keep the divide
// (a call, so that the compiler cannot if-convert the rare branch into a divide for every lane)
__device__ __noinline__ double recip_ieee(double x) { return 1.0 / x; }
__device__ __forceinline__ double recip_weight(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  if (__builtin_expect(!(x > 1e-290 && x < 1e290), 0)) r = recip_ieee(x);
  return r;
}

// varphi = 1 / R~ of the spatial update (ref: :1494-1498, :2915-2935, :3987-4011)
__device__ __forceinline__ double spatial_weight(double P, double R, const IlrmaDims &d) {
  if (d.model == SSSPY_SOURCE_GAUSS) return (d.p == 2.0) ? recip_weight(R) : 1.0 / pow(R, 2.0 / d.p);
  if (d.model == SSSPY_SOURCE_T) {
    const double w = d.mparam / (d.mparam + 2.0);
    const double r2p = (d.p == 2.0) ? R : pow(R, 2.0 / d.p);
    return 1.0 / (w * r2p + (1.0 - w) * P);
  }
  const double y2b = apply_floor(pow(P, 0.5 * (2.0 - d.mparam)), d.floor_kind, d.floor_eps);
  return 1.0 / ((2.0 / d.mparam) * y2b * pow(R, d.mparam / d.p));
}

// per-(n,i,j) term of the negative log-likelihood (ref: :1956-1960, :3301-3305, :4377-4381)
__device__ __forceinline__ double loss_term(double P, double R, const IlrmaDims &d) {
  const double lr = (2.0 / d.p) * log(R);
  if (d.model == SSSPY_SOURCE_GAUSS) return ((d.p == 2.0) ? P / R : P / pow(R, 2.0 / d.p)) + lr;
  if (d.model == SSSPY_SOURCE_T) {
    const double r2p = (d.p == 2.0) ? R : pow(R, 2.0 / d.p);
    return (1.0 + 0.5 * d.mparam) * log(1.0 + (2.0 / d.mparam) * P / r2p) + lr;
  }
  return pow(P, 0.5 * d.mparam) / pow(R, d.mparam / d.p) + lr;
}

}  // namespace ssspy
