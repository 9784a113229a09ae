// Source models of the tuned ("fast" and "small") ILRMA pass kernels: the per-element factors of the
// multiplicative updates, the model parameters the launchers precompute, and the log-sum helper of
// the loss by-product.  Shared by ilrma_fast.hip and ilrma_small.hip.
#pragma once

#include "common.hpp"
#include "fast_tiles.hpp"

namespace ssspy {
namespace fast {

// Source models of the tuned kernels (R = (T V)_nij, P = |y_nij|^2; ref: ssspy/bss/ilrma.py):
//   FM_GAUSS  domain 2: a = P / R^2,           varphi = 1 / R                        (:1116-1125, :1494-1498)
//   FM_T      domain 2: a = P / (R~ R), varphi = 1 / R~, R~ = w R + (1 - w) P, w = nu / (nu + 2)
//                                                                              (:2505-2518, :2915-2935)
//   FM_GGD    domain 2: a = (beta / 2) (P / R)^(beta/2) / R,
//                       varphi = 1 / ((2 / beta) floor(P^((2 - beta)/2)) R^(beta/2))  (:3810-3821, :3987-4011)
//   FM_GAUSS1 domain 1: a = P / R^3,           varphi = 1 / R^2
//   FM_GAUSSP domain p: a = P / R^((p+2)/p),   varphi = 1 / R^(2/p)   (any 0 < p <= 2; powers as
//                       exp2(e log2 R): ~1e-14 relative, a third of the instructions of pow)
// `expo`: exponent of the (num / den) ratio: p / (p + 2), 1 for the ME updates, p / (beta + p) for GGD.
constexpr int FM_GAUSS = 0, FM_T = 1, FM_GGD = 2, FM_GAUSS1 = 3, FM_GAUSSP = 4;
struct FastModel {
  double w, w1, nu, beta, expo;
  double pinv2, e_num;  // FM_GAUSSP: 2 / p and -(p + 2) / p
  int floor_kind;
  double floor_eps;
};

// x^e for x >= 0 through exp2 / log2 (about 1e-14 relative over the dynamic range met here, and
// less than half the instructions of the correctly rounded pow)
// (e is uniform: beta = 1, the Laplace-like GGD, needs nothing but square roots)
__device__ __forceinline__ double pow_nonneg(double x, double e) {
  if (e == 0.5) return sqrt_nr(x);
  if (e == 0.25) return sqrt_nr(sqrt_nr(x));  // (beta = 1/2)
  return x > 0.0 ? exp2(e * log2(x)) : 0.0;
}

__device__ __forceinline__ double ratio_pow(double ratio, double expo) {
  if (expo == 0.5) return sqrt(ratio);
  if (expo == 1.0) return ratio;
  return pow(ratio, expo);
}

// sum of log(x_i) without a log per value: the running product of the mantissas (each in [0.5, 1))
// and the integer sum of the exponents; the mantissa product is renormalised before it can
// underflow (every 16 factors is ample: >= 2^-16), one log at the very end.  2 VALU instructions per
// value instead of the ~40 of an fp64 log; the rounding of the product adds ~1e-16 per factor to a
// sum of logs of magnitude >= 1.
struct LogSum {
  double mant;
  int expo;
  __device__ __forceinline__ void clear() {
    mant = 1.0;
    expo = 0;
  }
  __device__ __forceinline__ void mul(double x) {
    mant *= __builtin_amdgcn_frexp_mant(x);
    expo += __builtin_amdgcn_frexp_exp(x);
  }
  __device__ __forceinline__ void renorm() {
    expo += __builtin_amdgcn_frexp_exp(mant);
    mant = __builtin_amdgcn_frexp_mant(mant);
  }
  __device__ __forceinline__ double value() const {
    return log(mant) + 0.6931471805599453094 * (double)expo;
  }
};

// numerator factor a of the multiplicative updates (rinv = 1 / R)
template <int MODEL>
__device__ __forceinline__ double mm_num_factor(double pw, double R, double rinv,
                                                const FastModel &fm);
template <>
__device__ __forceinline__ double mm_num_factor<FM_GAUSS>(double pw, double, double rinv,
                                                          const FastModel &) {
  return pw * rinv * rinv;
}
template <>
__device__ __forceinline__ double mm_num_factor<FM_T>(double pw, double R, double rinv,
                                                      const FastModel &fm) {
  return pw * rcp_nr(fma(fm.w, R, fm.w1 * pw)) * rinv;
}
template <>
__device__ __forceinline__ double mm_num_factor<FM_GGD>(double pw, double, double rinv,
                                                        const FastModel &fm) {
  return 0.5 * fm.beta * pow_nonneg(pw * rinv, 0.5 * fm.beta) * rinv;
}
template <>
__device__ __forceinline__ double mm_num_factor<FM_GAUSS1>(double pw, double, double rinv,
                                                           const FastModel &) {
  return pw * rinv * rinv * rinv;
}
template <>
__device__ __forceinline__ double mm_num_factor<FM_GAUSSP>(double pw, double R, double,
                                                           const FastModel &fm) {
  return pw * pow_nonneg(R, fm.e_num);
}

// fmodel: FM_*; mparam: dof (t) / beta (GGD); me: exponent 1 instead of p / (p + 2)
static inline FastModel make_fast_model(int fmodel, double mparam, int me, int floor_kind = 0,
                                        double floor_eps = 0.0) {
  FastModel fm;
  const bool t = fmodel == FM_T;
  fm.nu = t ? mparam : 1.0;
  fm.w = t ? mparam / (mparam + 2.0) : 1.0;
  fm.w1 = 1.0 - fm.w;
  fm.beta = fmodel == FM_GGD ? mparam : 2.0;
  // (FM_GAUSSP: the domain travels in mparam)
  const double p = fmodel == FM_GAUSS1 ? 1.0 : (fmodel == FM_GAUSSP ? mparam : 2.0);
  fm.pinv2 = 2.0 / p;
  fm.e_num = -(p + 2.0) / p;
  fm.expo = me ? 1.0 : (fmodel == FM_GGD ? p / (fm.beta + p) : p / (p + 2.0));
  fm.floor_kind = floor_kind;
  fm.floor_eps = floor_eps;
  return fm;
}


}  // namespace fast
}  // namespace ssspy
