// Shared pieces of the MFMA tile kernels (ILRMA and FastMNMF): lane layout helpers and the
// GEMM1 that produces R = T V on a 16 x 16 (bin x frame) tile.  See ilrma_kernels.hip for the
// description of the two tile orientations.
#pragma once

#include "common.hpp"

namespace ssspy {

// frame permutation inside a bin-major tile: D row rho = q + 4r  <->  frame j0 + 4q + r
__device__ __forceinline__ int tile_pi(int rho) { return 4 * (rho & 3) + (rho >> 2); }

// bin-major GEMM1: returns R[bin i0+c, frame j0+4q+r] in register r of lane (c, q).
// tb[ks]: B operand basis[n, bin(c), 4ks+q] (0 when 4ks+q >= K), hoisted by the caller when
// KSMALL (K <= 16); otherwise the basis row is read here.
template <bool KSMALL>
__device__ __forceinline__ double4_t nmf_rt_tile(const double *__restrict__ Vn,      // V[b,n] (K,T)
                                                 const double *__restrict__ Tn_bin,  // T[b,n,bin,:]
                                                 const double (&tb)[4], int K, int T, int j0,
                                                 int c, int q) {
  double4_t R = {0.0, 0.0, 0.0, 0.0};
  const int jf = j0 + tile_pi(c);
  const bool fvalid = jf < T;
  const int jc = fvalid ? jf : T - 1;
  if (KSMALL) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks * 4 < K) {
        const int kk = ks * 4 + q;
        double a = Vn[(long long)(kk < K ? kk : K - 1) * T + jc];
        a = (kk < K && fvalid) ? a : 0.0;
        R = mfma_f64(a, tb[ks], R);
      }
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 4) {
      const int kk = k0 + q;
      const int kc = kk < K ? kk : K - 1;
      double a = Vn[(long long)kc * T + jc];
      a = (kk < K && fvalid) ? a : 0.0;
      double t = Tn_bin[kc];
      t = kk < K ? t : 0.0;
      R = mfma_f64(a, t, R);
    }
  }
  return R;
}

}  // namespace ssspy
