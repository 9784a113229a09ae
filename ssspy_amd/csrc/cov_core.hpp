// Weighted spatial covariance accumulation shared by the ILRMA / AuxIVA / MNMF kernels.
//
// Lane layout of one wavefront ("bin-major tile"): lane = q*16 + c, c = bin inside a tile
// of 16 consecutive bins, q = one of 4 frame sub-groups; in each step lane (c, q) owns the
// 4 consecutive frames j0 + 4q + {0,1,2,3} of bin i0 + c, i.e. 64 contiguous bytes of every
// channel row of the (N, F, T) complex128 tensor.  A lane therefore accumulates the
// Hermitian N x N statistics of ONE bin privately over all of its frames; the only
// cross-lane traffic is the final 4-way fold over q (two shuffles) and the cross-wave fold
// through LDS.  This is the same layout the f64 MFMA produces for R^T = (T V)^T
// (columns = bins, rows = frames), so the NMF-weighted variant chains into it without any
// data movement.
#pragma once

#include "common.hpp"

namespace ssspy {

// Upper triangle of S Hermitian N x N accumulators: diagonal real, off-diagonal complex.
template <int N, int S>
struct CovAcc {
  double diag[S][N];
  c128 off[S][(N * (N - 1)) / 2 > 0 ? (N * (N - 1)) / 2 : 1];

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int a = 0; a < N; ++a) diag[s][a] = 0.0;
#pragma unroll
      for (int e = 0; e < (N * (N - 1)) / 2; ++e) off[s][e] = cmake(0.0, 0.0);
    }
  }

  // acc[s] += phi[s] * x x^H for one frame
  __device__ __forceinline__ void add(const c128 (&x)[N], const double (&phi)[S]) {
    int e = 0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
      const double p = cabs2(x[a]);
#pragma unroll
      for (int s = 0; s < S; ++s) diag[s][a] = fma(phi[s], p, diag[s][a]);
#pragma unroll
      for (int b = a + 1; b < N; ++b) {
        const c128 z = cmulc(x[a], x[b]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
          off[s][e].x = fma(phi[s], z.x, off[s][e].x);
          off[s][e].y = fma(phi[s], z.y, off[s][e].y);
        }
        ++e;
      }
    }
  }

  // fold the 4 frame sub-groups (lanes c, c+16, c+32, c+48) together
  __device__ __forceinline__ void fold_q() {
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int a = 0; a < N; ++a) {
        double v = diag[s][a];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        diag[s][a] = v;
      }
#pragma unroll
      for (int e = 0; e < (N * (N - 1)) / 2; ++e) {
        double vx = off[s][e].x, vy = off[s][e].y;
        vx += __shfl_xor(vx, 16, 64);
        vx += __shfl_xor(vx, 32, 64);
        vy += __shfl_xor(vy, 16, 64);
        vy += __shfl_xor(vy, 32, 64);
        off[s][e] = cmake(vx, vy);
      }
    }
  }
};

// Number of doubles one wave parks in LDS for the cross-wave fold: S*N*N values x 16 bins.
template <int N, int S>
constexpr int cov_lds_doubles_per_wave() {
  return S * N * N * 16;
}

// Cross-wave fold + store.  Every wave has called fold_q().  `lds` holds
// nwaves * cov_lds_doubles_per_wave() doubles.  Output element (bin, s, a, b) goes to
// U[((bin_global) * S_total + s0 + s) * N * N + a * N + b], scaled by `scale`.
template <int N, int S>
__device__ __forceinline__ void cov_reduce_store(const CovAcc<N, S> &acc, double *lds,
                                                 c128 *__restrict__ U, long long bin_base,
                                                 int i0, int F, int S_total, int s0, int s_count,
                                                 double scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  constexpr int PER = S * N * N;  // doubles per bin: [s][a][b] as (re, im) of upper triangle
  // layout in LDS: [wave][value v in 0..PER)][bin c]; value index v = s*N*N + slot, slots:
  // 0..N-1 diag, then pairs (re, im) of the off-diagonals
  double *mine = lds + (size_t)wave * PER * 16;
  if (q == 0) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int a = 0; a < N; ++a) mine[(s * N * N + a) * 16 + c] = acc.diag[s][a];
#pragma unroll
      for (int e = 0; e < (N * (N - 1)) / 2; ++e) {
        mine[(s * N * N + N + 2 * e) * 16 + c] = acc.off[s][e].x;
        mine[(s * N * N + N + 2 * e + 1) * 16 + c] = acc.off[s][e].y;
      }
    }
  }
  __syncthreads();
  // one thread per (bin, s, a, b >= a) output pair
  constexpr int NPAIR = (N * (N + 1)) / 2;
  for (int t = threadIdx.x; t < 16 * S * NPAIR; t += blockDim.x) {
    const int cb = t & 15;
    int rest = t >> 4;
    const int s = rest / NPAIR;
    int pr = rest % NPAIR;
    // decode pair index -> (a, b): diagonal first, then off-diagonals in row-major order
    int a, b, slot_re, slot_im;
    if (pr < N) {
      a = b = pr;
      slot_re = pr;
      slot_im = -1;
    } else {
      int e = pr - N, aa = 0, rem = e;
      while (rem >= N - 1 - aa) {
        rem -= N - 1 - aa;
        ++aa;
      }
      a = aa;
      b = aa + 1 + rem;
      slot_re = N + 2 * e;
      slot_im = slot_re + 1;
    }
    double re = 0.0, im = 0.0;
    for (int w = 0; w < nwaves; ++w) {
      const double *src = lds + (size_t)w * PER * 16;
      re += src[(s * N * N + slot_re) * 16 + cb];
      if (slot_im >= 0) im += src[(s * N * N + slot_im) * 16 + cb];
    }
    const int bin = i0 + cb;
    if (bin < F && s < s_count) {
      c128 *dst = U + ((bin_base + bin) * S_total + (s0 + s)) * (long long)(N * N);
      dst[a * N + b] = cmake(re * scale, im * scale);
      if (a != b) dst[b * N + a] = cmake(re * scale, -im * scale);
    }
  }
}

}  // namespace ssspy
