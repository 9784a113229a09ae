// k_basis of fmnmf_generic.hip taken apart: V0 as shipped, V1 without the halving exchange, V2 without
// the LDS reads, V3 with neither (loads + FMAs only).  hipcc -O3 --offload-arch=gfx950 basis_parts.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int KT = 8;
constexpr int BASIS_TMAX = 1024;
// one step of the halving exchange: the lanes with bit OFF set keep val[C..2C), the others val[0..C),
// each adds what its partner at distance OFF held of the same values.  (A template per step: written
// as one loop over (C, OFF), hipcc left the loop rolled and indexed val[] through 884 v_cndmask --
// 85 of the kernel's 125 us.)
template <int C, int OFF>
__device__ __forceinline__ void halve(double (&val)[2 * KT], int lane) {
  const bool upper = (lane & OFF) != 0;
#pragma unroll
  for (int q = 0; q < C; ++q) {
    const double send = upper ? val[q] : val[q + C];
    const double keep = upper ? val[q + C] : val[q];
    val[q] = keep + __shfl_xor(send, OFF, 64);
  }
}
template <int V>
__global__ __launch_bounds__(256) void k_basis(double *basis, const double *__restrict__ act,
                                               const double *__restrict__ A,
                                               const double *__restrict__ Bt, int N, int F, int T,
                                               int K, int nb, int floor_kind, double eps) {
  extern __shared__ __attribute__((aligned(16))) double vtile[];  // [KT][T] when T <= BASIS_TMAX
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long bn = blockIdx.z;
  const int k0 = blockIdx.y * KT;
  const bool tiled = T <= BASIS_TMAX;
  if (tiled) {
    for (int k = 0; k < KT; ++k)
      for (int j = threadIdx.x; j < T; j += blockDim.x)
        vtile[k * T + j] = k0 + k < K ? act[(bn * K + k0 + k) * T + j] : 0.0;
    __syncthreads();
  }
  const int i_begin = blockIdx.x * nb + wave, i_end = min((int)(blockIdx.x + 1) * nb, F);
  if (i_begin >= i_end) return;
  const int chunks = (T + 255) / 256;  // of four runs of 64 frames
  const double *abase = A + bn * F * T, *btbase = Bt + bn * F * T;
  double av[4], bv[4];
  auto fetch = [&](int i, int c, double(&a4)[4], double(&b4)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = c * 256 + 64 * u + lane;
      const bool in = j < T && i < i_end;
      a4[u] = in ? abase[(long long)i * T + j] : 0.0;
      b4[u] = in ? btbase[(long long)i * T + j] : 0.0;
    }
  };
  fetch(i_begin, 0, av, bv);
  for (int i = i_begin; i < i_end; i += 4) {
    double val[2 * KT];  // sn[0..8), sd[0..8)
#pragma unroll
    for (int k = 0; k < 2 * KT; ++k) val[k] = 0.0;
    for (int c = 0; c < chunks; ++c) {
      double na[4], nbv[4];
      const bool last = c + 1 == chunks;
      fetch(last ? i + 4 : i, last ? 0 : c + 1, na, nbv);  // (zeros past the run's end)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(c * 256 + 64 * u + lane, T - 1);  // (beyond T the traces above are zero)
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const double vv = V == 2 || V == 3 ? 1.0 : tiled ? vtile[k * T + j]
                                  : (k0 + k < K ? act[(bn * K + k0 + k) * T + j] : 0.0);
          val[k] = fma(vv, av[u], val[k]);
          val[KT + k] = fma(vv, bv[u], val[KT + k]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = na[u];
        bv[u] = nbv[u];
      }
    }
    if (V != 1 && V != 3) {
    halve<8, 32>(val, lane);
    halve<4, 16>(val, lane);
    halve<2, 8>(val, lane);
    halve<1, 4>(val, lane);
    }
    double tot = val[0];  // of value (lane >> 2) & 15, summed over the lanes that differ in bits 5..2
    tot += __shfl_xor(tot, 2, 64);
    tot += __shfl_xor(tot, 1, 64);
    const double den = __shfl(tot, (lane + 32) & 63, 64);  // sd[k] sits 32 lanes above sn[k]
    const int k = lane >> 2;
    if ((lane & 3) == 0 && k < KT && k0 + k < K) {
      const long long o = (bn * F + i) * K + k0 + k;
      basis[o] = basis[o] * sqrt(tot / den);
    }
  }
}

template <int V>
float run(double *basis, double *act, double *A, double *Bt, int NB, int F, int T, int K, int nb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid((F + nb - 1) / nb, 1, NB);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_basis<V>, grid, dim3(256), KT * T * 8, 0, basis, act, A, Bt, 8, F, T, K, nb, 0, 0.0);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_basis<V>, grid, dim3(256), KT * T * 8, 0, basis, act, A, Bt, 8, F, T, K, nb, 0, 0.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}
int main() {
  const int NB = 64, F = 513, T = 256, K = 8;
  double *basis, *act, *A, *Bt;
  hipMalloc(&basis, NB * F * K * 8); hipMalloc(&act, NB * K * T * 8);
  hipMalloc(&A, (size_t)NB * F * T * 8 * 2); Bt = A + (size_t)NB * F * T;
  hipMemset(basis, 0, NB * F * K * 8); hipMemset(act, 0, NB * K * T * 8); hipMemset(A, 0, (size_t)NB * F * T * 16);
  for (int nb : {4, 8, 22, 44, 129}) {
    printf("nb %3d: V0 %.1f  V1 %.1f  V2 %.1f  V3 %.1f us\n", nb, run<0>(basis, act, A, Bt, NB, F, T, K, nb),
           run<1>(basis, act, A, Bt, NB, F, T, K, nb), run<2>(basis, act, A, Bt, NB, F, T, K, nb), run<3>(basis, act, A, Bt, NB, F, T, K, nb));
  }
  return 0;
}
