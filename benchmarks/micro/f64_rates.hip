// Microbenchmark: issue rates of the fp64 matrix core and the fp64 VALU on gfx950, and whether
// one wave overlaps them.  Build: hipcc --offload-arch=gfx950 -O3 f64_rates.hip -o f64_rates
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__global__ void k_mfma(double *out, int iters, double a, double b) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void k_mfma_dep(double *out, int iters, double a, double b) {
  double4_t c0 = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0];
}

__global__ void k_fma(double *out, int iters, double a, double b) {
  double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
      x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_rcp(double *out, int iters, double a) {
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1);
      x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

// one MFMA followed by NV independent FMAs, fenced, 4 accumulators
template <int NV>
__global__ void k_mix(double *out, int iters, double a, double b) {
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) x[u] = a + u;
  for (int i = 0; i < iters; ++i) {
#define STEP(cc)                                                 \
  cc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cc, 0, 0, 0);  \
  FENCE();                                                       \
  _Pragma("unroll") for (int u = 0; u < NV; ++u) x[u & 7] = fma(x[u & 7], a, b); \
  FENCE();
    STEP(c0) STEP(c1) STEP(c2) STEP(c3)
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3];
#pragma unroll
  for (int u = 0; u < 8; ++u) s += x[u];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double *out;
  hipMalloc(&out, 1 << 24);
  const int iters = 2000;
  for (int wpb = 1; wpb <= 2; ++wpb) {  // waves per SIMD: 256 CUs x 4 SIMDs x wpb
    const int blocks = 256 * wpb, threads = 256;
    const double nw = 256.0 * 4 * wpb;
    auto rep = [&](const char *name, double ms, double per_wave_ops) {
      // cycles per op per SIMD assuming 2.4 GHz
      printf("%-28s waves/SIMD=%d  %8.3f ms  -> %.1f cycles/op/SIMD @2.4GHz\n", name, wpb, ms,
             ms * 1e-3 * 2.4e9 / (per_wave_ops * wpb));
      (void)nw;
    };
    rep("mfma_f64 4 indep chains", time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); }), 4.0 * iters);
    rep("mfma_f64 1 dependent chain", time_ms([&] { hipLaunchKernelGGL(k_mfma_dep, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); }), 4.0 * iters);
    rep("v_fma_f64 8 indep", time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); }), 32.0 * iters);
    rep("v_rcp_f64 4 indep", time_ms([&] { hipLaunchKernelGGL(k_rcp, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.5); }), 32.0 * iters);
    rep("mfma + 8 fma (per mfma)", time_ms([&] { hipLaunchKernelGGL((k_mix<8>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); }), 4.0 * iters);
    rep("mfma + 16 fma (per mfma)", time_ms([&] { hipLaunchKernelGGL((k_mix<16>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); }), 4.0 * iters);
    rep("mfma + 32 fma (per mfma)", time_ms([&] { hipLaunchKernelGGL((k_mix<32>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 2.0); }), 4.0 * iters);
  }
  return 0;
}
