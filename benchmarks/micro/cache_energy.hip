// Energy per byte of a read-only stream out of the 256 MB Infinity Cache against out of HBM (round 5,
// round-4 verdict item 6b): the same grid-stride 16-byte read kernel over a working set of <ws> MB,
// re-read back to back for <seconds>; benchmarks/cache_energy.py samples the socket power meanwhile.
// Build: hipcc --offload-arch=gfx950 -O3 cache_energy.hip -o bin/cache_energy
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ x, double *out, size_t n) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = x[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? (size_t)atoll(argv[1]) : 192;
  const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
  const int blocks = argc > 3 ? atoi(argv[3]) : 2048;
  const size_t n = mb * (1u << 20) / 16;
  double2 *x;
  double *out;
  hipMalloc(&x, n * 16);
  hipMalloc(&out, 8);
  hipMemset(x, 1, n * 16);
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  size_t launches = 0;
  double elapsed = 0.0;
  while (elapsed < seconds) {
    for (int r = 0; r < 64; ++r) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, x, out, n);
    hipDeviceSynchronize();
    launches += 64;
    elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  printf("{\"working_set_mb\": %zu, \"seconds\": %.3f, \"tb_per_s\": %.3f}\n", mb, elapsed,
         (double)launches * n * 16 / elapsed / 1e12);
  return 0;
}
