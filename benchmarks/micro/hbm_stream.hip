// Achievable HBM bandwidth on this box: read-only sum, copy, and read+half-write (the mix of a
// pass that reads X and writes an 8-byte value per 16-byte element).
// Build: hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o hbm_stream
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ x, double *out, size_t n) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = x[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ x, double2 *__restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}

__global__ __launch_bounds__(256) void k_read_halfwrite(const double2 *__restrict__ x, double *__restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = x[i];
    y[i] = v.x * v.x + v.y * v.y;
  }
}

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const size_t n = (size_t)1 << 28;  // 2^28 double2 = 4 GiB
  double2 *x, *y;
  double *out;
  hipMalloc(&x, n * 16);
  hipMalloc(&y, n * 16);
  hipMalloc(&out, 8);
  hipMemset(x, 1, n * 16);
  for (int blocks : {1024, 2048, 8192, 65536}) {
    const double r = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, x, out, n); }, 5);
    const double c = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, x, y, n); }, 5);
    const double h = time_ms([&] { hipLaunchKernelGGL(k_read_halfwrite, dim3(blocks), dim3(256), 0, 0, x, (double *)y, n); }, 5);
    printf("blocks %6d: read %.2f TB/s, copy %.2f TB/s (r+w), read + half write %.2f TB/s (r+w)\n", blocks,
           n * 16 / r / 1e9, 2.0 * n * 16 / c / 1e9, 1.5 * n * 16 / h / 1e9);
  }
  return 0;
}
