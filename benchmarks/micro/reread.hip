// Is an immediate second read of a block's own chunk served by the caches (L2 4 MB per XCD, 256 MB
// Infinity Cache) when every CU does it at once?  Each block walks chunks of `chunk` bytes and
// reads each one `passes` times before moving on; the run is timed against the single-pass walk.
// Decides whether fusing two per-bin passes over X (covariance -> demixing filter -> next basis
// statistics) could save an HBM pass.
// Build: hipcc --offload-arch=gfx950 -O3 reread.hip -o reread
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ __launch_bounds__(256) void k_walk(const double2 *__restrict__ x, double *out,
                                              size_t n_chunks, size_t chunk_elems, int passes) {
  double acc = 0.0;
  for (size_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const double2 *p = x + c * chunk_elems;
    for (int s = 0; s < passes; ++s) {
      for (size_t i = threadIdx.x; i < chunk_elems; i += blockDim.x) {
        const double2 v = p[i];
        acc += v.x + v.y * (s + 1);
      }
      __syncthreads();
    }
  }
  if (acc == 12345.678) out[0] = acc;
}

int main() {
  const size_t n = (size_t)1 << 28;  // 4 GiB of double2
  double2 *x;
  double *out;
  hipMalloc(&x, n * 16);
  hipMalloc(&out, 8);
  hipMemset(x, 1, n * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    for (size_t chunk_kb : {64, 128, 256, 512}) {
      const size_t chunk_elems = chunk_kb * 1024 / 16, n_chunks = n / chunk_elems;
      float ms[3];
      for (int passes = 1; passes <= 2; ++passes) {
        hipLaunchKernelGGL(k_walk, dim3(blocks), dim3(256), 0, 0, x, out, n_chunks, chunk_elems, passes);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r)
          hipLaunchKernelGGL(k_walk, dim3(blocks), dim3(256), 0, 0, x, out, n_chunks, chunk_elems, passes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[passes], e0, e1);
        ms[passes] /= 3;
      }
      printf("blocks %5d chunk %4zu KB (working set %6.1f MB): 1 pass %.3f ms (%.2f TB/s), 2 passes %.3f ms -> second pass costs %.2fx of the first\n",
             blocks, chunk_kb, blocks * chunk_kb / 1024.0, ms[1], n * 16 / ms[1] / 1e9, ms[2],
             (ms[2] - ms[1]) / ms[1]);
    }
  }
  return 0;
}
