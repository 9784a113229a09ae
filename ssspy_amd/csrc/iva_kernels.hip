// AuxIVA (IP1 / ISS1) kernels: frame power r_nj^2 = sum_i |y_nij|^2, auxiliary weights and the
// contrast part of the loss.  The covariance / IP1 / ISS1 steps are the shared operators of
// spatial_kernels.hip.
#include "common.hpp"
#include "wide_n.hpp"

namespace ssspy {

// lanes along frames (coalesced 1 KB rows), W_i is wave-uniform.  grid: (ceil(T/64), bin chunks, B);
// the four waves of a block take a quarter of the chunk's bins each and fold through LDS in wave
// order.  One chunk: r2 is stored directly; several: every chunk stores its sums to its slab
// part[chunk][b][n][j] and k_fold_slabs (common.hpp) adds them in chunk order -- no fp64 atomics: the
// weights of the next iteration, hence the whole trajectory, are the same on every run.
template <int N>
__global__ __launch_bounds__(256) void k_iva_frame_power(const c128 *X,
                                                         const c128 *__restrict__ W, double *r2,
                                                         int F, int T, int bins_per_chunk,
                                                         c128 *Yout = nullptr) {
  // Yout (round 5, needs W; may be X itself): y = W x is also stored -- the separate() that precedes
  // the next iteration's weights and their frame-power pass in one walk (ssspy_separate_frame_power)
  __shared__ double fold[4][N][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int b = blockIdx.z;
  const int c_begin = blockIdx.y * bins_per_chunk;
  const int c_end = min(F, c_begin + bins_per_chunk);
  const int per_wave = (c_end - c_begin + 3) >> 2;
  const int i_begin = c_begin + wave * per_wave;
  const int i_end = min(c_end, i_begin + per_wave);
  const int jc = min(j, T - 1);
  double acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0.0;
#pragma unroll 2
  for (int i = i_begin; i < i_end; ++i) {
    c128 x[N];
#pragma unroll
    for (int m = 0; m < N; ++m) x[m] = X[(((long long)b * N + m) * F + i) * T + jc];
    if (W) {
      const c128 *Wi = W + ((long long)b * F + i) * (N * N);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) cfma(y, Wi[n * N + m], x[m]);
        acc[n] += cabs2(y);
        if (Yout && j < T) Yout[(((long long)b * N + n) * F + i) * T + j] = y;
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] += cabs2(x[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < N; ++n) fold[wave][n][lane] = acc[n];
  __syncthreads();
  // (r2 is the output itself with one chunk, else the chunk's slab of the partial buffer)
  double *dst = r2 + (long long)blockIdx.y * gridDim.z * N * T;
  for (int e = threadIdx.x; e < N * 64; e += 256) {
    const int n = e >> 6, ln = e & 63;
    const int jj = blockIdx.x * 64 + ln;
    if (jj < T)
      dst[((long long)b * N + n) * T + jj] =
          ((fold[0][n][ln] + fold[1][n][ln]) + fold[2][n][ln]) + fold[3][n][ln];
  }
}

// The same sums with a thread per frame (256 frames per block, no fold): the shape for large batches,
// where (T / 256) B blocks per bin chunk already fill the chip (128 mixtures: 2.03 -> 1.8 ms per
// AuxIVA-IP iteration against the 64-frame form above, which wins below ~64 mixtures).
// grid: (ceil(T/256), bin chunks, B)
template <int N>
__global__ __launch_bounds__(256) void k_iva_frame_power_wide(const c128 *X,
                                                              const c128 *__restrict__ W,
                                                              double *r2, int F, int T,
                                                              int bins_per_chunk,
                                                              c128 *Yout = nullptr) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.z;
  const int i_begin = blockIdx.y * bins_per_chunk;
  const int i_end = min(F, i_begin + bins_per_chunk);
  if (j >= T) return;
  double acc[N];
#pragma unroll
  for (int n = 0; n < N; ++n) acc[n] = 0.0;
  for (int i = i_begin; i < i_end; ++i) {
    c128 x[N];
#pragma unroll
    for (int m = 0; m < N; ++m) x[m] = X[(((long long)b * N + m) * F + i) * T + j];
    if (W) {
      const c128 *Wi = W + ((long long)b * F + i) * (N * N);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < N; ++m) cfma(y, Wi[n * N + m], x[m]);
        acc[n] += cabs2(y);
        if (Yout) Yout[(((long long)b * N + n) * F + i) * T + j] = y;
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] += cabs2(x[n]);
    }
  }
  double *dst = r2 + (long long)blockIdx.y * gridDim.z * N * T;
#pragma unroll
  for (int n = 0; n < N; ++n) dst[((long long)b * N + n) * T + j] = acc[n];
}

__global__ __launch_bounds__(256) void k_iva_weight(const double *__restrict__ r2, double *weight,
                                                    double *variance, long long total, int F,
                                                    int contrast, int floor_kind, double eps) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const double p = r2[e];
  const double r = sqrt(p);
  const double denom = apply_floor(2.0 * r, floor_kind, eps);
  double dG;
  if (contrast == SSSPY_CONTRAST_GAUSS) {
    const double alpha = p / (double)F;
    variance[e] = alpha;
    dG = 2.0 * r / alpha;
  } else if (contrast == SSSPY_CONTRAST_GAUSS_FIXED) {
    dG = 2.0 * r / variance[e];
  } else {
    dG = 2.0;
  }
  weight[e] = dG / denom;
}

// out[b] = sum_n (1/T) sum_j G(r_nj); one block per mixture
__global__ __launch_bounds__(256) void k_iva_loss(const double *__restrict__ r2,
                                                  const double *__restrict__ variance, double *out,
                                                  int N, int F, int T, int contrast) {
  __shared__ double scratch[4];
  const int b = blockIdx.x;
  double local = 0.0;
  for (int e = threadIdx.x; e < N * T; e += blockDim.x) {
    const double p = r2[(long long)b * N * T + e];
    if (contrast == SSSPY_CONTRAST_GAUSS) {
      const double alpha = variance[(long long)b * N * T + e];
      local += (double)F * log(alpha) + p / alpha;
    } else {
      local += 2.0 * sqrt(p);
    }
  }
  const double total = block_sum(local, scratch);
  if (threadIdx.x == 0) out[b] = total / (double)T;
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

// Shape of the frame-power pass: frames per block (256: a thread per frame, large batches; 64: four
// waves share the bins, a handful of mixtures) and bin chunks -- enough blocks to fill the chip, at
// most 128 slabs for the fold.
struct FramePowerPlan {
  int frames_per_block, bins_per_chunk, chunks;
};
static FramePowerPlan frame_power_plan(int B, int F, int T) {
  FramePowerPlan p;
  p.frames_per_block = (long long)((T + 255) / 256) * B >= 128 ? 256 : 64;
  const int gx = (T + p.frames_per_block - 1) / p.frames_per_block;
  long long want = (p.frames_per_block == 256 ? 2048 : 1024) / ((long long)gx * B);
  if (want < 1) want = 1;
  if (want > 128) want = 128;
  if (want > F) want = F;
  p.bins_per_chunk = (int)((F + want - 1) / want);
  p.chunks = (F + p.bins_per_chunk - 1) / p.bins_per_chunk;
  return p;
}

size_t ssspy_iva_frame_power_workspace_bytes(int B, int N, int F, int T) {
  if (B <= 0 || N <= 0 || F <= 0 || T <= 0) return 0;
  const FramePowerPlan p = frame_power_plan(B, F, T);
  if (p.chunks <= 1) return 0;
  const long long total = (long long)B * N * T;
  return (size_t)p.chunks * total * sizeof(double) + fold_scratch_bytes(total, p.chunks);
}

static int frame_power_impl(const void *X, const void *W, double *r2, void *Yout, int B, int N,
                            int F, int T, void *workspace, size_t workspace_bytes, void *stream);

int ssspy_iva_frame_power(const void *X, const void *W, double *r2, int B, int N, int F, int T,
                          void *workspace, size_t workspace_bytes, void *stream) {
  return frame_power_impl(X, W, r2, nullptr, B, N, F, T, workspace, workspace_bytes, stream);
}

int ssspy_separate_frame_power(const void *X, const void *W, void *Y, double *r2, int B, int N,
                               int F, int T, void *workspace, size_t workspace_bytes,
                               void *stream) {
  SSSPY_REQUIRE(W && Y, "separate_frame_power: bad argument");
  if (rt_sources_ok(N))
    return fail(SSSPY_ERR_UNSUPPORTED, "separate_frame_power: up to 8 sources (use ssspy_separate + "
                                       "ssspy_iva_frame_power)");
  return frame_power_impl(X, W, r2, Y, B, N, F, T, workspace, workspace_bytes, stream);
}

static int frame_power_impl(const void *X, const void *W, double *r2, void *Yout, int B, int N,
                            int F, int T, void *workspace, size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && r2 && B > 0 && F > 0 && T > 0, "iva_frame_power: bad argument");
  hipStream_t st = as_stream(stream);
  const FramePowerPlan p = frame_power_plan(B, F, T);
  const size_t need = ssspy_iva_frame_power_workspace_bytes(B, N, F, T);
  SSSPY_REQUIRE(need == 0 || (workspace && workspace_bytes >= need),
                "iva_frame_power: workspace too small");
  double *dst = p.chunks > 1 ? (double *)workspace : r2;
  dim3 grid((T + p.frames_per_block - 1) / p.frames_per_block, p.chunks, B), block(256);
  if (rt_sources_ok(N)) {  // a thread per frame, whatever the plan's block shape (same slab layout)
    int rc = rt_frame_power(X, W, dst, B, N, F, T, p.bins_per_chunk, p.chunks, st);
    if (rc || p.chunks == 1) return rc;
    const long long total = (long long)B * N * T;
    return launch_fold_slabs((const double *)workspace,
                             (char *)workspace + (size_t)p.chunks * total * sizeof(double), r2,
                             total, p.chunks, st);
  }
  if (p.frames_per_block == 256) {
    DISPATCH_N(N, hipLaunchKernelGGL((k_iva_frame_power_wide<NN>), grid, block, 0, st,
                                     (const c128 *)X, (const c128 *)W, dst, F, T,
                                     p.bins_per_chunk, (c128 *)Yout));
  } else {
    DISPATCH_N(N, hipLaunchKernelGGL((k_iva_frame_power<NN>), grid, block, 0, st, (const c128 *)X,
                                     (const c128 *)W, dst, F, T, p.bins_per_chunk,
                                     (c128 *)Yout));
  }
  int rc = check_launch("k_iva_frame_power");
  if (rc || p.chunks == 1) return rc;
  const long long total = (long long)B * N * T;
  return launch_fold_slabs((const double *)workspace,
                           (char *)workspace + (size_t)p.chunks * total * sizeof(double), r2, total,
                           p.chunks, st);
}

int ssspy_iva_weight(const double *r2, double *weight, double *variance, int B, int N, int F, int T,
                     int contrast, int floor_kind, double floor_eps, void *stream) {
  SSSPY_REQUIRE(r2 && weight && B > 0, "iva_weight: bad argument");
  SSSPY_REQUIRE(contrast == SSSPY_CONTRAST_LAPLACE ||
                    ((contrast == SSSPY_CONTRAST_GAUSS || contrast == SSSPY_CONTRAST_GAUSS_FIXED) &&
                     variance),
                "iva_weight: bad contrast / variance");
  const long long total = (long long)B * N * T;
  hipLaunchKernelGGL(k_iva_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), r2, weight, variance, total, F, contrast, floor_kind,
                     floor_eps);
  return check_launch("k_iva_weight");
}

int ssspy_iva_loss_data(const double *r2, const double *variance, double *out, int B, int N, int F,
                        int T, int contrast, void *stream) {
  SSSPY_REQUIRE(r2 && out && B > 0, "iva_loss_data: bad argument");
  SSSPY_REQUIRE(contrast == SSSPY_CONTRAST_LAPLACE || (contrast == SSSPY_CONTRAST_GAUSS && variance),
                "iva_loss_data: bad contrast / variance");
  hipLaunchKernelGGL(k_iva_loss, dim3(B), dim3(256), 0, as_stream(stream), r2, variance, out, N, F,
                     T, contrast);
  return check_launch("k_iva_loss");
}

}  // extern "C"
