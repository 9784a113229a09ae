// AuxIVA (IP1 / ISS1) kernels: frame power r_nj^2 = sum_i |y_nij|^2, auxiliary weights and the
// contrast part of the loss.  The covariance / IP1 / ISS1 steps are the shared operators of
// spatial_kernels.hip.
#include "common.hpp"

namespace ssspy {

// lanes along frames (coalesced rows), block walks a chunk of bins; W_i is wave-uniform.
// grid: (ceil(T/256), bin chunks, B).  One chunk: r2 is stored directly; several: every chunk stores
// its sums to part[chunk][b][n][j] and k_iva_frame_power_fold adds them in chunk order -- no fp64
// atomics: the weights of the next iteration, hence the whole trajectory, are the same on every run.
template <int N>
__global__ __launch_bounds__(256) void k_iva_frame_power(const c128 *__restrict__ X,
                                                         const c128 *__restrict__ W, double *r2,
                                                         int F, int T, int bins_per_chunk) {
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
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] += cabs2(x[n]);
    }
  }
  // (r2 is the output itself with one chunk, else the chunk's slab of the partial buffer)
  double *dst = r2 + (long long)blockIdx.y * gridDim.z * N * T;
#pragma unroll
  for (int n = 0; n < N; ++n) dst[((long long)b * N + n) * T + j] = acc[n];
}

// r2[e] = sum over chunks, in chunk order; e over B * N * T
__global__ __launch_bounds__(256) void k_iva_frame_power_fold(const double *__restrict__ part,
                                                              double *r2, long long total,
                                                              int chunks) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  double s = 0.0;
  for (int ch0 = 0; ch0 < chunks; ch0 += 8) {  // eight loads in flight per round trip
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long long)min(ch0 + u, chunks - 1) * total + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += ch0 + u < chunks ? v[u] : 0.0;
  }
  r2[e] = s;
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

// bin chunks of the frame-power pass: enough blocks to fill the chip
static int frame_power_chunks(int B, int F, int T, int *bins_per_chunk) {
  const int gx = (T + 255) / 256;
  long long want = 2048 / ((long long)gx * B);
  if (want < 1) want = 1;
  if (want > F) want = F;
  *bins_per_chunk = (int)((F + want - 1) / want);
  return (F + *bins_per_chunk - 1) / *bins_per_chunk;
}

size_t ssspy_iva_frame_power_workspace_bytes(int B, int N, int F, int T) {
  if (B <= 0 || N <= 0 || F <= 0 || T <= 0) return 0;
  int bpc;
  const int chunks = frame_power_chunks(B, F, T, &bpc);
  return chunks > 1 ? (size_t)chunks * B * N * T * sizeof(double) : 0;
}

int ssspy_iva_frame_power(const void *X, const void *W, double *r2, int B, int N, int F, int T,
                          void *workspace, size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && r2 && B > 0 && F > 0 && T > 0, "iva_frame_power: bad argument");
  hipStream_t st = as_stream(stream);
  int bins_per_chunk;
  const int chunks = frame_power_chunks(B, F, T, &bins_per_chunk);
  const size_t need = ssspy_iva_frame_power_workspace_bytes(B, N, F, T);
  SSSPY_REQUIRE(need == 0 || (workspace && workspace_bytes >= need),
                "iva_frame_power: workspace too small");
  double *dst = chunks > 1 ? (double *)workspace : r2;
  dim3 grid((T + 255) / 256, chunks, B), block(256);
  DISPATCH_N(N, hipLaunchKernelGGL((k_iva_frame_power<NN>), grid, block, 0, st, (const c128 *)X,
                                   (const c128 *)W, dst, F, T, bins_per_chunk));
  int rc = check_launch("k_iva_frame_power");
  if (rc || chunks == 1) return rc;
  const long long total = (long long)B * N * T;
  hipLaunchKernelGGL(k_iva_frame_power_fold, dim3((unsigned)((total + 255) / 256)), block, 0, st,
                     (const double *)workspace, r2, total, chunks);
  return check_launch("k_iva_frame_power_fold");
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
