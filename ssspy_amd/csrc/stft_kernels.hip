// STFT / ISTFT either side of the separators: the reference's
// workflow calls scipy.signal.stft(x, window="hann", nperseg=n_fft, noverlap=n_fft - hop) before and
// scipy.signal.istft after the separator (tests/package/bss/test_*.py, notebooks); with these two
// the waveform can go in and come out of HBM without the spectrogram crossing PCIe.
//
// Conventions reproduced (scipy.signal defaults): boundary="zeros" (n_fft/2 zeros either side),
// padded=True (zeros at the end up to a whole number of hops), one-sided output, scaling="spectrum"
// (division by sum(window)), ISTFT = windowed overlap-add divided by the summed squared window
// (where it exceeds 1e-10), boundary removed.
//
// One workgroup transforms one segment in LDS: bit-reversed load, log2(n) radix-2 passes, twiddles
// from sincospi.  n_fft is a power of two up to 8192 (128 KB of complex128 in LDS).
#include "common.hpp"
#include "ssspy_amd.h"

namespace ssspy {

__device__ __forceinline__ void fft_inplace(c128 *buf, int n, int log2n, double sign) {
  for (int s = 1; s <= log2n; ++s) {
    const int half = 1 << (s - 1);
    __syncthreads();
    for (int e = threadIdx.x; e < n / 2; e += blockDim.x) {
      const int grp = e >> (s - 1), pos = e & (half - 1);
      const int i0 = (grp << s) + pos, i1 = i0 + half;
      double sn, cs;
      sincospi(sign * (double)pos / (double)half, &sn, &cs);
      const c128 a = buf[i0], b = buf[i1];
      const c128 t = cmake(b.x * cs - b.y * sn, b.x * sn + b.y * cs);
      buf[i0] = cmake(a.x + t.x, a.y + t.y);
      buf[i1] = cmake(a.x - t.x, a.y - t.y);
    }
  }
  __syncthreads();
}

// grid: (n_frames, C, B).  x (B, C, L) real -> Z (B, C, n/2+1, n_frames) complex
__global__ __launch_bounds__(256) void k_stft(const double *__restrict__ x, c128 *__restrict__ Z,
                                              long long L, int n, int log2n, int hop, int n_frames,
                                              const double *__restrict__ window, double scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *buf = reinterpret_cast<c128 *>(smem);
  const int frame = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int C = gridDim.y;
  const double *xs = x + ((long long)b * C + ch) * L;
  const long long start = (long long)frame * hop - n / 2;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const long long sidx = start + t;
    const double v = (sidx >= 0 && sidx < L) ? xs[sidx] * window[t] : 0.0;
    buf[__brev((unsigned)t) >> (32 - log2n)] = cmake(v, 0.0);
  }
  fft_inplace(buf, n, log2n, -1.0);
  const int F = n / 2 + 1;
  c128 *out = Z + ((long long)b * C + ch) * F * n_frames + frame;
  for (int k = threadIdx.x; k < F; k += blockDim.x)
    out[(long long)k * n_frames] = cmake(buf[k].x * scale, buf[k].y * scale);
}

// grid: (n_frames, C, B).  Z (B, C, n/2+1, n_frames) -> seg (B, C, n_frames, n) real, each segment
// = irfft(frame) * window * gain
__global__ __launch_bounds__(256) void k_istft_segments(const c128 *__restrict__ Z,
                                                        double *__restrict__ seg, int n, int log2n,
                                                        int n_frames,
                                                        const double *__restrict__ window,
                                                        double gain) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *buf = reinterpret_cast<c128 *>(smem);
  const int frame = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const int C = gridDim.y, F = n / 2 + 1;
  const c128 *in = Z + ((long long)b * C + ch) * F * n_frames + frame;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    c128 v;
    if (k < F) {
      v = in[(long long)k * n_frames];
      if (k == 0 || k == n / 2) v.y = 0.0;  // irfft ignores the imaginary part of DC / Nyquist
    } else {
      v = cconj(in[(long long)(n - k) * n_frames]);
    }
    buf[__brev((unsigned)k) >> (32 - log2n)] = v;
  }
  fft_inplace(buf, n, log2n, 1.0);
  double *out = seg + (((long long)b * C + ch) * n_frames + frame) * n;
  const double g = gain / (double)n;
  for (int t = threadIdx.x; t < n; t += blockDim.x) out[t] = buf[t].x * g * window[t];
}

// x[b, c, s] = sum_k seg[k][s + n/2 - k hop] / sum_k window^2[...] (where > 1e-10)
__global__ __launch_bounds__(256) void k_istft_overlap_add(const double *__restrict__ seg,
                                                           double *__restrict__ x, long long L_out,
                                                           int n, int hop, int n_frames,
                                                           const double *__restrict__ window) {
  const int ch = blockIdx.y, b = blockIdx.z, C = gridDim.y;
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= L_out) return;
  const long long p = s + n / 2;  // position in the padded signal
  long long k_hi = p / hop;
  if (k_hi > n_frames - 1) k_hi = n_frames - 1;
  long long k_lo = (p - n + hop) / hop;  // smallest k with k hop + n > p
  if (p - n + 1 <= 0) k_lo = 0;
  if (k_lo < 0) k_lo = 0;
  const double *sg = seg + ((long long)b * C + ch) * n_frames * n;
  double acc = 0.0, norm = 0.0;
  for (long long k = k_lo; k <= k_hi; ++k) {
    const long long t = p - k * hop;
    if (t < 0 || t >= n) continue;
    acc += sg[k * n + t];
    norm = fma(window[t], window[t], norm);
  }
  x[((long long)b * C + ch) * L_out + s] = acc / (norm > 1e-10 ? norm : 1.0);
}

static int ilog2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return (1 << l) == n ? l : -1;
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

// the segment sits in LDS: 16 bytes x n_fft, 128 KB of the CU's 160 at 8192 (above the 64 KB a
// launch gets by default: raise the kernel's limit first)
constexpr int STFT_MAX_NFFT = 8192;
static int allow_lds(const void *kernel, int n_fft) {
  const size_t bytes = (size_t)n_fft * sizeof(c128);
  if (bytes <= 64 * 1024) return SSSPY_OK;
  hipError_t e =
      hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? SSSPY_OK : fail(SSSPY_ERR_HIP, hipGetErrorString(e));
}

int ssspy_stft_frames(long long n_samples, int n_fft, int hop) {
  if (n_samples <= 0 || n_fft <= 0 || hop <= 0) return 0;
  long long Lp = n_samples + 2LL * (n_fft / 2);
  const long long rem = (Lp - n_fft) % hop;
  const long long nadd = ((hop - rem) % hop) % n_fft;
  Lp += nadd;
  return (int)((Lp - n_fft) / hop + 1);
}

int ssspy_stft(const double *x, void *Z, const double *window, double window_sum, int B, int C,
               long long n_samples, int n_fft, int hop, void *stream) {
  SSSPY_REQUIRE(x && Z && window && B > 0 && C > 0 && n_samples > 0 && hop > 0 && hop <= n_fft &&
                    window_sum != 0.0,
                "stft: bad argument");
  const int lg = ilog2_exact(n_fft);
  if (lg < 1 || n_fft > STFT_MAX_NFFT)
    return fail(SSSPY_ERR_UNSUPPORTED, "stft: n_fft must be a power of two in [2, 8192]");
  int rc0 = allow_lds((const void *)k_stft, n_fft);
  if (rc0) return rc0;
  const int n_frames = ssspy_stft_frames(n_samples, n_fft, hop);
  hipLaunchKernelGGL(k_stft, dim3(n_frames, C, B), dim3(256), (size_t)n_fft * sizeof(c128),
                     as_stream(stream), x, (c128 *)Z, n_samples, n_fft, lg, hop, n_frames, window,
                     1.0 / window_sum);
  return check_launch("k_stft");
}

long long ssspy_istft_samples(int n_frames, int n_fft, int hop) {
  if (n_frames <= 0 || n_fft <= 0 || hop <= 0) return 0;
  return (long long)n_fft + (long long)(n_frames - 1) * hop - 2LL * (n_fft / 2);
}

int ssspy_istft(const void *Z, double *x, const double *window, double window_sum,
                double *segments, int B, int C, int n_frames, int n_fft, int hop, void *stream) {
  SSSPY_REQUIRE(Z && x && window && segments && B > 0 && C > 0 && n_frames > 0 && hop > 0 &&
                    hop <= n_fft,
                "istft: bad argument");
  const int lg = ilog2_exact(n_fft);
  if (lg < 1 || n_fft > STFT_MAX_NFFT)
    return fail(SSSPY_ERR_UNSUPPORTED, "istft: n_fft must be a power of two in [2, 8192]");
  int rc0 = allow_lds((const void *)k_istft_segments, n_fft);
  if (rc0) return rc0;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_istft_segments, dim3(n_frames, C, B), dim3(256),
                     (size_t)n_fft * sizeof(c128), st, (const c128 *)Z, segments, n_fft, lg,
                     n_frames, window, window_sum);
  int rc = check_launch("k_istft_segments");
  if (rc) return rc;
  const long long L_out = ssspy_istft_samples(n_frames, n_fft, hop);
  hipLaunchKernelGGL(k_istft_overlap_add, dim3((unsigned)((L_out + 255) / 256), C, B), dim3(256), 0,
                     st, (const double *)segments, x, L_out, n_fft, hop, n_frames, window);
  return check_launch("k_istft_overlap_add");
}

}  // extern "C"
