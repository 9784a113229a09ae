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
// from sincospi.  A power-of-two n_fft up to 8192 (128 KB of complex128 in LDS) is transformed
// directly; any other n_fft up to 4096 (SciPy takes any nperseg) goes through Bluestein's chirp-z
// identity inside the same workgroup (below).  Longer transforms (round 6: a power of two up to
// 65536, any other length up to 32768) run the same code on a workgroup-private slice of the
// caller's workspace in HBM instead of LDS, STFT_GLOBAL_BLOCKS workgroups walking the segments --
// the passes then go through the L2; correct, not tuned: SciPy takes any nperseg, these sizes are
// rare.  Bluestein: x_t e^{-+ i pi t^2 / n} convolved with the chirp e^{+- i pi j^2 / n}
// by two radix-2 transforms of length m = the power of two >= 2 n - 1, the chirp's spectrum computed
// once per call into a caller-provided workspace (ssspy_stft_workspace_bytes).
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

// e^{sign * i pi t^2 / n}; t^2 is reduced mod 2 n in integers first (the phase has period 2 n)
__device__ __forceinline__ c128 chirp(long long t, int n, double sign) {
  const long long r = (t * t) % (2ll * n);
  double sn, cs;
  sincospi(sign * (double)r / (double)n, &sn, &cs);
  return cmake(cs, sn);
}

__device__ __forceinline__ void bitrev_permute(c128 *buf, int m, int log2m) {
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    const int j = (int)(__brev((unsigned)i) >> (32 - log2m));
    if (i < j) {
      const c128 a = buf[i], b = buf[j];
      buf[i] = b;
      buf[j] = a;
    }
  }
  __syncthreads();
}

// Spectrum of the Bluestein chirp b_j = e^{-sign * i pi j^2 / n}, j = -(n-1) .. n-1 at index j mod m
// (the DFT with exponent sign `sign` convolves with the chirp of the opposite sign).  One workgroup.
__global__ __launch_bounds__(256) void k_bluestein_chirp(c128 *__restrict__ bhat, int n, int m,
                                                         int log2m, double sign, c128 *gbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *buf = gbuf ? gbuf : reinterpret_cast<c128 *>(smem);
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    c128 v = cmake(0.0, 0.0);
    if (i < n) v = chirp(i, n, -sign);
    else if (i > m - n) v = chirp(m - i, n, -sign);
    buf[__brev((unsigned)i) >> (32 - log2m)] = v;
  }
  fft_inplace(buf, m, log2m, -1.0);
  for (int i = threadIdx.x; i < m; i += blockDim.x) bhat[i] = buf[i];
}

// DFT of length n (any n) of the sequence already in buf[0 .. n) in NATURAL order, exponent sign
// `sign`, result in buf[0 .. n) (unnormalised).  m > 0: Bluestein with the chirp spectrum `bhat`;
// m == 0: n is a power of two and buf is transformed directly.
__device__ __forceinline__ void dft_any(c128 *buf, int n, int log2n, int m, int log2m,
                                        const c128 *__restrict__ bhat, double sign) {
  if (m == 0) {
    bitrev_permute(buf, n, log2n);
    fft_inplace(buf, n, log2n, sign);
    return;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < m; t += blockDim.x)
    buf[t] = t < n ? cmul(buf[t], chirp(t, n, sign)) : cmake(0.0, 0.0);
  bitrev_permute(buf, m, log2m);
  fft_inplace(buf, m, log2m, -1.0);
  for (int k = threadIdx.x; k < m; k += blockDim.x) buf[k] = cmul(buf[k], bhat[k]);
  bitrev_permute(buf, m, log2m);
  fft_inplace(buf, m, log2m, 1.0);
  const double inv = 1.0 / (double)m;
  for (int k = threadIdx.x; k < n; k += blockDim.x)
    buf[k] = cscale(cmul(buf[k], chirp(k, n, sign)), inv);
  __syncthreads();
}

// grid: (n_frames, C, B) with the segment in LDS, or (blocks) walking the n_frames C B segments with
// the segment in the block's slice of `gscratch` (lds_len c128 each).
// x (B, C, L) real -> Z (B, C, n/2+1, n_frames) complex
__global__ __launch_bounds__(256) void k_stft(const double *__restrict__ x, c128 *__restrict__ Z,
                                              long long L, int n, int log2n, int hop, int n_frames,
                                              const double *__restrict__ window, double scale, int m,
                                              int log2m, const c128 *__restrict__ bhat, int C, int B,
                                              c128 *gscratch, int lds_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *buf = gscratch ? gscratch + (long long)blockIdx.x * lds_len : reinterpret_cast<c128 *>(smem);
  const long long total = (long long)n_frames * C * B;
  const long long first = gscratch ? blockIdx.x
                                   : ((long long)blockIdx.z * C + blockIdx.y) * n_frames + blockIdx.x;
  const long long step = gscratch ? gridDim.x : total;
  for (long long sgm = first; sgm < total; sgm += step) {
    const int frame = (int)(sgm % n_frames);
    const long long bc = sgm / n_frames;  // b * C + ch
    const double *xs = x + bc * L;
    const long long start = (long long)frame * hop - n / 2;
    __syncthreads();  // (the previous segment's reads of buf are done)
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
      const long long sidx = start + t;
      const double v = (sidx >= 0 && sidx < L) ? xs[sidx] * window[t] : 0.0;
      buf[t] = cmake(v, 0.0);
    }
    dft_any(buf, n, log2n, m, log2m, bhat, -1.0);
    const int F = n / 2 + 1;
    c128 *out = Z + bc * F * n_frames + frame;
    for (int k = threadIdx.x; k < F; k += blockDim.x)
      out[(long long)k * n_frames] = cmake(buf[k].x * scale, buf[k].y * scale);
  }
}

// grid as k_stft.  Z (B, C, n/2+1, n_frames) -> seg (B, C, n_frames, n) real, each segment
// = irfft(frame) * window * gain
__global__ __launch_bounds__(256) void k_istft_segments(const c128 *__restrict__ Z,
                                                        double *__restrict__ seg, int n, int log2n,
                                                        int n_frames,
                                                        const double *__restrict__ window,
                                                        double gain, int m, int log2m,
                                                        const c128 *__restrict__ bhat, int C, int B,
                                                        c128 *gscratch, int lds_len) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *buf = gscratch ? gscratch + (long long)blockIdx.x * lds_len : reinterpret_cast<c128 *>(smem);
  const int F = n / 2 + 1;
  const long long total = (long long)n_frames * C * B;
  const long long first = gscratch ? blockIdx.x
                                   : ((long long)blockIdx.z * C + blockIdx.y) * n_frames + blockIdx.x;
  const long long step = gscratch ? gridDim.x : total;
  for (long long sgm = first; sgm < total; sgm += step) {
    const int frame = (int)(sgm % n_frames);
    const long long bc = sgm / n_frames;
    const c128 *in = Z + bc * F * n_frames + frame;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
      c128 v;
      if (k < F) {
        v = in[(long long)k * n_frames];
        // irfft ignores the imaginary part of DC and (even n) of Nyquist
        if (k == 0 || (2 * k == n)) v.y = 0.0;
      } else {
        v = cconj(in[(long long)(n - k) * n_frames]);
      }
      buf[k] = v;
    }
    dft_any(buf, n, log2n, m, log2m, bhat, 1.0);
    double *out = seg + (bc * n_frames + frame) * n;
    const double g = gain / (double)n;
    for (int t = threadIdx.x; t < n; t += blockDim.x) out[t] = buf[t].x * g * window[t];
  }
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

// the segment sits in LDS: 16 bytes x the transform length, 128 KB of the CU's 160 at 8192 (above
// the 64 KB a launch gets by default: raise the kernel's limit first)
constexpr int STFT_MAX_LEN = 8192;
static int allow_lds(const void *kernel, int len) {
  const size_t bytes = (size_t)len * sizeof(c128);
  if (bytes <= 64 * 1024) return SSSPY_OK;
  hipError_t e =
      hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? SSSPY_OK : fail(SSSPY_ERR_HIP, hipGetErrorString(e));
}

// transform plan of a segment length: direct radix-2 (m = 0) or Bluestein of length m; in LDS up to
// STFT_MAX_LEN points, in a workgroup-private slice of the workspace up to STFT_GLOBAL_MAX
constexpr int STFT_GLOBAL_MAX = 65536, STFT_GLOBAL_BLOCKS = 256;
struct FftPlan {
  int log2n, m, log2m, lds_len;
  bool ok, global;
};
static FftPlan fft_plan(int n) {
  FftPlan p = {ilog2_exact(n), 0, 0, n, false, false};
  if (n < 2) return p;
  if (p.log2n < 1) {
    int m = 1, lg = 0;
    while (m < 2 * n - 1) {
      m <<= 1;
      ++lg;
    }
    p.log2n = 0;
    p.m = m;
    p.log2m = lg;
    p.lds_len = m;
  }
  p.global = p.lds_len > STFT_MAX_LEN;
  p.ok = p.lds_len <= STFT_GLOBAL_MAX;
  return p;
}
// workspace: the chirp's spectrum (m), then the workgroups' slices (STFT_GLOBAL_BLOCKS x lds_len)
static size_t plan_chirp_bytes(const FftPlan &p) { return (size_t)p.m * sizeof(c128); }
static size_t plan_scratch_bytes(const FftPlan &p) {
  return p.global ? (size_t)STFT_GLOBAL_BLOCKS * p.lds_len * sizeof(c128) : 0;
}

size_t ssspy_stft_workspace_bytes(int n_fft) {
  const FftPlan p = fft_plan(n_fft);
  return p.ok ? plan_chirp_bytes(p) + plan_scratch_bytes(p) : 0;
}

static int prepare_chirp(const FftPlan &p, int n, double sign, void *workspace, hipStream_t st) {
  if (!p.m && !p.global) return SSSPY_OK;
  SSSPY_REQUIRE(workspace, "stft: this n_fft needs ssspy_stft_workspace_bytes() of workspace");
  if (!p.m) return SSSPY_OK;
  c128 *gbuf = nullptr;  // (the first workgroup slice serves the one-block chirp kernel)
  if (p.global) gbuf = (c128 *)((char *)workspace + plan_chirp_bytes(p));
  int rc = p.global ? SSSPY_OK : allow_lds((const void *)k_bluestein_chirp, p.m);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bluestein_chirp, dim3(1), dim3(256),
                     p.global ? 0 : (size_t)p.m * sizeof(c128), st, (c128 *)workspace, n, p.m,
                     p.log2m, sign, gbuf);
  return check_launch("k_bluestein_chirp");
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
               long long n_samples, int n_fft, int hop, void *workspace, void *stream) {
  SSSPY_REQUIRE(x && Z && window && B > 0 && C > 0 && n_samples > 0 && hop > 0 && hop <= n_fft &&
                    window_sum != 0.0,
                "stft: bad argument");
  const FftPlan p = fft_plan(n_fft);
  if (!p.ok)
    return fail(SSSPY_ERR_UNSUPPORTED,
                "stft: n_fft must be a power of two <= 65536 or any length in [2, 32768]");
  hipStream_t st = as_stream(stream);
  int rc0 = prepare_chirp(p, n_fft, -1.0, workspace, st);
  if (rc0) return rc0;
  if (!p.global) {
    rc0 = allow_lds((const void *)k_stft, p.lds_len);
    if (rc0) return rc0;
  }
  const int n_frames = ssspy_stft_frames(n_samples, n_fft, hop);
  const long long total = (long long)n_frames * C * B;
  c128 *gscratch = p.global ? (c128 *)((char *)workspace + plan_chirp_bytes(p)) : nullptr;
  const dim3 grid = p.global ? dim3((unsigned)(total < STFT_GLOBAL_BLOCKS ? total : STFT_GLOBAL_BLOCKS))
                             : dim3(n_frames, C, B);
  hipLaunchKernelGGL(k_stft, grid, dim3(256), p.global ? 0 : (size_t)p.lds_len * sizeof(c128), st, x,
                     (c128 *)Z, n_samples, n_fft, p.log2n, hop, n_frames, window, 1.0 / window_sum,
                     p.m, p.log2m, (const c128 *)workspace, C, B, gscratch, p.lds_len);
  return check_launch("k_stft");
}

long long ssspy_istft_samples(int n_frames, int n_fft, int hop) {
  if (n_frames <= 0 || n_fft <= 0 || hop <= 0) return 0;
  return (long long)n_fft + (long long)(n_frames - 1) * hop - 2LL * (n_fft / 2);
}

int ssspy_istft(const void *Z, double *x, const double *window, double window_sum,
                double *segments, int B, int C, int n_frames, int n_fft, int hop, void *workspace,
                void *stream) {
  SSSPY_REQUIRE(Z && x && window && segments && B > 0 && C > 0 && n_frames > 0 && hop > 0 &&
                    hop <= n_fft,
                "istft: bad argument");
  const FftPlan p = fft_plan(n_fft);
  if (!p.ok)
    return fail(SSSPY_ERR_UNSUPPORTED,
                "istft: n_fft must be a power of two <= 65536 or any length in [2, 32768]");
  hipStream_t st = as_stream(stream);
  int rc0 = prepare_chirp(p, n_fft, 1.0, workspace, st);
  if (rc0) return rc0;
  if (!p.global) {
    rc0 = allow_lds((const void *)k_istft_segments, p.lds_len);
    if (rc0) return rc0;
  }
  const long long total = (long long)n_frames * C * B;
  c128 *gscratch = p.global ? (c128 *)((char *)workspace + plan_chirp_bytes(p)) : nullptr;
  const dim3 grid = p.global ? dim3((unsigned)(total < STFT_GLOBAL_BLOCKS ? total : STFT_GLOBAL_BLOCKS))
                             : dim3(n_frames, C, B);
  hipLaunchKernelGGL(k_istft_segments, grid, dim3(256),
                     p.global ? 0 : (size_t)p.lds_len * sizeof(c128), st, (const c128 *)Z, segments,
                     n_fft, p.log2n, n_frames, window, window_sum, p.m, p.log2m,
                     (const c128 *)workspace, C, B, gscratch, p.lds_len);
  int rc = check_launch("k_istft_segments");
  if (rc) return rc;
  const long long L_out = ssspy_istft_samples(n_frames, n_fft, hop);
  hipLaunchKernelGGL(k_istft_overlap_add, dim3((unsigned)((L_out + 255) / 256), C, B), dim3(256), 0,
                     st, (const double *)segments, x, L_out, n_fft, hop, n_frames, window);
  return check_launch("k_istft_overlap_add");
}

}  // extern "C"
