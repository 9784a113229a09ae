// Fused iterative source steering (ISS1): one read and one write of Y per iteration.
//
// The reference (ssspy/bss/_update_spatial_model.py:146-194) sweeps the whole (N, F, T) tensor N
// times, each sweep two full-size temporaries and a rewrite of Y.  Bins are independent, and one
// bin's slab Y_i (N x T complex128: 128 KB at N=8, T=1024) fits the register file of a 256-thread
// workgroup: thread t keeps its FPT frames of all N rows in VGPRs, so the N sequential rank-1 sweeps
// run on chip; per sweep only 3N reals (num_{n'} complex, den_{n'} real) cross lanes:
//   row-level  : 4 DPP butterfly steps inside each 16-lane row (full-rate VALU, no LDS traffic)
//   block-level: 16 row partials through LDS, one barrier; lane n' < N adds up the partials of its
//                source and forms the steering coefficient v_{n'} (one divide / square root per
//                sweep instead of N in every thread: 7.4 -> 6.4 ms at configs[2] x 32 mixtures),
//                v_readlane then hands the N coefficients to every thread as wave-uniform scalars.
// Tried and dropped (profiles/r01_other_configs.txt): 512-thread workgroups that prefetch the next
// bin's slab with LDS-direct loads (global_load_lds_dwordx4, 128 KB of LDS) while sweeping the
// current one -- the load latency hides, but the cross-lane reduction (12 DPP/add instructions per
// value and wave, independent of the frames per thread) doubles per SIMD and the kernel got slower
// (8.1 ms).
// While the updated slab is written back, |y|^2 is accumulated per (source, frame) over the bins of
// the block and added atomically to r2_next: the frame powers r_nj^2 of the NEXT iteration's
// auxiliary weights, which would otherwise need their own pass over Y (2 passes).
#include "common.hpp"

namespace ssspy {

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ double row_allsum(double v) {
#define SSSPY_DPP_STEP(ctrl)                                                              \
  {                                                                                       \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true); \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true); \
    v += __hiloint2double(hi, lo);                                                        \
  }
  SSSPY_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]  (lane ^ 1)
  SSSPY_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]  (lane ^ 2)
  SSSPY_DPP_STEP(0x141)  // row_half_mirror      (other quad of the 8-lane half)
  SSSPY_DPP_STEP(0x140)  // row_mirror           (other half of the row)
#undef SSSPY_DPP_STEP
  return v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// The N sequential rank-1 sweeps of one bin on the register-resident slab y[n][f] (thread-owned
// frames), weights phi[n][f]; `part` is the two-buffer LDS scratch of the block reduction (one barrier per sweep), `parity` its state.
// ref: ssspy/bss/_update_spatial_model.py:146-194.
template <int N, int FPT, int NW>
__device__ __forceinline__ void iss_sweeps(c128 (&y)[N][FPT], const double (&phi)[N][FPT],
                                           double (*part)[NW * 4 * 3 * N], int &parity, double invT,
                                           int floor_kind, double eps) {
#pragma unroll
  for (int n = 0; n < N; ++n) {
    // per source s: sum_j phi_s y_s conj(y_n) (complex) and sum_j phi_s |y_n|^2, SGR sources at a
    // time (register budget).  Block totals: 16-lane rows by DPP, row partials through LDS; lane
    // s < N then owns source s: it adds up the partials of its three sums and forms the steering
    // coefficient v_s, which v_readlane hands to every thread as a wave-uniform scalar (SGPR
    // operands of the update)
    constexpr int SGR = N > 4 ? 4 : N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *pp = part[parity];
    parity ^= 1;
#pragma unroll
    for (int s0 = 0; s0 < N; s0 += SGR) {
      double red[3 * SGR];
#pragma unroll
      for (int k = 0; k < 3 * SGR; ++k) red[k] = 0.0;
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const c128 yn = y[n][f];
        const double pn = cabs2(yn);
#pragma unroll
        for (int ss = 0; ss < SGR; ++ss) {
          const int s = s0 + ss < N ? s0 + ss : N - 1;
          const double w = phi[s][f];
          const c128 z = cmulc(y[s][f], yn);
          red[3 * ss] = fma(w, z.x, red[3 * ss]);
          red[3 * ss + 1] = fma(w, z.y, red[3 * ss + 1]);
          red[3 * ss + 2] = fma(w, pn, red[3 * ss + 2]);
        }
      }
#pragma unroll
      for (int k = 0; k < 3 * SGR; ++k) red[k] = row_allsum(red[k]);
      if ((lane & 15) == 0) {
        double *dst = pp + (wave * 4 + (lane >> 4)) * (3 * N) + 3 * s0;
#pragma unroll
        for (int k = 0; k < 3 * SGR; ++k)
          if (3 * s0 + k < 3 * N) dst[k] = red[k];
      }
    }
    __syncthreads();
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (lane < N) {
#pragma unroll
      for (int p = 0; p < NW * 4; ++p) {
        t0 += pp[p * (3 * N) + 3 * lane];
        t1 += pp[p * (3 * N) + 3 * lane + 1];
        t2 += pp[p * (3 * N) + 3 * lane + 2];
      }
    }
    const double den = apply_floor(t2 * invT, floor_kind, eps);
    const double vx = lane == n ? 1.0 - 1.0 / sqrt(den) : t0 * invT / den;
    const double vy = lane == n ? 0.0 : t1 * invT / den;
    c128 v[N];
#pragma unroll
    for (int s = 0; s < N; ++s) v[s] = cmake(readlane_f64(vx, s), readlane_f64(vy, s));
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      const c128 yn = y[n][f];
#pragma unroll
      for (int s = 0; s < N; ++s) cfms(y[s][f], v[s], yn);
    }
  }
}

// grid: (ceil(F / bins_per_block), B); 256 threads; thread t owns frames t + 256 f, f < FPT.
// weight: (B, N, T) when !PER_BIN, (B, N, F, T) when PER_BIN.
template <int N, int FPT, bool PER_BIN>
__global__ __launch_bounds__(256) void k_iss1_fused(c128 *Y, const double *__restrict__ weight,
                                                    double *r2_next, int F, int T,
                                                    int bins_per_block, int floor_kind, double eps) {
  __shared__ double part[2][4 * 4 * 3 * N];
  const int b = blockIdx.y;
  const int i_begin = blockIdx.x * bins_per_block;
  const int i_end = min(F, i_begin + bins_per_block);
  const double invT = 1.0 / (double)T;
  bool fv[FPT];
  int jj[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) {
    const int j = threadIdx.x + 256 * f;
    fv[f] = j < T;
    jj[f] = fv[f] ? j : T - 1;
  }
  double phi[N][FPT], r2acc[N][FPT];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      r2acc[n][f] = 0.0;
      phi[n][f] = (!PER_BIN && fv[f]) ? weight[((long long)b * N + n) * T + jj[f]] : 0.0;
    }
  int parity = 0;
  for (int i = i_begin; i < i_end; ++i) {
    c128 y[N][FPT];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const c128 v = Y[(((long long)b * N + n) * F + i) * T + jj[f]];
        y[n][f] = fv[f] ? v : cmake(0.0, 0.0);
        if (PER_BIN)
          phi[n][f] = fv[f] ? weight[(((long long)b * N + n) * F + i) * T + jj[f]] : 0.0;
      }
    iss_sweeps<N, FPT, 4>(y, phi, part, parity, invT, floor_kind, eps);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        if (fv[f]) Y[(((long long)b * N + n) * F + i) * T + jj[f]] = y[n][f];
        r2acc[n][f] += cabs2(y[n][f]);
      }
  }
  if (r2_next) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int f = 0; f < FPT; ++f)
        if (fv[f]) atomicAdd(r2_next + ((long long)b * N + n) * T + jj[f], r2acc[n][f]);
  }
}

template <int N>
constexpr int iss_max_fpt() {
  return N <= 2 ? 8 : (N <= 4 ? 8 : 4);
}

template <int N, int FPT>
static int launch_iss(void *Y, const double *weight, bool per_bin, double *r2_next, int B, int F,
                      int T, int floor_kind, double eps, hipStream_t st) {
  // a few bins per block amortise the weight loads and the r2 atomics; keep >= ~2 blocks per CU
  long long want_blocks = 1024;
  int bpb = (int)(((long long)B * F + want_blocks - 1) / want_blocks);
  if (bpb < 1) bpb = 1;
  if (bpb > 8) bpb = 8;
  dim3 grid((F + bpb - 1) / bpb, B), block(256);
  if (per_bin)
    hipLaunchKernelGGL((k_iss1_fused<N, FPT, true>), grid, block, 0, st, (c128 *)Y, weight, r2_next,
                       F, T, bpb, floor_kind, eps);
  else
    hipLaunchKernelGGL((k_iss1_fused<N, FPT, false>), grid, block, 0, st, (c128 *)Y, weight, r2_next,
                       F, T, bpb, floor_kind, eps);
  return check_launch("k_iss1_fused");
}

template <int N>
static int dispatch_iss(void *Y, const double *weight, bool per_bin, double *r2_next, int B, int F,
                        int T, int floor_kind, double eps, hipStream_t st) {
  const int fpt = (T + 255) / 256;
  if (fpt <= 1) return launch_iss<N, 1>(Y, weight, per_bin, r2_next, B, F, T, floor_kind, eps, st);
  if (fpt <= 2) return launch_iss<N, 2>(Y, weight, per_bin, r2_next, B, F, T, floor_kind, eps, st);
  if (fpt <= 4) return launch_iss<N, 4>(Y, weight, per_bin, r2_next, B, F, T, floor_kind, eps, st);
  if (iss_max_fpt<N>() >= 8 && fpt <= 8)
    return launch_iss<N, (iss_max_fpt<N>() >= 8 ? 8 : 4)>(Y, weight, per_bin, r2_next, B, F, T,
                                                          floor_kind, eps, st);
  return fail(SSSPY_ERR_UNSUPPORTED, "iss1_fused: n_frames too large for the register-resident slab");
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

int ssspy_iss1_fused_max_frames(int N) {
  if (N < 1 || N > SSSPY_MAX_SOURCES) return 0;
  return 256 * (N <= 4 ? 8 : 4);
}

int ssspy_iss1_fused(void *Y, const double *weight, int weight_kind, double *r2_next, int B, int N,
                     int F, int T, int floor_kind, double floor_eps, void *stream) {
  SSSPY_REQUIRE(Y && weight && B > 0 && F > 0 && T > 0, "iss1_fused: bad argument");
  SSSPY_REQUIRE(weight_kind == SSSPY_WEIGHT_FRAME || weight_kind == SSSPY_WEIGHT_BIN_FRAME,
                "iss1_fused: weight_kind must be FRAME or BIN_FRAME");
  SSSPY_REQUIRE(T <= ssspy_iss1_fused_max_frames(N), "iss1_fused: n_frames above the fused limit");
  const bool per_bin = weight_kind == SSSPY_WEIGHT_BIN_FRAME;
  if (r2_next) {
    hipError_t e = hipMemsetAsync(r2_next, 0, (size_t)B * N * T * sizeof(double), as_stream(stream));
    if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  }
  DISPATCH_N(N, return dispatch_iss<NN>(Y, weight, per_bin, r2_next, B, F, T, floor_kind, floor_eps,
                                        as_stream(stream)));
  return SSSPY_OK;
}

}  // extern "C"
