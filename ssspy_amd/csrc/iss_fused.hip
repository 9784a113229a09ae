// Fused iterative source steering (ISS1): one read and one write of Y per iteration.
//
// The reference (ssspy/bss/_update_spatial_model.py:146-194) sweeps the whole (N, F, T) tensor N
// times, each sweep two full-size temporaries and a rewrite of Y.  Bins are independent, and one
// bin's slab Y_i (N x T complex128: 128 KB at N=8, T=1024) fits the register file of a 256-thread
// workgroup: thread t keeps its FPT frames of all N rows in VGPRs, so the N sequential rank-1 sweeps
// run on chip; per sweep only 3N reals (num_{n'} complex, den_{n'} real) cross lanes.
//
// Round-2 structure (configs[2] x 32 mixtures: 6.46 -> see profiles/): the kernel is sized for TWO
// workgroups per CU (<= 256 VGPRs, <= 80 KB LDS) so that one workgroup's slab load / store overlaps
// the other's sweeps -- at one workgroup per CU load, 8 sweeps and store of a bin ran back to back:
//   * registers hold the slab (128 VGPRs at N*FPT = 32) and the frame weights (64); the frame-power
//     accumulators of the next iteration's weights moved to LDS (thread-private slots, one
//     read-modify-write per slab instead of one per sweep for the weights);
//   * the cross-lane reduction is a reduce-scatter on gfx950's row swaps: v_permlane32_swap and
//     v_permlane16_swap fold the two 32-lane halves and the row pairs at 1.5 instructions per value
//     while halving the number of live values each time (12 -> 6 -> 3 per source group), then 4 DPP
//     butterfly steps finish the remaining 3 values inside the 16-lane rows: 63 instead of 144
//     instructions per group of 4 sources, and 4 partials per value in LDS instead of 16;
//   * lane n' < N adds up the 4 wave partials of its source and forms the steering coefficient
//     v_{n'} (one divide / square root per sweep instead of N in every thread); v_readlane hands the
//     N coefficients to every thread as wave-uniform scalars.
// Tried and dropped in round 1 (profiles/r01_other_configs.txt): 512-thread workgroups that prefetch
// the next bin's slab with LDS-direct loads (128 KB of LDS) -- the cross-lane work doubled per SIMD.
// While the updated slab is written back, |y|^2 is accumulated per (source, frame) over the bins of
// the block and summed over the blocks (tree_fold) into r2_next: the frame powers r_nj^2 of the NEXT iteration's
// auxiliary weights, which would otherwise need their own pass over Y (2 passes).
#include <cstdlib>

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

// v_permlane32_swap: rows 2-3 of the first operand trade places with rows 0-1 of the second, so the
// sum of the two results is x folded over the 32-lane halves in lanes 0-31 (x[l] + x[l+32]) and y
// folded in lanes 32-63.  Two values in, one live value out.
__device__ __forceinline__ double fold_halves(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// v_permlane16_swap: odd rows of the first operand trade places with even rows of the second: the
// sum is x folded over each row pair in the even rows, y folded in the odd rows.
__device__ __forceinline__ double fold_row_pairs(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// Wave totals of NV (multiple of 4) per-lane values by reduce-scatter: afterwards every lane of row
// r holds, in v[u] (u < NV / 4), the wave total of the input value 4 u + iss_row_slot(r).
__device__ __forceinline__ int iss_row_slot(int row) { return ((row & 1) << 1) | (row >> 1); }

template <int NV>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[NV]) {
  static_assert(NV % 4 == 0, "pad the value list to a multiple of 4");
#pragma unroll
  for (int p = 0; p < NV / 2; ++p) v[p] = fold_halves(v[2 * p], v[2 * p + 1]);
#pragma unroll
  for (int u = 0; u < NV / 4; ++u) v[u] = fold_row_pairs(v[2 * u], v[2 * u + 1]);
#pragma unroll
  for (int u = 0; u < NV / 4; ++u) v[u] = row_allsum(v[u]);
}

template <int N>
struct IssShape {
  static constexpr int SGR = N > 4 ? 4 : N;                 // sources per reduction group
  static constexpr int NG = (N + SGR - 1) / SGR;            // groups
  static constexpr int NVP = ((3 * SGR + 3) / 4) * 4;       // values per group, padded
  static constexpr int PART = 4 * NG * NVP;                 // doubles per parity buffer (4 waves)
};

// The N sequential rank-1 sweeps of one bin on the register-resident slab y[n][f] (thread-owned
// frames), weights phi[n][f]; `part` is the two-buffer LDS scratch of the block reduction (one
// barrier per sweep), `parity` its state.  ref: ssspy/bss/_update_spatial_model.py:146-194.
// TRACK: lane n also sums log d_n over the sweeps into `ld`.  The sweep of source n multiplies the
// demixing matrix by (I - v e_n^T), whose determinant is 1 - v_n = d_n^(-1/2): the log-determinant of
// the (never formed) filter moves by -1/2 log d_n, which is all compute_loss() needs of it.
// (kept as a product of mantissas and a sum of exponents -- three registers, four instructions per
// sweep -- with one log per workgroup at the end: an fp64 log inside the sweeps cost the N = 8 slab
// kernel, which sits at the register cap, a third of its speed)
struct LogProd {
  double mant;
  int expo;
};
template <int N, int FPT, bool TRACK = false>
__device__ __forceinline__ void iss_sweeps(c128 (&y)[N][FPT], const double (&phi)[N][FPT],
                                           double *part, int &parity, double invT, int floor_kind,
                                           double eps, LogProd *ld = nullptr) {
  using S = IssShape<N>;
  constexpr int SGR = S::SGR, NG = S::NG, NVP = S::NVP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = iss_row_slot(lane >> 4);
#pragma unroll
  for (int n = 0; n < N; ++n) {
    // per source s: sum_j phi_s y_s conj(y_n) (complex) and sum_j phi_s |y_n|^2, SGR sources at a
    // time (register budget)
    double *pp = part + parity * S::PART;
    parity ^= 1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      double red[NVP];
#pragma unroll
      for (int k = 0; k < NVP; ++k) red[k] = 0.0;
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        const c128 yn = y[n][f];
        const double pn = cabs2(yn);
#pragma unroll
        for (int ss = 0; ss < SGR; ++ss) {
          const int s = g * SGR + ss;
          if (s < N) {
            const double w = phi[s][f];
            const c128 z = cmulc(y[s][f], yn);
            red[3 * ss] = fma(w, z.x, red[3 * ss]);
            red[3 * ss + 1] = fma(w, z.y, red[3 * ss + 1]);
            red[3 * ss + 2] = fma(w, pn, red[3 * ss + 2]);
          }
        }
      }
      wave_reduce_scatter<NVP>(red);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int u = 0; u < NVP / 4; ++u) pp[(wave * NG + g) * NVP + 4 * u + slot] = red[u];
      }
    }
    __syncthreads();
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (lane < N) {
      const int g = lane / SGR, ss = lane - g * SGR;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        const double *src = pp + (wv * NG + g) * NVP + 3 * ss;
        t0 += src[0];
        t1 += src[1];
        t2 += src[2];
      }
    }
    const double den = apply_floor(t2 * invT, floor_kind, eps);
    if (TRACK) {  // lane n keeps d_n; the others multiply by one
      const double f = lane == n ? den : 1.0;
      ld->mant *= __builtin_amdgcn_frexp_mant(f);
      ld->expo += __builtin_amdgcn_frexp_exp(f);
    }
#ifdef SSSPY_ISS_IEEE_DIV
    const double vx = lane == n ? 1.0 - 1.0 / sqrt(den) : t0 * invT / den;
    const double vy = lane == n ? 0.0 : t1 * invT / den;
#else
    // The coefficient lane is the serial section of a sweep (every thread waits for it at the
    // readlanes below): v_rcp_f64 / v_rsq_f64 + two Newton steps (~1 ulp, benchmarks/micro/
    // rcp_precision.hip) instead of two IEEE divides and a square root (~90 dependent instructions).
    // (den = 0, Inf or NaN, or so small that 1 / den overflows: the hardware seeds are what the
    //  divide and the square root return there -- Inf, 0, NaN -- and the Newton steps would turn
    //  them into NaN, so those lanes keep the seeds.  Round 6: the branch that called the IEEE
    //  divide and sqrt instead sat in this serial section with its temporaries: hipcc spilled around
    //  it and reloaded right here, a round trip through scratch per sweep.)
    const bool special = !(den > 8.9e-308) || !(den < 1.7976931348623157e308);
    const double rd0 = __builtin_amdgcn_rcp(den);
    double e1 = fma(-den, rd0, 1.0);
    double rd = fma(rd0, e1, rd0);
    e1 = fma(-den, rd, 1.0);
    rd = fma(rd, e1, rd);
    const double rs0 = __builtin_amdgcn_rsq(den);
    double h = 0.5 * den * rs0;
    double e2 = fma(-h, rs0, 0.5);
    double rs = fma(rs0, e2, rs0);
    h = 0.5 * den * rs;
    e2 = fma(-h, rs, 0.5);
    rs = fma(rs, e2, rs);
    rd = special ? rd0 : rd;
    rs = special ? rs0 : rs;
    const double vx = lane == n ? 1.0 - rs : t0 * invT * rd;
    const double vy = lane == n ? 0.0 : t1 * invT * rd;
#endif
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
// TRACK: logdet_delta[b] += sum over this block's bins of -1/2 sum_n log d_n (see iss_sweeps)
template <int N, int FPT, bool PER_BIN, bool TRACK = false>
__global__ __launch_bounds__(256, 2) void k_iss1_fused(c128 *Y, const double *__restrict__ weight,
                                                       double *r2_slabs, int F, int T,
                                                       int bins_per_block, int floor_kind,
                                                       double eps, double *logdet_delta) {
  __shared__ double part[2 * IssShape<N>::PART];
  __shared__ double r2s[N * FPT * 256];  // [n][f][thread]: thread-private slots, no barrier needed
  const int b = blockIdx.y;
  const int i_begin = blockIdx.x * bins_per_block;
  const int i_end = min(F, i_begin + bins_per_block);
  const double invT = 1.0 / (double)T;
  // thread-varying part of every address: one unsigned frame index per owned frame, so the loads
  // and stores take the (scalar row base + 32-bit lane offset) form and no 64-bit address lives in
  // VGPRs across the sweeps
  bool fv[FPT];
  unsigned jj[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) {
    const int j = threadIdx.x + 256 * f;
    fv[f] = j < T;
    jj[f] = fv[f] ? j : T - 1;
  }
  double phi[N][FPT];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      r2s[(n * FPT + f) * 256 + threadIdx.x] = 0.0;
      const double wv = PER_BIN ? 0.0 : weight[((long long)b * N + n) * T + jj[f]];
      phi[n][f] = fv[f] ? wv : 0.0;
    }
  int parity = 0;
  LogProd ld{1.0, 0};
  for (int i = i_begin; i < i_end; ++i) {
    c128 y[N][FPT];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const long long row = (((long long)b * N + n) * F + i) * T;  // wave-uniform
      const __amdgpu_buffer_rsrc_t yr = make_rsrc(Y + row, (unsigned)T * 16u);
      const __amdgpu_buffer_rsrc_t wr = make_rsrc(weight + (PER_BIN ? row : 0), (unsigned)T * 8u);
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
        // frames beyond T re-read frame T-1 with weight 0 and are zeroed (0 * Inf of a non-finite
        // sample must not reach the sums): they add nothing, are never stored, and the loads stay
        // unconditional
#if defined(SSSPY_ISS_DBG) && (SSSPY_ISS_DBG & 2)  // (floor measurement: no slab loads)
        const c128 yv = cmake(1.0 + 1e-3 * (double)(jj[f] + n), 0.5 + 1e-3 * (double)i);
#else
        const c128 yv = buffer_load_c128(yr, jj[f] * 16u);
#endif
        y[n][f] = fv[f] ? yv : c128{0.0, 0.0};
        if (PER_BIN) {
          const double wv = buffer_load_f64(wr, jj[f] * 8u);
          phi[n][f] = fv[f] ? wv : 0.0;
        }
      }
    }
#if !(defined(SSSPY_ISS_DBG) && (SSSPY_ISS_DBG & 1))  // (floor measurement: no sweeps)
    iss_sweeps<N, FPT, TRACK>(y, phi, part, parity, invT, floor_kind, eps, &ld);
#endif
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const long long row = (((long long)b * N + n) * F + i) * T;
      const __amdgpu_buffer_rsrc_t yr = make_rsrc(Y + row, (unsigned)T * 16u);
#pragma unroll
      for (int f = 0; f < FPT; ++f) {
#if defined(SSSPY_ISS_DBG) && (SSSPY_ISS_DBG & 4)  // (floor measurement: no slab stores)
        if (fv[f] && floor_kind == 12345) buffer_store_c128(yr, jj[f] * 16u, y[n][f]);
#else
        if (fv[f]) buffer_store_c128(yr, jj[f] * 16u, y[n][f]);
#endif
        r2s[(n * FPT + f) * 256 + threadIdx.x] += cabs2(y[n][f]);
      }
    }
  }
  if (TRACK) {
    // every wave holds the same sums in its lanes 0..N-1: wave 0 reports
    // (at most 8 bins x 8 sweeps of factors in [1/2, 1): the mantissa product cannot underflow)
    double v = (threadIdx.x < N) ? log(ld.mant) + 0.6931471805599453094 * (double)ld.expo : 0.0;
    if (threadIdx.x < 64) {
      v = wave_sum(v);
      // the block's share goes to its own slot [block][b]; the launcher adds the slots to the
      // caller's running log-determinant in block order (no atomics)
      if (threadIdx.x == 0) logdet_delta[(long long)blockIdx.x * gridDim.y + b] = -0.5 * v;
    }
  }
  if (r2_slabs) {
    // the block's sums over its bins go to its own slab r2_slabs[block][b][n][j]; k_fold_slabs adds
    // the slabs in block order: no fp64 atomics -- the next iteration's weights, and with them the
    // whole trajectory, are the same on every run
    // (through a buffer descriptor: one 32-bit lane offset instead of N FPT 64-bit addresses)
    const int len = N * T;
    const __amdgpu_buffer_rsrc_t sr = make_rsrc(
        r2_slabs + ((long long)blockIdx.x * gridDim.y + b) * len, (unsigned)len * 8u);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int f = 0; f < FPT; ++f)
        if (fv[f])
          buffer_store_f64(sr, ((unsigned)n * (unsigned)T + jj[f]) * 8u,
                           r2s[(n * FPT + f) * 256 + threadIdx.x]);
  }
}

template <int N>
constexpr int iss_max_fpt() {
  return N <= 2 ? 8 : (N <= 4 ? 8 : 4);
}

// bins per block: a few bins amortise the weight loads and the frame-power slab a block leaves
// (N T doubles against N T complex per bin in and out: 1/(4 bpb) of the pass's traffic, and as much
// again when the slabs are added up); keep >= ~4 blocks per CU.  With frame powers requested large
// batches take up to 16 bins per block (3 %).
static int iss_bins_per_block(int B, int F, bool with_r2) {
  const long long want_blocks = 1024;
  int bpb = (int)(((long long)B * F + want_blocks - 1) / want_blocks);
  if (bpb < 1) bpb = 1;
  const int cap = with_r2 ? 16 : 8;
  return bpb > cap ? cap : bpb;
}

template <int N, int FPT>
static int launch_iss(void *Y, const double *weight, bool per_bin, double *r2_slabs, int B, int F,
                      int T, int floor_kind, double eps, double *logdet_delta, hipStream_t st) {
  const int bpb = iss_bins_per_block(B, F, r2_slabs != nullptr);
  dim3 grid((F + bpb - 1) / bpb, B), block(256);
  if (logdet_delta) {  // tracked variants (a separate instantiation: the untracked hot kernels keep
                       // their register allocation)
    if (per_bin)
      hipLaunchKernelGGL((k_iss1_fused<N, FPT, true, true>), grid, block, 0, st, (c128 *)Y, weight,
                         r2_slabs, F, T, bpb, floor_kind, eps, logdet_delta);
    else
      hipLaunchKernelGGL((k_iss1_fused<N, FPT, false, true>), grid, block, 0, st, (c128 *)Y, weight,
                         r2_slabs, F, T, bpb, floor_kind, eps, logdet_delta);
  } else if (per_bin)
    hipLaunchKernelGGL((k_iss1_fused<N, FPT, true>), grid, block, 0, st, (c128 *)Y, weight, r2_slabs,
                       F, T, bpb, floor_kind, eps, (double *)nullptr);
  else
    hipLaunchKernelGGL((k_iss1_fused<N, FPT, false>), grid, block, 0, st, (c128 *)Y, weight, r2_slabs,
                       F, T, bpb, floor_kind, eps, (double *)nullptr);
  return check_launch("k_iss1_fused");
}

template <int N>
static int dispatch_iss(void *Y, const double *weight, bool per_bin, double *r2_slabs, int B,
                        int F, int T, int floor_kind, double eps, double *ld, hipStream_t st) {
  const int fpt = (T + 255) / 256;
  if (fpt <= 1) return launch_iss<N, 1>(Y, weight, per_bin, r2_slabs, B, F, T, floor_kind, eps, ld, st);
  if (fpt <= 2) return launch_iss<N, 2>(Y, weight, per_bin, r2_slabs, B, F, T, floor_kind, eps, ld, st);
  if (fpt <= 4) return launch_iss<N, 4>(Y, weight, per_bin, r2_slabs, B, F, T, floor_kind, eps, ld, st);
  if (iss_max_fpt<N>() >= 8 && fpt <= 8)
    return launch_iss<N, (iss_max_fpt<N>() >= 8 ? 8 : 4)>(Y, weight, per_bin, r2_slabs, B, F,
                                                          T, floor_kind, eps, ld, st);
  return fail(SSSPY_ERR_UNSUPPORTED, "iss1_fused: n_frames too large for the register-resident slab");
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

int ssspy_iss1_fused_max_frames(int N) {
  if (N < 1 || N > SSSPY_MAX_SOURCES) return 0;
  return 256 * (N <= 4 ? 8 : 4);
}

// slabs [block][b][N T], then the scratch of the fold
static size_t iss_r2_layout(int B, int N, int F, int T, size_t *scratch_off) {
  const int bpb = iss_bins_per_block(B, F, true);
  const int nblk = (F + bpb - 1) / bpb;
  const long long total = (long long)B * N * T;
  const size_t slabs = (size_t)nblk * total * sizeof(double);
  if (scratch_off) *scratch_off = slabs;
  return (slabs + fold_scratch_bytes(total, nblk) + 255) & ~(size_t)255;
}
// behind it: the per-block shares of the tracked log-determinant (at most one block per bin)
static size_t iss_logdet_slots_bytes(int B, int F) { return scalar_slots_bytes(B, F); }

size_t ssspy_iss1_fused_workspace_bytes(int B, int N, int F, int T) {
  if (B <= 0 || N <= 0 || F <= 0 || T <= 0) return 0;
  return iss_r2_layout(B, N, F, T, nullptr) + iss_logdet_slots_bytes(B, F);
}

static int iss1_fused_impl(void *Y, const double *weight, int weight_kind, double *r2_next, int B,
                           int N, int F, int T, int floor_kind, double floor_eps,
                           double *logdet_delta, void *workspace, size_t workspace_bytes,
                           void *stream) {
  SSSPY_REQUIRE(Y && weight && B > 0 && F > 0 && T > 0, "iss1_fused: bad argument");
  SSSPY_REQUIRE(weight_kind == SSSPY_WEIGHT_FRAME || weight_kind == SSSPY_WEIGHT_BIN_FRAME,
                "iss1_fused: weight_kind must be FRAME or BIN_FRAME");
  SSSPY_REQUIRE(T <= ssspy_iss1_fused_max_frames(N), "iss1_fused: n_frames above the fused limit");
  size_t scratch_off = 0;
  const size_t r2_bytes = iss_r2_layout(B, N, F, T, &scratch_off);
  const size_t need = r2_bytes + iss_logdet_slots_bytes(B, F);
  SSSPY_REQUIRE((!r2_next && !logdet_delta) || (workspace && workspace_bytes >= need),
                "iss1_fused: frame powers / the tracked log-determinant need the workspace of "
                "ssspy_iss1_fused_workspace_bytes");
  const bool per_bin = weight_kind == SSSPY_WEIGHT_BIN_FRAME;
  hipStream_t st = as_stream(stream);
  double *slabs = r2_next ? (double *)workspace : nullptr;
  void *ld_ws = logdet_delta ? (char *)workspace + r2_bytes : nullptr;  // slots [block][b]
  int rc = SSSPY_OK;
  DISPATCH_N(N, rc = dispatch_iss<NN>(Y, weight, per_bin, slabs, B, F, T, floor_kind, floor_eps,
                                      (double *)ld_ws, st));
  if (rc) return rc;
  const int bpb = iss_bins_per_block(B, F, r2_next != nullptr);
  const int nblk = (F + bpb - 1) / bpb;
  if (logdet_delta) {  // every block wrote its slot: logdet[b] += the blocks' shares, in block order
    rc = scalar_slots_fold(ld_ws, B, nblk, logdet_delta, 1, st);
    if (rc) return rc;
  }
  if (!r2_next) return rc;
  return launch_fold_slabs(slabs, (char *)workspace + scratch_off, r2_next, (long long)B * N * T,
                           nblk, st);
}

int ssspy_iss1_fused(void *Y, const double *weight, int weight_kind, double *r2_next, int B, int N,
                     int F, int T, int floor_kind, double floor_eps, void *workspace,
                     size_t workspace_bytes, void *stream) {
  return iss1_fused_impl(Y, weight, weight_kind, r2_next, B, N, F, T, floor_kind, floor_eps, nullptr,
                         workspace, workspace_bytes, stream);
}

int ssspy_iss1_fused_tracked(void *Y, const double *weight, int weight_kind, double *r2_next, int B,
                             int N, int F, int T, int floor_kind, double floor_eps,
                             double *logdet, void *workspace, size_t workspace_bytes,
                             void *stream) {
  SSSPY_REQUIRE(logdet, "iss1_fused_tracked: bad argument");
  return iss1_fused_impl(Y, weight, weight_kind, r2_next, B, N, F, T, floor_kind, floor_eps, logdet,
                         workspace, workspace_bytes, stream);
}

}  // extern "C"
