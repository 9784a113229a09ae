// Throughput kernels of the ILRMA iteration for the common case
//   n_basis <= 16, n_sources <= 4, source model one of FM_* below (Gauss at domain 2 or 1,
//   Student-t and GGD at domain 2).
// Same math and MFMA tilings as ilrma_kernels.hip (see the header comment there); what
// changes is the scheduling, driven by the first rocprof PMC pass on MI355X (profiles/):
// the generic kernels sat in s_waitcnt 71 % of the time at one wave per SIMD because every
// tile did ~50 dependent global loads between short compute segments.  Here
//   * a workgroup is 4 waves that share the operand tile that changes along the walk:
//     bin-major kernels (basis, covariance): waves own 4 adjacent 16-bin tiles and walk the
//     frame tiles together, the 16x16 activation tile of every source is staged once per
//     workgroup in LDS (double buffered, one barrier per tile);
//     frame-major kernel (activation): waves own 4 adjacent 16-frame tiles and walk the bin
//     tiles together, the basis tile and the demixing matrices are staged in LDS;
//   * basis and activation kernels are on a register diet (<= 256 VGPRs, demixing rows read
//     from LDS) so two workgroups are resident per CU and the second wave of each SIMD fills
//     the MFMA-dependency stalls of the first (measured: 1.66 -> 1.21 ms and 1.35 -> 1.00 ms
//     at 128 mixtures); the covariance kernel keeps 64 fp64 accumulators per lane, cannot fit
//     that budget, and instead prefetches the 16-byte x loads of the NEXT tile into a second
//     register set at one wave per SIMD;
//   * the tile body is straight-line (no uniform branches, no pow) so hipcc can interleave
//     MFMA issue with the fp64 VALU work of the neighbouring source;
//   * 1/R is v_rcp_f64 + 2 Newton steps (~1 ulp) instead of the IEEE divide sequence;
//   * accumulators stay with the wave that owns the bins: no cross-wave fold.
// Compiled once per N (-DSSSPY_N=2..4).
#include <cstdlib>

#include "common.hpp"
#include "cov_core.hpp"
#include "fast_model.hpp"
#include "fast_tiles.hpp"
#include "tail_plan.hpp"

#ifndef SSSPY_N
#error "compile with -DSSSPY_N=<n_sources>"
#endif
#if SSSPY_N > 4
#error "fast path is built for n_sources <= 4"
#endif

#define SSSPY_CAT_(a, b) a##b
#define SSSPY_CAT(a, b) SSSPY_CAT_(a, b)
#define LAUNCHER(name) SSSPY_CAT(SSSPY_CAT(name, _n), SSSPY_N)

namespace ssspy {
namespace SSSPY_CAT(ilrma_fast_n, SSSPY_N) {

constexpr int N = SSSPY_N;
using fast::rcp_nr;
using fast::FastModel;
using fast::FM_GAUSS;
using fast::FM_GAUSS1;
using fast::FM_GAUSSP;
using fast::FM_GGD;
using fast::FM_T;
using fast::LogSum;
using fast::mm_num_factor;
using fast::pow_nonneg;
using fast::ratio_pow;
using fast::rt_from_lds;
using fast::tile_pi;
using fast::VROW;
using fast::XPATCH;
typedef fast::VStage<N> VStage;
typedef fast::XTile<N> XTile;

#ifndef SSSPY_FAST_PART
#define SSSPY_FAST_PART 0  // 0: the whole file; 1: the basis pass only; 2: everything else
#endif
// (the build compiles the basis pass as its own unit: it alone gains from the max-ilp scheduling
//  strategy -- 1.140 -> 1.099 ms at 128 mixtures, the activation pass loses 6 % with it; _build.py)

// IN: what X holds -- IN_X the mixture (filter applied here), IN_Y the separated spectrogram, IN_P its
// power |y|^2 as (B, N, F, T) f64 (the grouped passes of a wide mixture: half the bytes).
enum { IN_Y = 0, IN_X = 1, IN_P = 2 };

#if SSSPY_FAST_PART != 2
// =============================================================================== basis (pass 1)
// grid: (ceil(F/64), 1, B); 256 threads; wave w owns bins [64*bx + 16w, +16).
// Register diet for 2 waves per SIMD (<= 256 VGPR+AGPR): the demixing rows live in LDS and are
// re-read per (source, channel); the second resident workgroup hides the x-load latency that
// the one-wave version covered with a register prefetch.
// grid: 1-D, see TailPlan.  Unsplit blocks update their 64 bins in place; split blocks write
// partial num/den to `part` ([tail item][chunk][n][64 bins][16][2]) for k_basis_finalize.
// HAS_W = false: the ISS / IPA state passes the separated spectrogram itself (y = x_n, no filter)
// LOSS: also accumulate the data term of the negative log-likelihood of the state the pass sees,
// sum_{n,i} mean_j (|y|^2 / R + log R) (model variants as in k_loss_fast; not the t model, whose
// term is not linear in the accumulators), into its loss slot: the pass already forms |y|^2 and R
// under the current (W, T, V), so the loss of iteration t comes out of the basis pass of iteration
// t + 1 instead of a fourth pass over X (ref: ssspy/bss/ilrma.py:1946-1965).
// KS = 4: n_basis <= 16, one work item per (mixture, bin group), updated in place.
// KS = 8: n_basis <= 32: the k-range is two 16-wide tiles and a work item is (mixture, bin group,
// k tile): both items of a group run the whole walk (GEMM1 over all 32 k) and keep the accumulators
// of their own tile, so the pass costs about twice the n_basis <= 16 one -- against 6x for the generic
// kernels.  Its 64-register basis operand only fits because |y|^2 of all sources is formed first
// (the x tile is dead before GEMM1's operands go live); results go to `basis_out` (the sibling
// item still reads the old basis), which the launcher copies back.
template <int IN, int MODEL, bool LOSS, int KS>
__global__ __launch_bounds__(256, KS >= 8 ? 1 : 2) void k_basis_fast(const c128 *__restrict__ X,
                                                       const c128 *__restrict__ W,
                                                       const double *basis, double *basis_out,
                                                       const double *__restrict__ act, int F,
                                                       int T, int K, int floor_kind, double eps,
                                                       TailPlan plan, double *__restrict__ part,
                                                       FastModel fm, double *__restrict__ loss_slots,
                                                       int B) {
  constexpr bool HAS_W = IN == IN_X, PIN = IN == IN_P;
  constexpr int KR = 4 * KS;  // staged activation rows per source
  // Round-4 experiment, off by default (-DSSSPY_BASIS_PREFETCH builds it): spill-free at 210 VGPRs, two
  // waves per SIMD, the x loads fully hidden -- and not faster (1.103 against 1.078 ms at 128
  // mixtures): the pass runs at the 1400 W package cap (profiles/r04_power_cap.md), where the
  // rate is set by the energy of a tile, not by its stalls.
#ifdef SSSPY_BASIS_PREFETCH
  constexpr bool PREFETCH = KS < 8;
#else
  constexpr bool PREFETCH = false;
#endif
  constexpr bool PWF = KS >= 8 || PREFETCH;  // powers first
  // with PREFETCH the GEMM1 basis operand is read from LDS per use (wave-private rows, 8 KB per wave)
  // instead of living in 32 registers: the budget goes to the powers
  constexpr bool TB_LDS = PREFETCH;
  __shared__ __attribute__((aligned(16))) double tls[TB_LDS ? 4 : 1][TB_LDS ? N * 256 : 1];
  __shared__ __attribute__((aligned(16))) double vs[2][N * KR * VROW];
  constexpr int WSTRIDE = N * N + 1;  // 16-byte slots per bin: odd, so 16 bins never share a bank
  __shared__ __attribute__((aligned(16))) c128 wl[4][16 * WSTRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const BlockWork work = block_work(plan);
  const int b = work.b, nchunks = work.nchunks;
  constexpr int KT = KS / 4;  // 16-wide k tiles of the n_basis range: 1, 2 or 4
  // KTI: k tiles ONE item accumulates.  16 < n_basis <= 32: both (GEMM1 over the 32 k once per tile,
  // GEMM2 for both tiles, their accumulators in the one-wave register file); n_basis above 32: one
  // (four items per bin group, each repeating GEMM1).  -DSSSPY_KTILE_ITEMS: KTI = 1, the round-2 scheme of one item
  // per k tile, each repeating GEMM1.
#ifdef SSSPY_KTILE_ITEMS
  constexpr int KTI = 1;
#else
  // (KS = 16: the 128-register GEMM1 operand leaves no room -- two tiles per item spill 74 VGPRs,
  //  four 252 -- so n_basis above 32 keeps one tile per item)
  constexpr int KTI = KS == 8 ? 2 : 1;
#endif
  const int group = work.group / (KT / KTI);
  const int kt0 = (work.group % (KT / KTI)) * KTI;  // first k tile of this item
  const int i0 = group * 64 + wave * 16;
  const int bin = min(i0 + c, F - 1);
  const bool bin_valid = i0 + c < F;
  const int ksteps = (K + 3) >> 2;
  double lacc = 0.0;
  LogSum lr;  // log R of everything this lane visits
  lr.clear();
  const fast::XSrc<N> xsrc =
      PIN ? fast::make_psrc<N>(reinterpret_cast<const double *>(X) + (long long)b * N * F * T, F, T)
          : fast::make_xsrc<N>(X + (long long)b * N * F * T, F, T);
  const double *act_b = act + (long long)b * N * K * T;

  // demixing matrices of the wave's 16 bins -> LDS (wave-private region, filled by the wave)
  for (int e = lane; e < 16 * N * N; e += 64) {
    const int bl = e / (N * N), rem = e % (N * N);
    const int bi = min(i0 + bl, F - 1);
    wl[wave][bl * WSTRIDE + rem] = W ? W[((long long)b * F + bi) * (N * N) + rem]
                                     : cmake((rem / N) == (rem % N) ? 1.0 : 0.0, 0.0);
  }
  const c128 *wmine = wl[wave] + c * WSTRIDE;
  double tb[N][KS];
  if (TB_LDS) {
    fast::stage_basis_rows<N>(tls[wave], basis + (long long)b * N * F * K, F, K, i0, lane);
  } else {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + q;
        tb[n][ks] = kk < K ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
      }
  }
  double4_t num[N][KTI], den[N][KTI];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ti = 0; ti < KTI; ++ti) {
      num[n][ti] = double4_t{0.0, 0.0, 0.0, 0.0};
      den[n][ti] = double4_t{0.0, 0.0, 0.0, 0.0};
    }

  const int ntiles = (T + 15) >> 4;
  const int tpc = (ntiles + nchunks - 1) / nchunks;
  const int jt_begin = work.chunk * tpc, jt_end = min(ntiles, jt_begin + tpc);
  fast::VStage<N, KR> st;
  XTile cur;
  fast::PTile<N> pcur;
  fast::vstage_load<N, KR>(st, act_b, K, T, min(jt_begin, ntiles - 1) * 16);
  fast::vstage_store<N, KR>(st, vs[0]);
  __syncthreads();

  // PREFETCH (n_basis <= 16): |y|^2 of every source is formed FIRST, which is the last use of the x
  // tile, and the same registers are re-loaded with the NEXT tile right away: its 16 loads have the
  // two GEMMs of this tile to land instead of being waited for at the top of the next trip (the
  // register price is the 16 powers kept live through the GEMM phase; the x tile is dead there;
  // fast::pin_after_powers keeps LLVM from sinking the power arithmetic below the re-load).
  if (PREFETCH) {
    if constexpr (PIN) fast::ptile_load_binmajor<N>(pcur, xsrc, T, bin, jt_begin * 16, q);
    else fast::xtile_load_binmajor<N>(cur, xsrc, T, bin, jt_begin * 16, q);
  }
  for (int jt = jt_begin; jt < jt_end; ++jt) {
    const int j0 = jt * 16;
    const int jn = min(jt + 1, jt_end - 1) * 16;  // last iteration re-fetches its own tile
    if (!PREFETCH) {
      if constexpr (PIN) fast::ptile_load_binmajor<N>(pcur, xsrc, T, bin, j0, q);
      else fast::xtile_load_binmajor<N>(cur, xsrc, T, bin, j0, q);
    }
    if (KS < 8) fast::vstage_load<N, KR>(st, act_b, K, T, jn);
    const double *vcur = vs[(jt - jt_begin) & 1];
    double pwall[PWF ? N : 1][4];
    if (PWF) {
      // |y|^2 of every source first: the x tile dies here.  One demixing coefficient at a time and
      // (wide variants) the next activation tile requested only afterwards keep this phase inside
      // the budget.
#pragma unroll
      for (int n = 0; n < N; ++n) {
        c128 y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = HAS_W ? cmake(0.0, 0.0) : cur.x[n][r];
        if (HAS_W) {
#pragma unroll
          for (int m = 0; m < N; ++m) {
            const c128 w = wmine[n * N + m];
#pragma unroll
            for (int r = 0; r < 4; ++r) cfma(y[r], w, cur.x[m][r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pwall[n][r] = PIN ? pcur.p[n][r] : cabs2(y[r]);
      }
      if (KS >= 8) fast::vstage_load<N, KR>(st, act_b, K, T, jn);
      if constexpr (PREFETCH) {
        const int jnp = fast::pin_after_powers<N>(pwall, jn);
        if constexpr (PIN) fast::ptile_load_binmajor<N>(pcur, xsrc, T, bin, jnp, q);
        else fast::xtile_load_binmajor<N>(cur, xsrc, T, bin, jnp, q);
      }
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *vn = vcur + n * KR * VROW;
      const double4_t R = TB_LDS ? rt_from_lds(vn, tls[wave] + n * 256, c, q, ksteps)
                                 : rt_from_lds<KS>(vn, tb[n], c, q, ksteps);
      // GEMM2 B operand: V[n, k = 16 kt + c, frame q + 4r] (slots 4q .. 4q+3 of the permuted row)
      double vb[KTI][4];
#pragma unroll
      for (int ti = 0; ti < KTI; ++ti) {
        const double *vrow = vn + (16 * (kt0 + ti) + c) * VROW + 4 * q;
        const double2 vb01 = *reinterpret_cast<const double2 *>(vrow);
        const double2 vb23 = *reinterpret_cast<const double2 *>(vrow + 2);
        vb[ti][0] = vb01.x;
        vb[ti][1] = vb01.y;
        vb[ti][2] = vb23.x;
        vb[ti][3] = vb23.y;
      }
      c128 wr[N];
      if (!PWF) {
#pragma unroll
        for (int m = 0; m < N; ++m) wr[m] = wmine[n * N + m];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double pw;
        if (PWF) {
          pw = pwall[n][r];
        } else {
          c128 y = cur.x[n][r];
          if (HAS_W) {
            y = cmake(0.0, 0.0);
#pragma unroll
            for (int m = 0; m < N; ++m) cfma(y, wr[m], cur.x[m][r]);
          }
          pw = PIN ? pcur.p[n][r] : cabs2(y);
        }
#ifdef SSSPY_ASSUME_FULL
        const bool valid = true;
#else
        const bool valid = j0 + q + 4 * r < T;
#endif
        const double rinv = rcp_nr(R[r]);
        const double bb = valid ? rinv : 0.0;
        const double aa = valid ? mm_num_factor<MODEL>(pw, R[r], rinv, fm) : 0.0;
#pragma unroll
        for (int ti = 0; ti < KTI; ++ti) {
          num[n][ti] = mfma_f64(aa, vb[ti][r], num[n][ti]);
          den[n][ti] = mfma_f64(bb, vb[ti][r], den[n][ti]);
        }
        if (LOSS) {
          // log R of every element: LogSum, no log here.  The P/R part needs nothing per element:
          // see the epilogue.
          const bool lv = valid && bin_valid;
          lr.mul(lv ? R[r] : 1.0);
        }
      }
    }
    if (LOSS) lr.renorm();
    fast::vstage_store<N, KR>(st, vs[(jt - jt_begin + 1) & 1]);
    __syncthreads();
  }
  // D: col = basis index 16 kt + c, row = q + 4r -> bin i0 + q + 4r
  // Loss by-product: sum_j a_nij R_nij = sum_k t_nik num_nik with the basis the pass started from, and
  // a R is the model's data term up to a constant (Gauss P/R, domain 1 P/R^2, GGD (beta/2)(P/R)^(beta/2)):
  // it falls out of the finished accumulators (the split items' share is added by k_basis_finalize).
#pragma unroll
  for (int ti = 0; ti < KTI; ++ti) {
    const int kout = 16 * (kt0 + ti) + c;
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ob = i0 + q + 4 * r;
        if (ob < F && kout < K) {
          if (nchunks == 1) {
            const long long o = (((long long)b * N + n) * F + ob) * K + kout;
            const double told = basis[o];
            if (LOSS) lacc = fma(told, num[n][ti][r], lacc);
            const double ratio = num[n][ti][r] / den[n][ti][r];
            basis_out[o] = apply_floor(ratio_pow(ratio, fm.expo) * told, floor_kind, eps);
          } else {
            // partial sums of a split item: one 16-k record per (item, chunk, k tile of the item)
            const long long slot = ((long long)work.tail_idx * nchunks + work.chunk) * KTI + ti;
            double *dst = part + (((slot * N + n) * 64 + (ob - group * 64)) * 16 + c) * 2;
            dst[0] = num[n][ti][r];
            dst[1] = den[n][ti][r];
          }
        }
      }
  }
  if (LOSS) {
    if (MODEL == FM_GGD) lacc *= 2.0 / fm.beta;
    if (kt0 == 0)  // (2 / p) log R, once
      lacc += (MODEL == FM_GAUSS1 ? 2.0 : (MODEL == FM_GAUSSP ? fm.pinv2 : 1.0)) * lr.value();
    lacc = wave_sum(lacc);
    // no atomics: every (bin group, chunk, wave) of a mixture owns a slot (loss_slot_count())
    const int maxsplit = plan.split > 1 ? plan.split : 1;
    if (lane == 0)
      loss_slots[(long long)((work.group * maxsplit + work.chunk) * 4 + wave) * B + b] =
          lacc / (double)T;
  }
}

// basis <- floor(basis * sqrt(sum_chunks num / sum_chunks den)) for the split (tail) items;
// grid: (N*64*16/256, tail items); one thread per (n, local bin, k)
// ktiles: 1 (n_basis <= 16), 2 (<= 32) or 4 (<= 64): the item index also carries the k tile, see
// k_basis_fast<.., 8 | 16>
// loss_slots / loss_scale: the split items' share of the loss by-product sum_k t num (see k_basis_fast)
__global__ __launch_bounds__(256) void k_basis_finalize(const double *basis, double *basis_out,
                                                        const double *__restrict__ part, int F,
                                                        int K, TailPlan plan, int ktiles,
                                                        int floor_kind, double eps, double expo,
                                                        double *__restrict__ loss_slots,
                                                        double loss_scale, int B, int slot0) {
  const int tail_idx = blockIdx.y;
  const int item = plan.full + tail_idx;
  const int b = item / plan.groups, g2 = item - b * plan.groups;
  // ktiles > 0: the item index carries the k tile (one record per item and chunk);
  // ktiles < 0: |ktiles| tiles inside every item, this launch's blockIdx.z picks one
  const int kti = ktiles < 0 ? -ktiles : 1;
  const int group = ktiles < 0 ? g2 : g2 / ktiles;
  const int kt = ktiles < 0 ? (int)blockIdx.z : g2 % ktiles;
  const int ti = ktiles < 0 ? (int)blockIdx.z : 0;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // (n, local bin, k16)
  const int k = 16 * kt + (e & 15), lb = (e >> 4) & 63, n = e >> 10;
  const int bin = group * 64 + lb;
  double contrib = 0.0;
  if (n < N && k < K && bin < F) {
    const long long o = (((long long)b * N + n) * F + bin) * K + k;
    const double told = basis[o];
    // chunks in order, eight loads per round trip (ordered_sum, common.hpp)
    const double2 s2 = ordered_sum(reinterpret_cast<const double2 *>(part) +
                                       ((((long long)tail_idx * plan.split) * kti + ti) * N + n) * 1024 +
                                       (e & 1023),
                                   (long long)kti * N * 1024, plan.split);
    const double sn = s2.x, sd = s2.y;
    contrib = told * sn;
    const double ratio = sn / sd;
    basis_out[o] = apply_floor(ratio_pow(ratio, expo) * told, floor_kind, eps);
  }
  if (loss_slots) {  // uniform per launch; slots behind those of k_basis_fast
    contrib = wave_sum(contrib);
    if ((threadIdx.x & 63) == 0)
      loss_slots[(long long)(slot0 + (g2 * (int)gridDim.x + (int)blockIdx.x) * 4 +
                             (int)(threadIdx.x >> 6)) * B + b] = contrib * loss_scale;
  }
}

#endif  // basis part

#if SSSPY_FAST_PART != 1
// ================================================================================ loss data
// sum_{n,i} mean_j ( |y|^2 / R + log R ), the data term of compute_loss(), as one slot per wave (the
// launcher folds them into out[b] in a fixed order).  Same walk as
// the basis pass without its second GEMM; the logarithms are summed as a mantissa product and an
// exponent sum (LogSum): no fp64 log in the walk at all.
// HAS_W = false: the ISS / IPA state passes the separated spectrogram itself (y = x_n, no filter)
template <bool HAS_W, int MODEL>
__global__ __launch_bounds__(256, 2) void k_loss_fast(const c128 *__restrict__ X,
                                                      const c128 *__restrict__ W,
                                                      const double *__restrict__ basis,
                                                      const double *__restrict__ act,
                                                      double *__restrict__ slots, int F, int T,
                                                      int K, TailPlan plan, FastModel fm, int B) {
  __shared__ __attribute__((aligned(16))) double vs[2][N * 16 * VROW];
  constexpr int WSTRIDE = N * N + 1;
  __shared__ __attribute__((aligned(16))) c128 wl[4][16 * WSTRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const BlockWork work = block_work(plan);
  const int b = work.b, nchunks = work.nchunks;
  const int i0 = work.group * 64 + wave * 16;
  const bool bin_valid = i0 + c < F;
  const int bin = min(i0 + c, F - 1);
  const fast::XSrc<N> xsrc = fast::make_xsrc<N>(X + (long long)b * N * F * T, F, T);
  const double *act_b = act + (long long)b * N * K * T;
  for (int e = lane; e < 16 * N * N; e += 64) {
    const int bl = e / (N * N), rem = e % (N * N);
    const int bi = min(i0 + bl, F - 1);
    wl[wave][bl * WSTRIDE + rem] = W ? W[((long long)b * F + bi) * (N * N) + rem]
                                     : cmake((rem / N) == (rem % N) ? 1.0 : 0.0, 0.0);
  }
  const c128 *wmine = wl[wave] + c * WSTRIDE;
  double tb[N][4];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = 4 * ks + q;
      tb[n][ks] = kk < K ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  const int ntiles = (T + 15) >> 4;
  const int tpc = (ntiles + nchunks - 1) / nchunks;
  const int jt_begin = work.chunk * tpc, jt_end = min(ntiles, jt_begin + tpc);
  VStage st;
  XTile cur;
  fast::vstage_load<N>(st, act_b, K, T, min(jt_begin, ntiles - 1) * 16);
  fast::vstage_store<N>(st, vs[0]);
  __syncthreads();
  double acc = 0.0;
  LogSum lr, lt;
  lr.clear();
  lt.clear();
  for (int jt = jt_begin; jt < jt_end; ++jt) {
    const int j0 = jt * 16;
    const int jn = min(jt + 1, jt_end - 1) * 16;
    fast::xtile_load_binmajor<N>(cur, xsrc, T, bin, j0, q);
    fast::vstage_load<N>(st, act_b, K, T, jn);
    const double *vcur = vs[(jt - jt_begin) & 1];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double4_t R = rt_from_lds(vcur + n * 16 * VROW, tb[n], c, q, (K + 3) >> 2);
      c128 wr[N];
#pragma unroll
      for (int m = 0; m < N; ++m) wr[m] = wmine[n * N + m];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c128 y = cur.x[n][r];
        if (HAS_W) {
          y = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < N; ++m) cfma(y, wr[m], cur.x[m][r]);
        }
        const bool valid = bin_valid && (j0 + q + 4 * r < T);
        const double rr = valid ? R[r] : 1.0;
        const double ri = rcp_nr(rr);
        const double pr = (valid ? cabs2(y) : 0.0) * ri;
        if (MODEL == FM_T)
          lt.mul(fma(2.0 / fm.nu, pr, 1.0));  // (1 + nu/2) log(1 + (2/nu) P / R), ilrma.py:3301-3305
        else if (MODEL == FM_GGD)
          acc += pow_nonneg(pr, 0.5 * fm.beta);  // (P / R)^(beta/2), ilrma.py:4377-4381
        else if (MODEL == FM_GAUSS1)
          acc += pr * ri;                        // P / R^2
        else if (MODEL == FM_GAUSSP)
          acc += (valid ? cabs2(y) : 0.0) * pow_nonneg(rr, -fm.pinv2);  // P / R^(2/p)
        else
          acc += pr;
        lr.mul(rr);
      }
    }
    lr.renorm();
    if (MODEL == FM_T) lt.renorm();
    fast::vstage_store<N>(st, vs[(jt - jt_begin + 1) & 1]);
    __syncthreads();
  }
  acc += (MODEL == FM_GAUSS1 ? 2.0 : (MODEL == FM_GAUSSP ? fm.pinv2 : 1.0)) * lr.value();  // (2 / p) log R
  if (MODEL == FM_T) acc = fma(1.0 + 0.5 * fm.nu, lt.value(), acc);
  acc = wave_sum(acc);
  // no atomics: one slot per (bin group, chunk, wave) of the mixture, folded in order afterwards
  const int maxsplit = plan.split > 1 ? plan.split : 1;
  if (lane == 0)
    slots[(long long)((work.group * maxsplit + work.chunk) * 4 + wave) * B + b] = acc / (double)T;
}

// ================================================================== weighted covariance (pass 3)
// U[b,i,n] = (1/T) sum_j x x^H / R.
// The N Hermitian accumulators of a bin (N*N*N reals) do not fit a 256-register budget next to the
// x tile, so the weight sets (sources) are split over waves: a workgroup is WB bin tiles x NG
// source groups of SG sources (N = 4: 2 bin tiles x 2 groups of 2).  Each wave recomputes the
// x x^H products it needs; the x loads of the two group-waves hit the same L1 lines.
// fp64 MFMA and fp64 VALU do not overlap on gfx950 (benchmarks/micro/f64_rates.hip), so what the
// second resident wave per SIMD hides is LDS / HBM latency, not arithmetic.
// Round 3 built the alternative -- a wave owning its bin tile for ALL four sources (no duplicate
// x x^H products: -28 % VALU instructions; one fetch of the x tile: half the requests) -- three
// ways and measured each at 128 mixtures against this kernel's 0.92 ms: (i) two waves per SIMD
// with the x tile fetched by LDS-direct loads (buffer_load_dwordx4 ... lds, bank-conflict-free
// through a swizzle on the source side, no staging registers): 1.86 ms -- at the 256-register
// budget hipcc parks the GEMM1 operand in scratch and reloads it before every MFMA behind a full
// s_waitcnt; (ii) one wave per SIMD (no spills), LDS-direct loads double-buffered one tile ahead:
// 1.97 ms; (iii) one wave per SIMD with the next tile prefetched into a second register set:
// 2.00 ms, of which 1.45 ms remain with the x loads removed altogether -- a single wave per SIMD
// exposes the latency of every dependent fp64 chain (reciprocal + Newton steps, accumulator
// round trips through AGPRs, the per-frame LDS reads), which the second wave of the split form
// hides.  Correct in all three forms (parity suite), not kept.
constexpr int WC_SG = N >= 4 ? 2 : N;            // sources per wave
constexpr int WC_NG = (N + WC_SG - 1) / WC_SG;   // source groups
constexpr int WC_WB = 4 / WC_NG;                 // bin tiles per workgroup

constexpr int WC_BINS = 16 * WC_WB;              // bins per workgroup
// two source groups (4 channels): the waves of a bin tile share its x tile (xtile_load_shared)
#ifdef SSSPY_WCOV_PRIVATE_TILE
constexpr bool WC_SHARE = false;
#else
constexpr bool WC_SHARE = WC_NG == 2 && N == 4;
#endif
static_assert(!WC_SHARE || WC_WB * N * 16 * 17 <= 4 * XPATCH, "shared x patches fit the array");

// grid: 1-D, see TailPlan.  Unsplit blocks store U directly; split blocks store their partial sums
// (already scaled by 1/T) to `upart` ([tail item][chunk][WC_BINS][N][N][N]) for k_wcov_fold.
// t and GGD models: varphi depends on |w_n^H x|^2, so the wave also needs its bins' demixing rows
// KS: k-steps of GEMM1 compiled in (4: n_basis <= 16, 8: n_basis <= 32; no k tiles here, the pass has
// no second GEMM)
template <int MODEL, int KS>
__global__ __launch_bounds__(256, KS >= 16 ? 1 : 2) void k_wcov_fast(const c128 *__restrict__ X,
                                                      const c128 *__restrict__ W,
                                                      const double *__restrict__ basis,
                                                      const double *__restrict__ act,
                                                      c128 *__restrict__ U, int F, int T, int K,
                                                      TailPlan plan, c128 *__restrict__ upart,
                                                      FastModel fm) {
  constexpr int SG = WC_SG;
  constexpr int WSTRIDE = N * N + 1;
  __shared__ __attribute__((aligned(16))) c128 xpatch[4][XPATCH];
  constexpr bool NEEDS_Y = MODEL == FM_T || MODEL == FM_GGD;
  __shared__ __attribute__((aligned(16))) c128 wlc[NEEDS_Y ? 4 * 16 * WSTRIDE : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const BlockWork work = block_work(plan);
  const int b = work.b, nchunks = work.nchunks;
  const int g = wave % WC_NG, wb = wave / WC_NG;
  const int s0 = g * SG;
  const int i0 = (work.group * WC_WB + wb) * 16;
  const int bin = min(i0 + c, F - 1);
  const fast::XSrc<N> xsrc = fast::make_xsrc<N>(X + (long long)b * N * F * T, F, T);
  c128 *xshare = &xpatch[0][0] + wb * (N * 16 * 17);  // WC_SHARE: one patch per bin tile
  const double *act_b = act + (long long)b * N * K * T;
  double tb[SG][KS];
#pragma unroll
  for (int s = 0; s < SG; ++s)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = 4 * ks + q, n = min(s0 + s, N - 1);
      tb[s][ks] = kk < K ? basis[(((long long)b * N + n) * F + bin) * K + kk] : 0.0;
    }
  c128 *wmine = wlc + (NEEDS_Y ? (wave * 16 + c) * WSTRIDE : 0);
  if (NEEDS_Y) {
    for (int e = lane; e < 16 * N * N; e += 64) {
      const int bl = e / (N * N), rem = e % (N * N);
      wlc[(wave * 16 + bl) * WSTRIDE + rem] = W[((long long)b * F + min(i0 + bl, F - 1)) * (N * N) + rem];
    }
  }
  CovAcc<N, SG> acc;
  acc.clear();
  const int ntiles = (T + 15) >> 4;
  const int tpc = (ntiles + nchunks - 1) / nchunks;
  const int jt_begin = work.chunk * tpc, jt_end = min(ntiles, jt_begin + tpc);
  XTile cur;
  if (NEEDS_Y) __syncthreads();
  for (int jt = jt_begin; jt < jt_end; ++jt) {
    const int j0 = jt * 16;
    // GEMM1 A operand straight from global memory: V[n, 4ks+q, j0+c] (the activation of a mixture
    // is 0.26 MB, shared by all its items on this XCD: L2 hits), 128-byte runs per (ks, q)
    double va[SG][KS];
    const int jv = j0 + c;
#pragma unroll
    for (int s = 0; s < SG; ++s)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kk = 4 * ks + q, n = min(s0 + s, N - 1);
        va[s][ks] = (kk < K && jv < T) ? act_b[((long long)n * K + kk) * T + jv] : 0.0;
      }
    if constexpr (WC_SHARE) {
      // the two source-group waves of a bin tile fetch HALF of its channels each and exchange them
      // through the tile's LDS patch: every byte of x enters the CU once
      fast::xtile_load_shared<N, WC_NG>(cur, xsrc, T, i0, j0, c, q,
                                         __builtin_amdgcn_readfirstlane(g), xshare);
    } else {
      fast::xtile_load_transposed<N>(cur, xsrc, T, i0, j0, c, q, xpatch[wave]);
    }
    double4_t R[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      R[s] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (ks < ((K + 3) >> 2)) R[s] = mfma_f64(va[s][ks], tb[s][ks], R[s]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool valid = j0 + q + 4 * r < T;
      c128 x[N];
      double phi[SG];
#pragma unroll
      for (int m = 0; m < N; ++m) x[m] = cur.x[m][r];
#pragma unroll
      for (int s = 0; s < SG; ++s) {
        double den = R[s][r];
        if (NEEDS_Y) {
          c128 y = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < N; ++m) cfma(y, wmine[min(s0 + s, N - 1) * N + m], x[m]);
          const double pw = cabs2(y);
          if (MODEL == FM_T) {
            den = fma(fm.w, den, fm.w1 * pw);
          } else {  // GGD: (2 / beta) floor(P^((2 - beta) / 2)) R^(beta / 2)
            const double y2b = apply_floor(pow_nonneg(pw, 0.5 * (2.0 - fm.beta)), fm.floor_kind,
                                           fm.floor_eps);
            den = (2.0 / fm.beta) * y2b * pow_nonneg(den, 0.5 * fm.beta);
          }
        }
        if (MODEL == FM_GAUSS1) den = den * den;  // R^(2/p), p = 1
        const double ph = MODEL == FM_GAUSSP ? pow_nonneg(den, -fm.pinv2) : rcp_nr(den);
        phi[s] = (valid && s0 + s < N) ? ph : 0.0;
      }
      acc.add(x, phi);
    }
  }
  acc.fold_q();
  // every q-lane holds the full sums of bin i0+c for this wave's SG sources; spread the stores:
  // lane q writes rows a with (a & 3) == q
  const double scale = 1.0 / (double)T;
  const int ob = i0 + c;
  if (ob < F) {
    c128 *dst = nchunks == 1
                    ? U + ((long long)b * F + ob) * (long long)(N * N * N)
                    : upart + (((long long)work.tail_idx * nchunks + work.chunk) * WC_BINS +
                               (ob - work.group * WC_BINS)) * (long long)(N * N * N);
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int n = s0 + s;
      if (n < N) {
        int e = 0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
          const bool mine = (a & 3) == q;
          if (mine) dst[(n * N + a) * N + a] = cmake(acc.diag[s][a] * scale, 0.0);
#pragma unroll
          for (int bb = a + 1; bb < N; ++bb) {
            const c128 z = acc.off[s][e];
            if (mine) {
              dst[(n * N + a) * N + bb] = cmake(z.x * scale, z.y * scale);
              dst[(n * N + bb) * N + a] = cmake(z.x * scale, -z.y * scale);
            }
            ++e;
          }
        }
      }
    }
  }
}

// The same pass with per-frame weights phi[b, s, j] (B, N, T) instead of the NMF variance: the
// weighted covariance of AuxIVA (ref: ssspy/bss/iva.py:1816-1840).  No activation tile to stage, so
// no barrier in the walk; items are never split here (the launcher takes this kernel only when the
// batch fills the chip, see ilrma_fast_wcov_frame).  grid: B * ceil(F / WC_BINS) workgroups.
__global__ __launch_bounds__(256, 2) void k_wcov_frame_fast(const c128 *__restrict__ X,
                                                            const double *__restrict__ weight,
                                                            c128 *__restrict__ U, int F, int T,
                                                            int groups) {
  constexpr int SG = WC_SG;
  __shared__ __attribute__((aligned(16))) c128 xpatch[4][XPATCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int item = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = item / groups, group = item - b * groups;
  const int g = wave % WC_NG, wb = wave / WC_NG;
  const int s0 = g * SG;
  const int i0 = (group * WC_WB + wb) * 16;
  const c128 *Xb = X + (long long)b * N * F * T;  // flat loads: measured faster here (A/B, round 2)
  const fast::XSrc<N> xsrc = fast::make_xsrc<N>(Xb, F, T);
  c128 *xshare = &xpatch[0][0] + wb * (N * 16 * 17);  // WC_SHARE: one patch per bin tile
  const double *wgt = weight + (long long)b * N * T;
  CovAcc<N, SG> acc;
  acc.clear();
  const int ntiles = (T + 15) >> 4;
  XTile cur;
  for (int jt = 0; jt < ntiles; ++jt) {
    const int j0 = jt * 16;
    if constexpr (WC_SHARE) {
      fast::xtile_load_shared<N, WC_NG>(cur, xsrc, T, i0, j0, c, q,
                                         __builtin_amdgcn_readfirstlane(g), xshare);
    } else {
      fast::xtile_load_transposed<N>(cur, Xb, F, T, i0, j0, c, q, xpatch[wave]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + q + 4 * r;
      const bool valid = j < T;
      c128 x[N];
      double phi[SG];
#pragma unroll
      for (int m = 0; m < N; ++m) x[m] = cur.x[m][r];
#pragma unroll
      for (int s = 0; s < SG; ++s)
        phi[s] = (valid && s0 + s < N) ? wgt[(long long)(s0 + s) * T + j] : 0.0;
      acc.add(x, phi);
    }
  }
  acc.fold_q();
  const double scale = 1.0 / (double)T;
  const int ob = i0 + c;
  if (ob < F) {
    c128 *dst = U + ((long long)b * F + ob) * (long long)(N * N * N);
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const int n = s0 + s;
      if (n < N) {
        int e = 0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
          const bool mine = (a & 3) == q;
          if (mine) dst[(n * N + a) * N + a] = cmake(acc.diag[s][a] * scale, 0.0);
#pragma unroll
          for (int bb = a + 1; bb < N; ++bb) {
            const c128 z = acc.off[s][e];
            if (mine) {
              dst[(n * N + a) * N + bb] = cmake(z.x * scale, z.y * scale);
              dst[(n * N + bb) * N + a] = cmake(z.x * scale, -z.y * scale);
            }
            ++e;
          }
        }
      }
    }
  }
}

// U[tail items] = sum of their chunks; grid: (WC_BINS*N^3/256 rounded up, tail items)
__global__ __launch_bounds__(256) void k_wcov_fold(c128 *__restrict__ U,
                                                   const c128 *__restrict__ upart, int F,
                                                   TailPlan plan) {
  constexpr int PER = WC_BINS * N * N * N;
  const int tail_idx = blockIdx.y;
  const int item = plan.full + tail_idx;
  const int b = item / plan.groups, group = item - b * plan.groups;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int lb = e / (N * N * N);
  if (e >= PER || group * WC_BINS + lb >= F) return;
  const c128 s2 = ordered_sum(upart + (long long)tail_idx * plan.split * PER + e, (long long)PER,
                              plan.split);  // chunks in order, eight loads per round trip
  U[((long long)b * F + group * WC_BINS) * (long long)(N * N * N) + e] = s2;
}

// ========================================================================= activation (pass 2)
// grid: (ceil(T/64), chunks, B); wave w owns frames [64*bx + 16w, +16) and walks the bin tiles
// of its chunk; the basis tile (all sources) and the 16 demixing matrices are staged in LDS.
template <int KS>
struct TStage {
  double t[(N * 16 * 4 * KS + 255) / 256];
  c128 w;
};

// staged basis rows: [n][bin 16][k < 4 KS] with one slot of padding per row
template <int KS>
constexpr int trow() {
  return 4 * KS + 1;
}

template <int KS>
__device__ __forceinline__ void tstage_load(TStage<KS> &st, const double *__restrict__ basis_b,
                                            const c128 *__restrict__ W_b, int F, int K, int i0) {
  constexpr int KR = 4 * KS;
#pragma unroll
  for (int u = 0; u < (N * 16 * KR + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;  // (n, bin, k)
    const int k = idx % KR, bl = (idx / KR) & 15, n = idx / (16 * KR);
    const int bi = i0 + bl;
    double v = 0.0;
    if (idx < N * 16 * KR && k < K && bi < F) v = basis_b[((long long)n * F + bi) * K + k];
    st.t[u] = v;
  }
  {
    const int idx = threadIdx.x;  // (bin, n, m), N*N*16 <= 256
    const int bl = idx / (N * N), rem = idx % (N * N);
    const int bi = min(i0 + bl, F - 1);
    c128 v = cmake(0.0, 0.0);
    if (idx < 16 * N * N)
      v = W_b ? W_b[(long long)bi * (N * N) + rem] : cmake((rem / N) == (rem % N) ? 1.0 : 0.0, 0.0);
    st.w = v;
  }
}

constexpr int AWSTRIDE = N * N + 1;  // 16-byte slots per staged demixing matrix

template <int KS>
__device__ __forceinline__ void tstage_store(const TStage<KS> &st, double *tbuf, c128 *wbuf) {
  constexpr int KR = 4 * KS;
#pragma unroll
  for (int u = 0; u < (N * 16 * KR + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int k = idx % KR, row = idx / KR;  // row = n*16 + bin
    if (idx < N * 16 * KR) tbuf[row * trow<KS>() + k] = st.t[u];
  }
  // one 16-byte slot of padding per bin: lanes of different q read different bins at once, and an
  // unpadded N*N-slot stride (256 B at N = 4) would put them all on the same banks
  if (threadIdx.x < 16 * N * N)
    wbuf[(threadIdx.x / (N * N)) * AWSTRIDE + threadIdx.x % (N * N)] = st.w;
}

// HAS_W = false: the ISS / IPA state passes the separated spectrogram itself (y = x_n, no filter)
// KS = 8 / 16 (16 < n_basis <= 32 / 64): grid.y carries (bin chunk, k tile); every k-tile item runs
// GEMM1 over all k and keeps the sums of its own 16 (see k_basis_fast); one wave per SIMD.
template <int IN, int MODEL, int KS>
__global__ __launch_bounds__(256, KS >= 8 ? 1 : 2) void k_activation_fast(
    const c128 *__restrict__ X, const c128 *__restrict__ W, const double *__restrict__ basis,
    const double *__restrict__ act, double *__restrict__ part, int F, int T, int K,
    int tiles_per_chunk, int nchunks, FastModel fm) {
  constexpr bool HAS_W = IN == IN_X, PIN = IN == IN_P;
  constexpr int TROW = trow<KS>();
  __shared__ __attribute__((aligned(16))) double ts[2][N * 16 * TROW];
  __shared__ __attribute__((aligned(16))) c128 ws[2][16 * AWSTRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  // (frame group, bin chunk [x k tile], mixture) from the XCD-contiguous item: the frame groups of
  // one (mixture, chunk) share the staged basis tiles and demixing matrices
  const GridItem gi = xcd_contiguous_grid();
  constexpr int KT = KS / 4;  // k tiles of the n_basis range
  // k tiles one item accumulates: both at 16 < n_basis <= 32, one above (see k_basis_fast);
  // grid.y = chunks x (KT / KTI)
#ifdef SSSPY_KTILE_ITEMS
  constexpr int KTI = 1;
#else
  constexpr int KTI = KS == 8 ? 2 : 1;
#endif
  const int chunk = gi.y / (KT / KTI), b = gi.z;
  const int kt0 = (gi.y % (KT / KTI)) * KTI;
  const int ksteps = (K + 3) >> 2;
  const int j0 = (gi.x * 4 + wave) * 16;
  const int jf = j0 + c;
  const bool fvalid = jf < T;
  const int jc = fvalid ? jf : T - 1;
  const fast::XSrc<N> xsrc =
      PIN ? fast::make_psrc<N>(reinterpret_cast<const double *>(X) + (long long)b * N * F * T, F, T)
          : fast::make_xsrc<N>(X + (long long)b * N * F * T, F, T);
  const double *basis_b = basis + (long long)b * N * F * K;
  const c128 *W_b = W ? W + (long long)b * F * N * N : nullptr;

  double vb[N][KS];  // GEMM1 B operand V[n, 4ks+q, frame]
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = 4 * ks + q;
      vb[n][ks] = (kk < K && fvalid) ? act[(((long long)b * N + n) * K + kk) * T + jc] : 0.0;
    }
  double4_t numv[N][KTI], denv[N][KTI];
#pragma unroll
  for (int n = 0; n < N; ++n)
#pragma unroll
    for (int ti = 0; ti < KTI; ++ti) {
      numv[n][ti] = double4_t{0.0, 0.0, 0.0, 0.0};
      denv[n][ti] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
  const int ntiles = (F + 15) >> 4;
  const int t_begin = chunk * tiles_per_chunk;
  const int t_end = min(ntiles, t_begin + tiles_per_chunk);
  TStage<KS> st;
  XTile cur;
  fast::PTile<N> pcur;
  tstage_load<KS>(st, basis_b, W_b, F, K, t_begin * 16);
  tstage_store<KS>(st, ts[0], ws[0]);
  __syncthreads();
  for (int it = t_begin; it < t_end; ++it) {
    const int i0 = it * 16;
    const int in = min(it + 1, t_end - 1) * 16;
    if constexpr (PIN) fast::ptile_load_framemajor<N>(pcur, xsrc, T, i0, jc, q);
    else fast::xtile_load_framemajor<N>(cur, xsrc, T, i0, jc, q);
    tstage_load<KS>(st, basis_b, W_b, F, K, in);
    const int pb = (it - t_begin) & 1;
    const double *tcur = ts[pb];
    const c128 *wcur = ws[pb];
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const double *tn = tcur + n * 16 * TROW;
      // GEMM1: A[row = c -> bin i0+c][kk = q] = T[n, i0+c, 4ks+q]
      double4_t R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (ks < ksteps) R = mfma_f64(tn[c * TROW + 4 * ks + q], vb[n][ks], R);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bl = q + 4 * r;
        const c128 *wr = wcur + bl * AWSTRIDE + n * N;
        c128 y = cur.x[n][r];
        if (HAS_W) {
          y = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < N; ++m) cfma(y, wr[m], cur.x[m][r]);
        }
        const bool valid = fvalid && (i0 + bl < F);
        const double rinv = rcp_nr(R[r]);
        const double bb = valid ? rinv : 0.0;
        const double pw = PIN ? pcur.p[n][r] : cabs2(y);
        const double aa = valid ? mm_num_factor<MODEL>(pw, R[r], rinv, fm) : 0.0;
        // GEMM2: A[row = c -> basis index 16 kt + c][kk = q] = T[n, bin i0+q+4r, 16 kt + c]
        // (zero-staged pads)
#pragma unroll
        for (int ti = 0; ti < KTI; ++ti) {
          const double ta = tn[bl * TROW + 16 * (kt0 + ti) + c];
          numv[n][ti] = mfma_f64(ta, aa, numv[n][ti]);
          denv[n][ti] = mfma_f64(ta, bb, denv[n][ti]);
        }
      }
    }
    tstage_store<KS>(st, ts[pb ^ 1], ws[pb ^ 1]);
    __syncthreads();
  }
#pragma unroll
  for (int ti = 0; ti < KTI; ++ti)
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ok = 16 * (kt0 + ti) + q + 4 * r;
        if (ok < K && fvalid) {
          const long long base = ((((long long)b * nchunks + chunk) * N + n) * 2) * K;
          part[(base + ok) * T + jf] = numv[n][ti][r];
          part[(base + K + ok) * T + jf] = denv[n][ti][r];
        }
      }
}

#endif  // other passes
}  // namespace ilrma_fast_n<N>
using namespace SSSPY_CAT(ilrma_fast_n, SSSPY_N);
using fast::make_fast_model;

#define SSSPY_FAST_LAUNCH_M(kernel, HW, ...)                                              \
  switch (fmodel) {                                                                       \
    case FM_T: hipLaunchKernelGGL((kernel<HW, FM_T>), grid, block, 0, st, __VA_ARGS__); break;     \
    case FM_GGD: hipLaunchKernelGGL((kernel<HW, FM_GGD>), grid, block, 0, st, __VA_ARGS__); break; \
    case FM_GAUSS1:                                                                       \
      hipLaunchKernelGGL((kernel<HW, FM_GAUSS1>), grid, block, 0, st, __VA_ARGS__);       \
      break;                                                                              \
    case FM_GAUSSP:                                                                       \
      hipLaunchKernelGGL((kernel<HW, FM_GAUSSP>), grid, block, 0, st, __VA_ARGS__);       \
      break;                                                                              \
    default: hipLaunchKernelGGL((kernel<HW, FM_GAUSS>), grid, block, 0, st, __VA_ARGS__); break;   \
  }
#define SSSPY_FAST_LAUNCH2(kernel, A, ...)                  \
  do {                                                      \
    if (A) {                                                \
      SSSPY_FAST_LAUNCH_M(kernel, true, __VA_ARGS__)        \
    } else {                                                \
      SSSPY_FAST_LAUNCH_M(kernel, false, __VA_ARGS__)       \
    }                                                       \
  } while (0)

#if SSSPY_FAST_PART != 2
// `part` must hold the scratch of ilrma_api.hip's basis_part_bytes() (used only when items are split)
// loss_out: nullptr, or B zeroed doubles that receive the data term of the loss of the state at entry.
// K <= 16: basis_out == basis (in place); 16 < K <= 32: basis_out must be a separate (B,N,F,K) buffer
// (two k-tile items per bin group read the old basis) and the caller copies it back.
// power_in (only with W == nullptr and loss_out == nullptr): X holds |y|^2 as (B, N, F, T) f64.
int LAUNCHER(ilrma_fast_basis)(const void *X, const void *W, const double *basis, double *basis_out,
                               const double *act, int B, int F, int T, int K, int floor_kind,
                               double eps, double *part, int fmodel, double mparam, int me,
                               double *loss_out, void *loss_ws, int power_in, hipStream_t st,
                               long long loss_stride) {
  // loss_ws: LAUNCHER(ilrma_fast_loss_ws_bytes)() of scratch behind loss_out (the per-wave shares
  // are stored there and added up in a fixed order: no atomics)
  // loss_stride > 0 (round 5): loss_out is the caller's raw slot array instead, slot s of mixture b
  // at loss_out[s * loss_stride + b], zeroed by the caller and folded by it (ssspy_fold_scalar_slots)
  // once for many iterations: no memset and no fold launch here
  if (power_in && (W != nullptr || loss_out != nullptr))
    return fail(SSSPY_ERR_BADARG, "ilrma_fast_basis: power input excludes a filter and the loss");
  if (loss_out && !loss_ws) return fail(SSSPY_ERR_BADARG, "ilrma_fast_basis: loss without scratch");
  const int ktiles = K > 32 ? 4 : (K > 16 ? 2 : 1);
#ifdef SSSPY_KTILE_ITEMS
  const int item_tiles = ktiles;  // one item per (bin group, k tile)
#else
  const int item_tiles = ktiles == 2 ? 1 : ktiles;  // n_basis <= 32: both k tiles inside the item
#endif
  // (the wide variants hold one workgroup per CU)
  const TailPlan plan =
      make_tail_plan(B, ((F + 63) / 64) * item_tiles, (T + 15) / 16, ktiles >= 2 ? 256 : SLOTS,
                     ktiles >= 2 ? 256 : 1024);  // (basis_part_bytes(): 1024 records)
  const FastModel fm = make_fast_model(fmodel, mparam, me);
  dim3 grid(plan.full + plan.tail * plan.split), block(256);
  // loss slots per mixture: (bin group, chunk, wave) of the pass, then (bin group, block, wave) of
  // the fold of the split items
  const int maxsplit = plan.split > 1 ? plan.split : 1;
  const int nbx = N * 64 * 16 / 256;
  const int slots_pass = plan.groups * maxsplit * 4;
  const int nslots = slots_pass + plan.groups * nbx * 4;
  const bool with_loss = loss_out != nullptr && ktiles == 1 && fmodel != FM_T;
  const bool raw = with_loss && loss_stride > 0;
  double *loss_slots = with_loss ? (raw ? loss_out : (double *)loss_ws) : nullptr;
  const int slot_stride = raw ? (int)loss_stride : B;
  if (with_loss && !raw) {
    const int rc0 = scalar_slots_begin(loss_ws, B, nslots, st);
    if (rc0) return rc0;
  }
#define SSSPY_BASIS_LAUNCH(HW, M, L, KS_)                                                          \
  hipLaunchKernelGGL((k_basis_fast<HW, M, L, KS_>), grid, block, 0, st, (const c128 *)X,           \
                     (const c128 *)W, basis, basis_out, act, F, T, K, floor_kind, eps, plan, part, \
                     fm, loss_slots, slot_stride)
#define SSSPY_BASIS_LAUNCH_M(HW, L, KS_)                                  \
  switch (fmodel) {                                                       \
    case FM_T: SSSPY_BASIS_LAUNCH(HW, FM_T, false, KS_); break; /* no by-product for the t model */ \
    case FM_GGD: SSSPY_BASIS_LAUNCH(HW, FM_GGD, L, KS_); break;           \
    case FM_GAUSS1: SSSPY_BASIS_LAUNCH(HW, FM_GAUSS1, L, KS_); break;     \
    case FM_GAUSSP: SSSPY_BASIS_LAUNCH(HW, FM_GAUSSP, L, KS_); break;     \
    default: SSSPY_BASIS_LAUNCH(HW, FM_GAUSS, L, KS_); break;             \
  }
  if (power_in) {
    if (ktiles == 4) {
      SSSPY_BASIS_LAUNCH_M(IN_P, false, 16)
    } else if (ktiles == 2) {
      SSSPY_BASIS_LAUNCH_M(IN_P, false, 8)
    } else {
      SSSPY_BASIS_LAUNCH_M(IN_P, false, 4)
    }
  } else if (ktiles == 4) {  // the wide variants carry no loss by-product (register budget)
    if (W != nullptr) {
      SSSPY_BASIS_LAUNCH_M(true, false, 16)
    } else {
      SSSPY_BASIS_LAUNCH_M(false, false, 16)
    }
  } else if (ktiles == 2) {
    if (W != nullptr) {
      SSSPY_BASIS_LAUNCH_M(true, false, 8)
    } else {
      SSSPY_BASIS_LAUNCH_M(false, false, 8)
    }
  } else if (W != nullptr) {
    if (with_loss) {
      SSSPY_BASIS_LAUNCH_M(true, true, 4)
    } else {
      SSSPY_BASIS_LAUNCH_M(true, false, 4)
    }
  } else {
    if (with_loss) {
      SSSPY_BASIS_LAUNCH_M(false, true, 4)
    } else {
      SSSPY_BASIS_LAUNCH_M(false, false, 4)
    }
  }
#undef SSSPY_BASIS_LAUNCH_M
#undef SSSPY_BASIS_LAUNCH
  int rc = check_launch("k_basis_fast");
  if (rc) return rc;
  if (plan.tail > 0) {
    // split items' share of the loss by-product (never requested for the t model or the wide variant)
    const double loss_scale = (fmodel == FM_GGD ? 2.0 / fm.beta : 1.0) / (double)T;
    const int inner = ktiles / item_tiles;  // k tiles inside an item
    hipLaunchKernelGGL(k_basis_finalize, dim3(nbx, plan.tail, inner), block, 0, st, basis,
                       basis_out, part, F, K, plan, inner > 1 ? -inner : item_tiles, floor_kind, eps,
                       fm.expo, loss_slots, loss_scale, slot_stride, slots_pass);
    rc = check_launch("k_basis_finalize");
    if (rc) return rc;
  }
  return (with_loss && !raw) ? scalar_slots_fold(loss_ws, B, nslots, loss_out, 0, st) : rc;
}

// slots per mixture of the loss by-product of LAUNCHER(ilrma_fast_basis) (the raw form's array)
int LAUNCHER(ilrma_fast_basis_loss_slots)(int B, int F, int T) {
  const TailPlan plan = make_tail_plan(B, (F + 63) / 64, (T + 15) / 16, SLOTS, 1024);
  const int maxsplit = plan.split > 1 ? plan.split : 1;
  return plan.groups * maxsplit * 4 + plan.groups * (N * 64 * 16 / 256) * 4;
}

// scratch of the deterministic loss sums (both the by-product of the basis pass and the loss pass)
size_t LAUNCHER(ilrma_fast_loss_ws_bytes)(int B, int F) {
  const int groups = (F + 63) / 64;
  return scalar_slots_bytes(B, groups * 4 * (16 + N * 64 * 16 / 256));
}

#endif

#if SSSPY_FAST_PART != 1
int LAUNCHER(ilrma_fast_activation)(const void *X, const void *W, const double *basis,
                                    const double *act, double *part, int nchunks, int B, int F,
                                    int T, int K, int fmodel, double mparam, int power_in,
                                    hipStream_t st) {
  if (power_in && W != nullptr)
    return fail(SSSPY_ERR_BADARG, "ilrma_fast_activation: power input excludes a filter");
  const int ntiles = (F + 15) / 16;
  const int tiles_per_chunk = (ntiles + nchunks - 1) / nchunks;
  const FastModel fm = make_fast_model(fmodel, mparam, 0);
  const int ktiles = K > 32 ? 4 : (K > 16 ? 2 : 1);
#ifdef SSSPY_KTILE_ITEMS
  const int item_tiles = ktiles;
#else
  const int item_tiles = ktiles == 2 ? 1 : ktiles;  // n_basis <= 32: both k tiles inside the item
#endif
  dim3 grid((T + 63) / 64, nchunks * item_tiles, B), block(256);
#define SSSPY_ACT_LAUNCH(HW, M, KS_)                                                             \
  hipLaunchKernelGGL((k_activation_fast<HW, M, KS_>), grid, block, 0, st, (const c128 *)X,        \
                     (const c128 *)W, basis, act, part, F, T, K, tiles_per_chunk, nchunks, fm)
#define SSSPY_ACT_LAUNCH_M(HW, KS_)                        \
  switch (fmodel) {                                        \
    case FM_T: SSSPY_ACT_LAUNCH(HW, FM_T, KS_); break;     \
    case FM_GGD: SSSPY_ACT_LAUNCH(HW, FM_GGD, KS_); break; \
    case FM_GAUSS1: SSSPY_ACT_LAUNCH(HW, FM_GAUSS1, KS_); break; \
    case FM_GAUSSP: SSSPY_ACT_LAUNCH(HW, FM_GAUSSP, KS_); break; \
    default: SSSPY_ACT_LAUNCH(HW, FM_GAUSS, KS_); break;   \
  }
  if (power_in) {
    if (ktiles == 4) {
      SSSPY_ACT_LAUNCH_M(IN_P, 16)
    } else if (ktiles == 2) {
      SSSPY_ACT_LAUNCH_M(IN_P, 8)
    } else {
      SSSPY_ACT_LAUNCH_M(IN_P, 4)
    }
  } else if (ktiles == 4) {
    if (W != nullptr) {
      SSSPY_ACT_LAUNCH_M(true, 16)
    } else {
      SSSPY_ACT_LAUNCH_M(false, 16)
    }
  } else if (ktiles == 2) {
    if (W != nullptr) {
      SSSPY_ACT_LAUNCH_M(true, 8)
    } else {
      SSSPY_ACT_LAUNCH_M(false, 8)
    }
  } else if (W != nullptr) {
    SSSPY_ACT_LAUNCH_M(true, 4)
  } else {
    SSSPY_ACT_LAUNCH_M(false, 4)
  }
#undef SSSPY_ACT_LAUNCH_M
#undef SSSPY_ACT_LAUNCH
  return check_launch("k_activation_fast");
}

// out[b] = the data term; loss_ws: LAUNCHER(ilrma_fast_loss_ws_bytes)() of scratch
int LAUNCHER(ilrma_fast_loss)(const void *X, const void *W, const double *basis, const double *act,
                              double *out, void *loss_ws, int B, int F, int T, int K, int fmodel,
                              double mparam, hipStream_t st) {
  const TailPlan plan = make_tail_plan(B, (F + 63) / 64, (T + 15) / 16);
  const FastModel fm = make_fast_model(fmodel, mparam, 0);
  dim3 grid(plan.full + plan.tail * plan.split), block(256);
  const int nslots = plan.groups * (plan.split > 1 ? plan.split : 1) * 4;
  int rc = scalar_slots_begin(loss_ws, B, nslots, st);
  if (rc) return rc;
  double *slots = (double *)loss_ws;
  SSSPY_FAST_LAUNCH2(k_loss_fast, W != nullptr, (const c128 *)X, (const c128 *)W,
                     basis, act, slots, F, T, K, plan, fm, B);
  rc = check_launch("k_loss_fast");
  return rc ? rc : scalar_slots_fold(loss_ws, B, nslots, out, 0, st);
}

// `upart` must hold u_part_bytes() of ilrma_api.hip (used only when some items are split);
// W is read by the t and GGD models only; the floor is GGD's (on |y|^(2 - beta))
// split_out (optional): the caller folds the split items' records itself (k_ip1_small): no
// k_wcov_fold here; *split_out = chunks per item, *rbins_out = bins per record -- only when EVERY
// item is split (a handful of mixtures); otherwise *split_out = 0 and U is complete on return.
int LAUNCHER(ilrma_fast_wcov)(const void *X, const void *W, const double *basis, const double *act,
                              void *U, int B, int F, int T, int K, void *upart, int fmodel,
                              double mparam, int floor_kind, double floor_eps, hipStream_t st,
                              int *split_out = nullptr, int *rbins_out = nullptr) {
  if (split_out) *split_out = 0;
  if (rbins_out) *rbins_out = WC_BINS;
  const TailPlan plan =
      make_tail_plan(B, (F + WC_BINS - 1) / WC_BINS, (T + 15) / 16, K > 32 ? 256 : SLOTS,
                     K > 32 ? 256 : 1024);  // (u_part_bytes(): 1024 records)
  const FastModel fm = make_fast_model(fmodel, mparam, 0, floor_kind, floor_eps);
  dim3 grid(plan.full + plan.tail * plan.split), block(256);
#define SSSPY_WCOV_LAUNCH(M, KS_)                                                                 \
  hipLaunchKernelGGL((k_wcov_fast<M, KS_>), grid, block, 0, st, (const c128 *)X, (const c128 *)W,  \
                     basis, act, (c128 *)U, F, T, K, plan, (c128 *)upart, fm)
#define SSSPY_WCOV_LAUNCH_M(KS_)                       \
  switch (fmodel) {                                    \
    case FM_T: SSSPY_WCOV_LAUNCH(FM_T, KS_); break;    \
    case FM_GGD: SSSPY_WCOV_LAUNCH(FM_GGD, KS_); break; \
    case FM_GAUSS1: SSSPY_WCOV_LAUNCH(FM_GAUSS1, KS_); break; \
    case FM_GAUSSP: SSSPY_WCOV_LAUNCH(FM_GAUSSP, KS_); break; \
    default: SSSPY_WCOV_LAUNCH(FM_GAUSS, KS_); break;  \
  }
  if (K > 32) {  // (one workgroup per CU: the 64-row GEMM1 operands take the whole register file)
    SSSPY_WCOV_LAUNCH_M(16)
  } else if (K > 16) {
    SSSPY_WCOV_LAUNCH_M(8)
  } else {
    SSSPY_WCOV_LAUNCH_M(4)
  }
#undef SSSPY_WCOV_LAUNCH_M
#undef SSSPY_WCOV_LAUNCH
  int rc = check_launch("k_wcov_fast");
  if (rc || plan.tail == 0) return rc;
  if (split_out && plan.full == 0) {
    *split_out = plan.split;
    return rc;
  }
  hipLaunchKernelGGL(k_wcov_fold, dim3((WC_BINS * N * N * N + 255) / 256, plan.tail), block, 0, st,
                     (c128 *)U, (const c128 *)upart, F, plan);
  return check_launch("k_wcov_fold");
}

// Frame-weighted covariance U (B,F,N,N,N) from weight (B,N,T).  Returns -1 without launching when
// the batch is too small for unsplit items to fill the chip (the caller then takes the generic
// kernel, whose waves split the frames), otherwise the status of the launch.
int LAUNCHER(ilrma_fast_wcov_frame)(const void *X, const double *weight, void *U, int B, int F,
                                    int T, hipStream_t st) {
  static const bool disabled = std::getenv("SSSPY_AMD_NO_FAST") != nullptr;
  const int groups = (F + WC_BINS - 1) / WC_BINS;
  const long long items = (long long)B * groups;
  if (disabled || items < SLOTS || (long long)F * T * 16 >= (1ll << 32)) return -1;
  hipLaunchKernelGGL(k_wcov_frame_fast, dim3((unsigned)items), dim3(256), 0, st, (const c128 *)X,
                     weight, (c128 *)U, F, T, groups);
  return check_launch("k_wcov_frame_fast");
}

#endif

#undef SSSPY_FAST_LAUNCH2
#undef SSSPY_FAST_LAUNCH_M

}  // namespace ssspy
