// Tile plumbing shared by the throughput ("fast") pass kernels of ILRMA and FastMNMF: staging of the
// activation tile in LDS, the two ways a wave fetches its 16-bin x 16-frame x tile, GEMM1 from the
// staged tile, the Newton reciprocal.  NS = sources (rows of the NMF pair), NC = channels of X.
// Lane layout of a wave throughout: lane = 16 q + c, c = bin inside the wave's 16-bin tile, q = frame
// sub-group; register r of lane (c, q) holds frame j0 + q + 4 r.
#pragma once

#include "common.hpp"

namespace ssspy {
namespace fast {

constexpr int VROW = 18;  // doubles per staged row (16 + 2 pad: 144-byte stride)

__device__ __forceinline__ double rcp_nr(double x) {
  // v_rcp_f64 + 2 Newton steps: exact to ~1 ulp (benchmarks/micro/rcp_precision.hip)
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  return r;
}

// sqrt(x) for x >= 0 as x * rsq(x): v_rsq_f64 + 2 Newton steps (~1 ulp, the recipe of the fused ISS
// sweep's coefficient lane) -- 11 instructions against the ~25 of the correctly rounded sqrt();
// zero, Inf, NaN and arguments whose reciprocal square root leaves the range keep sqrt()
__device__ __forceinline__ double sqrt_nr(double x) {
  double rs = __builtin_amdgcn_rsq(x);
  double h = 0.5 * x * rs;
  double e = fma(-h, rs, 0.5);
  rs = fma(rs, e, rs);
  h = 0.5 * x * rs;
  e = fma(-h, rs, 0.5);
  rs = fma(rs, e, rs);
  double r = x * rs;
  if (__builtin_expect(!(x > 1e-290 && x < 1e290), 0)) r = sqrt(x);
  return r;
}

__device__ __forceinline__ int tile_pi(int rho) { return 4 * (rho & 3) + (rho >> 2); }

// ---- stage the activation tile V[b, n, 0:KR, j0:j0+16] of every source into LDS rows of VROW
// doubles (zero beyond K rows / T frames).  256 threads, NS*KR rows * 8 double2 chunks; KR = 16
// (n_basis <= 16) or 32.
template <int NS, int KR = 16>
struct VStage {
  double2 v[(NS * KR * 8 + 255) / 256];
};

template <int NS, int KR = 16>
__device__ __forceinline__ void vstage_load(VStage<NS, KR> &st, const double *__restrict__ act_b,
                                            int K, int T, int j0) {
#pragma unroll
  for (int u = 0; u < (NS * KR * 8 + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int row = idx >> 3, chunk = idx & 7;  // row = n*KR + k
    const int n = row / KR, k = row % KR;
    const int j = j0 + 2 * chunk;
    double2 val = make_double2(0.0, 0.0);
#ifdef SSSPY_ASSUME_FULL
    {
      const double2_a8 v = *reinterpret_cast<const double2_a8 *>(act_b + ((long long)n * K + k) * T + j);
      val = make_double2(v.x, v.y);
    }
#else
    if (idx < NS * KR * 8 && k < K) val = load_pair_in_row(act_b + ((long long)n * K + k) * T, j, T);
#endif
    st.v[u] = val;
  }
}

template <int NS, int KR = 16>
__device__ __forceinline__ void vstage_store(const VStage<NS, KR> &st, double *buf) {
#pragma unroll
  for (int u = 0; u < (NS * KR * 8 + 255) / 256; ++u) {
    const int idx = threadIdx.x + 256 * u;
    const int row = idx >> 3, chunk = idx & 7;
    if (idx < NS * KR * 8) {  // frame f of the tile lives in slot tile_pi(f)
      buf[row * VROW + tile_pi(2 * chunk)] = st.v[u].x;
      buf[row * VROW + tile_pi(2 * chunk + 1)] = st.v[u].y;
    }
  }
}

template <int NC>
struct XTile {
  c128 x[NC][4];
};

// Order pin for a register-neutral prefetch: returns v unchanged, but through an opaque asm that
// reads every power -- so all of them (the last uses of the x tile) are computed BEFORE the asm and
// whatever is addressed with the result (the re-load of the x registers) comes AFTER it.  Pure
// arithmetic carries no ordering against loads in LLVM IR: without the pin the IR-level sinking moves
// the |y|^2 arithmetic below the loads and both tiles are live at once (240 bytes of spills measured).
template <int NS>
__device__ __forceinline__ int pin_after_powers(const double (&pw)[NS][4], int v) {
  static_assert(NS >= 1 && NS <= 4, "pin_after_powers: 1..4 sources");
  if constexpr (NS == 4)
    asm volatile("" : "+v"(v) : "v"(pw[0][0]), "v"(pw[0][1]), "v"(pw[0][2]), "v"(pw[0][3]),
                 "v"(pw[1][0]), "v"(pw[1][1]), "v"(pw[1][2]), "v"(pw[1][3]), "v"(pw[2][0]),
                 "v"(pw[2][1]), "v"(pw[2][2]), "v"(pw[2][3]), "v"(pw[3][0]), "v"(pw[3][1]),
                 "v"(pw[3][2]), "v"(pw[3][3]));
  else if constexpr (NS == 3)
    asm volatile("" : "+v"(v) : "v"(pw[0][0]), "v"(pw[0][1]), "v"(pw[0][2]), "v"(pw[0][3]),
                 "v"(pw[1][0]), "v"(pw[1][1]), "v"(pw[1][2]), "v"(pw[1][3]), "v"(pw[2][0]),
                 "v"(pw[2][1]), "v"(pw[2][2]), "v"(pw[2][3]));
  else if constexpr (NS == 2)
    asm volatile("" : "+v"(v) : "v"(pw[0][0]), "v"(pw[0][1]), "v"(pw[0][2]), "v"(pw[0][3]),
                 "v"(pw[1][0]), "v"(pw[1][1]), "v"(pw[1][2]), "v"(pw[1][3]));
  else
    asm volatile("" : "+v"(v) : "v"(pw[0][0]), "v"(pw[0][1]), "v"(pw[0][2]), "v"(pw[0][3]));
  return v;
}

// bin-major x tile: lane (c, q) reads frames j0+q+4r of bin `bin`, so one load instruction
// (fixed r) takes 64 contiguous bytes per bin from the 4 q-lanes: 16 half cache lines instead of the
// 32 quarter lines of a "4 consecutive frames per lane" split (TCP tag-conflict stalls, profiles/)
template <int NC>
__device__ __forceinline__ void xtile_load_binmajor(XTile<NC> &xt, const c128 *__restrict__ Xb,
                                                    int F, int T, int bin, int j0, int q) {
  const int j = j0 + q;
#pragma unroll
  for (int m = 0; m < NC; ++m) {
    const c128 *row = Xb + ((long long)m * F + bin) * T;
#pragma unroll
    for (int r = 0; r < 4; ++r) xt.x[m][r] = row[min(j + 4 * r, T - 1)];
  }
}

// ---- the same tiles through buffer descriptors (one per channel row block, base in SGPRs): the
// per-lane address state is ONE 32-bit offset per tile -- the four frames (or bin rows) of a lane go
// into the instruction's immediate or a scalar offset -- where the flat form costs ~5 VALU
// instructions of 64-bit address arithmetic per load (80 per tile in kernels that are issue-bound).
// Out-of-range frames / bins are not clamped: they read a neighbouring row of the same channel
// (finite data) or, past the end of the channel, the zeros the bounds check returns; every consumer
// masks them.  Needs F * T * 16 < 2^32 per channel (the launchers check).
template <int NC>
struct XSrc {
  __amdgpu_buffer_rsrc_t ch[NC];
};

template <int NC>
__device__ __forceinline__ XSrc<NC> make_xsrc(const c128 *Xb, int F, int T) {
  XSrc<NC> s;
#pragma unroll
  for (int m = 0; m < NC; ++m)
    s.ch[m] = make_rsrc(Xb + (long long)m * F * T, (unsigned)F * (unsigned)T * 16u);
  return s;
}

// |y|^2 instead of y (the grouped passes of a wide mixture read the power of the separated
// spectrogram, (B, N, F, T) f64: half the bytes): per-channel descriptors over 8-byte elements and
// the two tile fetches on them.
template <int NC>
struct PTile {
  double p[NC][4];
};
template <int NC>
__device__ __forceinline__ XSrc<NC> make_psrc(const double *Pb, int F, int T) {
  XSrc<NC> s;
#pragma unroll
  for (int m = 0; m < NC; ++m)
    s.ch[m] = make_rsrc(Pb + (long long)m * F * T, (unsigned)F * (unsigned)T * 8u);
  return s;
}
__device__ __forceinline__ double f64_from(u32x2_t v) {
  return __hiloint2double((int)v[1], (int)v[0]);
}
// bin-major: frames j0 + q + 4r of bin `bin`
template <int NC>
__device__ __forceinline__ void ptile_load_binmajor(PTile<NC> &pt, const XSrc<NC> &src, int T,
                                                    int bin, int j0, int q) {
  const unsigned voff = ((unsigned)bin * (unsigned)T + (unsigned)(j0 + q)) * 8u;
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      pt.p[m][r] = f64_from(__builtin_amdgcn_raw_buffer_load_b64(src.ch[m], voff + 32u * r, 0, 0));
}
// frame-major: frame jc of bins i0 + q + 4r
template <int NC>
__device__ __forceinline__ void ptile_load_framemajor(PTile<NC> &pt, const XSrc<NC> &src, int T,
                                                      int i0, int jc, int q) {
  const unsigned voff = ((unsigned)(i0 + q) * (unsigned)T + (unsigned)jc) * 8u;
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      pt.p[m][r] = f64_from(
          __builtin_amdgcn_raw_buffer_load_b64(src.ch[m], voff, 4u * r * (unsigned)T * 8u, 0));
}

__device__ __forceinline__ c128 c128_from(u32x4_t v) {
  return cmake(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
}

// bin-major: frames j0 + q + 4r of bin `bin`
template <int NC>
__device__ __forceinline__ void xtile_load_binmajor(XTile<NC> &xt, const XSrc<NC> &src, int T,
                                                    int bin, int j0, int q) {
  const unsigned voff = ((unsigned)bin * (unsigned)T + (unsigned)(j0 + q)) * 16u;
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      xt.x[m][r] = c128_from(__builtin_amdgcn_raw_buffer_load_b128(src.ch[m], voff + 64u * r, 0, 0));
}

// frame-major: frame jc of bins i0 + q + 4r
template <int NC>
__device__ __forceinline__ void xtile_load_framemajor(XTile<NC> &xt, const XSrc<NC> &src, int T,
                                                      int i0, int jc, int q) {
  const unsigned voff = ((unsigned)(i0 + q) * (unsigned)T + (unsigned)jc) * 16u;
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      xt.x[m][r] = c128_from(
          __builtin_amdgcn_raw_buffer_load_b128(src.ch[m], voff, 4u * r * (unsigned)T * 16u, 0));
}

// The same tile, fetched with coalesced addresses and transposed through a wave-private LDS
// patch: a load instruction takes 4 bin rows x 256 contiguous bytes (lane = frame), the patch
// turns (lane = frame, register = bin) into (lane = bin, register = frame).  Two channels per
// pass so the patch stays at 8.5 KB per wave; rows are 17 slots apart, which makes both the
// frame-major writes and the bin-major reads bank-conflict free.  The patch is wave-private, so no
// workgroup barrier is needed -- but the exchange is between lanes, which the per-thread memory
// model does not order: every write and read phase is fenced explicitly (see below).  (Measured on
// the covariance kernel, whose 2 waves per bin tile made the texture addresser the limiter:
// 1.32 -> 1.10 ms.)
constexpr int XPATCH = 2 * 16 * 17;  // c128 slots per wave

template <int NC>
__device__ __forceinline__ void xtile_transpose(XTile<NC> &xt, int c, int q, c128 *patch);

template <int NC>
__device__ __forceinline__ void xtile_load_transposed(XTile<NC> &xt, const c128 *__restrict__ Xb,
                                                      int F, int T, int i0, int j0, int c, int q,
                                                      c128 *patch) {
  const int jf = min(j0 + c, T - 1);
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      xt.x[m][rr] = Xb[((long long)m * F + min(i0 + 4 * rr + q, F - 1)) * T + jf];
  xtile_transpose<NC>(xt, c, q, patch);
}

// buffer-descriptor form of the coalesced fetch (frame j0 + c of bins i0 + q + 4 rr)
template <int NC>
__device__ __forceinline__ void xtile_load_transposed(XTile<NC> &xt, const XSrc<NC> &src, int T,
                                                      int i0, int j0, int c, int q, c128 *patch) {
  const unsigned voff = ((unsigned)(i0 + q) * (unsigned)T + (unsigned)min(j0 + c, T - 1)) * 16u;
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      xt.x[m][rr] = c128_from(
          __builtin_amdgcn_raw_buffer_load_b128(src.ch[m], voff, 4u * rr * (unsigned)T * 16u, 0));
  xtile_transpose<NC>(xt, c, q, patch);
}

// The tile shared by the NG waves that work on the same 16 bins (one per source group): wave g
// fetches channels g*NC/NG ... only, all of them write one patch (NC * 16 rows of 17 slots) and
// every wave reads the whole tile back transposed.  Two workgroup barriers per tile -- the patch of
// the previous tile has been read by everyone / the writes of this one have landed -- so EVERY wave
// of the workgroup must walk the same tiles.  Halves (NG = 2) the global loads and the LDS writes
// of the private form; the waves of a bin tile no longer fetch the same lines twice.
template <int NC, int NG>
__device__ __forceinline__ void xtile_load_shared(XTile<NC> &xt, const XSrc<NC> &src, int T, int i0,
                                                  int j0, int c, int q, int g, c128 *patch) {
  static_assert(NC % NG == 0, "channels split evenly over the source groups");
  constexpr int PER = NC / NG;
  const unsigned voff = ((unsigned)(i0 + q) * (unsigned)T + (unsigned)min(j0 + c, T - 1)) * 16u;
  u32x4_t ld[PER][4];
#pragma unroll
  for (int gg = 0; gg < NG; ++gg)
    if (gg == g) {  // wave-uniform: the descriptor stays in scalar registers
#pragma unroll
      for (int mm = 0; mm < PER; ++mm)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          ld[mm][rr] = __builtin_amdgcn_raw_buffer_load_b128(src.ch[gg * PER + mm], voff,
                                                             4u * rr * (unsigned)T * 16u, 0);
    }
  __syncthreads();  // the previous tile has been read out of the patch by every wave
#pragma unroll
  for (int mm = 0; mm < PER; ++mm)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      patch[((g * PER + mm) * 16 + 4 * rr + q) * 17 + c] = c128_from(ld[mm][rr]);
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NC; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) xt.x[m][r] = patch[(m * 16 + c) * 17 + q + 4 * r];
}

// (lane = frame, register = bin) -> (lane = bin, register = frame) through the wave's LDS patch
template <int NC>
__device__ __forceinline__ void xtile_transpose(XTile<NC> &xt, int c, int q, c128 *patch) {
#pragma unroll
  for (int m0 = 0; m0 < NC; m0 += 2) {
    // The patch is reused by every pass and every tile, and the exchange is between LANES: nothing
    // in the per-thread memory model orders this pass's writes after the previous pass's reads
    // (measured: without the wait a barrier-free walk returned wrong tiles for N >= 3).  Drain the
    // wave's outstanding LDS reads and pin the order for the compiler.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
      if (m0 + mm < NC) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) patch[(mm * 16 + 4 * rr + q) * 17 + c] = xt.x[m0 + mm][rr];
      }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // the writes of all lanes have landed
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
      if (m0 + mm < NC) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xt.x[m0 + mm][r] = patch[(mm * 16 + c) * 17 + q + 4 * r];
      }
  }
}

// GEMM1 of the bin-major tile from the staged V: R[bin c, frame j0+q+4r] in register r
// (D row q+4r reads slot tile_pi(q+4r) = 4q+r, which holds frame tile_pi(4q+r) = q+4r)
// ksteps = ceil(K / 4): k-slabs beyond n_basis are zero on both sides and are skipped
template <int KS>
__device__ __forceinline__ double4_t rt_from_lds(const double *vs_n, const double (&tb)[KS], int c,
                                                 int q, int ksteps) {
  double4_t R = {0.0, 0.0, 0.0, 0.0};
  const int col = tile_pi(c);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#ifdef SSSPY_ASSUME_FULL
    R = mfma_f64(vs_n[(4 * ks + q) * VROW + col], tb[ks], R);
#else
    if (ks < ksteps) R = mfma_f64(vs_n[(4 * ks + q) * VROW + col], tb[ks], R);
#endif
  return R;
}

// The same with the B operand (the basis rows of the wave's 16 bins) read from LDS as well:
// tl_n[k * 16 + c] = T[n, bin c, k], zero beyond n_basis -- for kernels that cannot afford the 32
// VGPRs of a hoisted operand (FastMNMF keeps every source's GEMM1 output live at once)
__device__ __forceinline__ double4_t rt_from_lds(const double *vs_n, const double *tl_n, int c,
                                                 int q, int ksteps) {
  double4_t R = {0.0, 0.0, 0.0, 0.0};
  const int col = tile_pi(c);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    if (ks < ksteps) R = mfma_f64(vs_n[(4 * ks + q) * VROW + col], tl_n[(4 * ks + q) * 16 + c], R);
  return R;
}

// stage T[n, i0 + 0..15, 0..K) of every source into a wave-private LDS block tl[n][k][bin]
template <int NS>
__device__ __forceinline__ void stage_basis_rows(double *tl, const double *__restrict__ basis_b,
                                                 int F, int K, int i0, int lane) {
  for (int e = lane; e < NS * 256; e += 64) {
    const int k = e & 15, bl = (e >> 4) & 15, n = e >> 8;
    const int bi = min(i0 + bl, F - 1);
    tl[(n * 16 + k) * 16 + bl] = k < K ? basis_b[((long long)n * F + bi) * K + k] : 0.0;
  }
}

}  // namespace fast
}  // namespace ssspy
