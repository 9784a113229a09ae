// IPA source step for 5..8 sources with a BIN ON 8 LANES (round 5): lane r of an 8-lane group holds
// row r of every N x N operand (herm_rows8.hpp), the source count and the source index are run-time
// arguments (padded to 8; rows / columns S and >= N of the reduced problem are zero and the Jacobi
// rotations that would touch them see a zero off-diagonal).  The lane-per-bin kernel of
// ipa_kernels.hip is one function per (N, source, mode) -- 72 of them above 4 sources, each with
// unrolled 7 x 7 / 8 x 8 Jacobi bodies on 1 000-2 000 spilled VGPRs.
//
// Same algorithm, statement by statement, as ipa_source_step of ipa_kernels.hip (see there for the reference lines:
// ssspy/bss/_update_spatial_model.py:398-513, :611-645, ssspy/linalg/lqpqm.py:13-352):
//   a_m, b_m from to_psd(U_m), m != S (Cholesky test of the floor, eigen route when it acts);
//   U_S^-1 (doubly floored) -> C, d, z;  H, v;  LQPQM2 (Hermitian Jacobi, Cardano start, Newton);
//   q, q~, p = U_S^-1 q~ / floor(sqrt(q~^H U_S^-1 q~));  G_S;  chained: V_m <- G_S V_m G_S^H, G <- G_S G.
// Differences in rounding only: C x = d goes through the Cholesky inverse of C (Hermitian positive
// definite as a principal block of U_S^-1) instead of an LU, column sums run over lanes.
#include <cstdlib>

#include "herm_rows8.hpp"
#include "ipa_common.hpp"
#include "ssspy_amd.h"

namespace ssspy {

using namespace rows8;

namespace {

constexpr int BINS = 32;  // bins per 256-thread block

__device__ __forceinline__ double seld(const double (&a)[8], int i) {
  double r = a[0];
#pragma unroll
  for (int c = 1; c < 8; ++c) r = (i == c) ? a[c] : r;
  return r;
}
__device__ __forceinline__ void putd(double (&a)[8], int i, double v) {
#pragma unroll
  for (int c = 0; c < 8; ++c) a[c] = (i == c) ? v : a[c];
}

// row r of the Hermitian part of the N x N matrix at src, zero-padded
__device__ __forceinline__ void load_herm(const c128 *__restrict__ src, int r, int N, c128 (&row)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    c128 v = cmake(0.0, 0.0);
    if (r < N && c < N) {
      const c128 u = src[r * N + c], l = src[c * N + r];
      v = cmake(0.5 * (u.x + l.x), 0.5 * (u.y - l.y));
    }
    row[c] = v;
  }
}

// lam_min(A) > shift, by the pivots of the Cholesky factorisation of A - shift I (padding: unit diagonal).
// First Gershgorin's discs: lam_min >= min_r (a_rr - sum_{c != r} |a_rc|); the statistics of a
// spectrogram that is already mostly separated are diagonally dominant, and when every matrix of
// the wave passes, the eight-step factorisation (a chain of dependent pivots) is skipped.
__device__ __forceinline__ bool shifted_pd(const c128 (&a)[8], int r, int N, double shift) {
  {
    double offsum = 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) offsum += sqrt(cabs2(a[c]));
    const double arr = sel(a, r).x;
    const bool clear = r >= N || (arr - (offsum - fabs(arr)) > shift);
    if (__all(clear)) return true;
  }
  c128 t[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) t[c] = a[c];
  put(t, r, r < N ? cmake(sel(a, r).x - shift, 0.0) : cmake(1.0, 0.0));
  return chol_upper(t, r);
}

// Inverse of the Hermitian positive definite matrix with row r in `a` (padding: unit diagonal set by
// the caller): rows of A^-1 = V V^H, V = U^-1.  False when not positive definite.
__device__ __forceinline__ bool chol_inverse_rows(c128 (&a)[8], c128 *X, int r, c128 (&inv)[8]) {
  const bool ok = chol_upper(a, r);
  wsync();
  store_row(X, r, a);
  wsync();
  c128 col[8];
  trtri_col(X, r, col);
  wsync();
#pragma unroll
  for (int k = 0; k < 8; ++k) X[k * LD + r] = col[k];
  wsync();
  c128 vrow[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) vrow[c] = X[r * LD + c];
  mul_rows_adj(vrow, X, inv);  // sum_k V[r][k] conj(V[c][k])
  wsync();
  return ok;
}

// Round 6: the whole sweep in one launch -- a workgroup walks the N source steps of its 32 bins of
// ONE mixture and, at every Newton vote, meets the mixture's other workgroups (vote word + arrival
// counter per (mixture, step) in newton_ws, see k_ipa_sweep_fused in ipa_kernels.hip) instead of
// ending the kernel: 4 launches per source step before (vote memset, probe, step count, apply --
// the apply repeating the probe's work).
enum { ROWS_FUSED_FIXED = 4 };
__device__ int g_ipa_rows_barrier_timeouts;

// One source step (S at compile time: as a loop variable it costs the register arrays indexed by
// it their registers -- 86 -> 615 spilled VGPRs -- so the sweep is eight instances in a row).
template <int MODE, int S>
__device__ __forceinline__ void ipa_rows_step(const c128 *Ub, c128 *__restrict__ G, c128 *Vchain,
                                              c128 *X, unsigned long long *vote_word,
                                              long long bin, bool live, int r, int lane, int N,
                                              int normalization, int max_iter, int floor_kind,
                                              double eps, int *info, unsigned long long *newton_ws,
                                              int B, int *not_converged) {
  const double f0 = floor_of_zero(floor_kind, eps);
  const int chain_first = S == 0 ? 1 : 0;
  unsigned long long *word =
      MODE == NEWTON_FUSED ? newton_ws + (long long)blockIdx.y * N + S : nullptr;
  const bool rest = r < N && r != S;

  // ---- a_m = Re to_psd(U_m)[S][S], b_m = to_psd(U_m)[S][m], m != S (every lane keeps all of them)
  double a[8];
  c128 b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    a[c] = 1.0;
    b[c] = cmake(0.0, 0.0);
  }
#pragma unroll 1
  for (int m = 0; m < N; ++m) {
    if (m == S) continue;
    c128 row[8];
    load_herm(Ub + m * N * N, r, N, row);
    double ass = shfl8(sel(row, S).x, S);
    c128 bsm = shfl8(sel(row, m), S);
    bool idle = floor_kind != SSSPY_FLOOR_MAX;
    if (floor_kind == SSSPY_FLOOR_ADD) ass += eps;
    if (floor_kind == SSSPY_FLOOR_MAX) idle = shifted_pd(row, r, N, eps);
    if (!idle) {  // the floor may act: the literal route (whole 8-lane groups take it)
      double lam = sel(row, r).x;
      c128 p[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) p[c] = cmake(c == r ? 1.0 : 0.0, 0.0);
      jacobi<true>(row, lam, p, r);
      const double lf = apply_floor(lam, floor_kind, eps);
      ass = 0.0;
      bsm = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double lk = shfl8(lf, k);
        const c128 ps = shfl8(p[k], S), pm = shfl8(p[k], m);
        if (k < N) {
          ass = fma(lk, cabs2(ps), ass);
          const c128 t = cmulc(ps, pm);
          bsm.x = fma(lk, t.x, bsm.x);
          bsm.y = fma(lk, t.y, bsm.y);
        }
      }
    }
    putd(a, m, ass);
    put(b, m, bsm);
  }

  // ---- U_S: its Hermitian part `us`; the doubly floored inverse `uinv`
  c128 us[8], uinv[8], pl[8];  // pl: eigenvectors of the literal route
  double lamf = 1.0;           // floored eigenvalue r of the literal route
  load_herm(Ub + S * N * N, r, N, us);
  bool literal = floor_kind == SSSPY_FLOOR_MAX && !shifted_pd(us, r, N, eps);
  if (!literal) {
    c128 t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) t[c] = us[c];
    const double shift = floor_kind == SSSPY_FLOOR_ADD ? 2.0 * eps : 0.0;
    put(t, r, r < N ? cmake(sel(us, r).x + shift, 0.0) : cmake(1.0, 0.0));
    literal = !chol_inverse_rows(t, X, r, uinv);  // (not positive definite: the eigen route copes)
  }
  if (literal) {
    c128 t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      t[c] = us[c];
      pl[c] = cmake(c == r ? 1.0 : 0.0, 0.0);
    }
    double lam = sel(us, r).x;
    jacobi<true>(t, lam, pl, r);
    lamf = apply_floor(lam, floor_kind, eps);
    rebuild(pl, r < N ? 1.0 / apply_floor(lamf, floor_kind, eps) : 0.0, X, r, uinv);
    wsync();
  }

  // ---- C = conj(Uinv)[rest][rest], d = conj(Uinv)[rest][S];  x = C^-1 d;  z = Uinv[S][S] - d^H x
  c128 cm[8];  // row r of C embedded in 8 x 8 (zero row / column at S and beyond N)
#pragma unroll
  for (int c = 0; c < 8; ++c) cm[c] = (rest && c < N && c != S) ? cconj(uinv[c]) : cmake(0.0, 0.0);
  const c128 d = rest ? cconj(sel(uinv, S)) : cmake(0.0, 0.0);
  c128 x;
  {
    c128 t[8], cinv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) t[c] = cm[c];
    if (!rest) put(t, r, cmake(1.0, 0.0));
    const bool ok = chol_inverse_rows(t, X, r, cinv);
    if (!ok && info && live && r == 0) atomicAdd(info, 1);
    x = cmake(0.0, 0.0);
#pragma unroll
    for (int c = 0; c < 8; ++c) cfma(x, cinv[c], shfl8(d, c));
    if (!rest) x = cmake(0.0, 0.0);
  }
  const double dCd = sum8(d.x * x.x + d.y * x.y);
  double z = shfl8(sel(uinv, S).x, S) - dCd;
  const double as_r = sqrt(seld(a, r));
  c128 h[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const double sc = 1.0 / (as_r * sqrt(a[c]));
    h[c] = cscale(cm[c], sc);  // (zero where cm is)
  }
  const double tr = sum8(rest ? sel(h, r).x : 0.0);
  const c128 br = sel(b, r);
  // v = -b / a_sqrt - a_sqrt * C^-1 d
  c128 v = rest ? cmake(-br.x / as_r - as_r * x.x, -br.y / as_r - as_r * x.y) : cmake(0.0, 0.0);
  if (normalization) {
    const double it = 1.0 / tr;
#pragma unroll
    for (int c = 0; c < 8; ++c) h[c] = cscale(h[c], it);
    z *= it;
  }
  // hermitize H (its two triangles come from different lanes)
  wsync();
  store_row(X, r, h);
  wsync();
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const c128 t = X[c * LD + r];
    h[c] = cmake(0.5 * (h[c].x + t.x), 0.5 * (h[c].y - t.y));
  }
  wsync();

  // ---- LQPQM2: H = sigma diag(phi) sigma^H on the rest indices
  c128 sg[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) sg[c] = cmake(c == r ? 1.0 : 0.0, 0.0);
  double phi_r = sel(h, r).x;
  jacobi<true>(h, phi_r, sg, r);
  double phi[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) phi[l] = shfl8(phi_r, l);
  const double vnorm2 = sum8(cabs2(v));
  const bool is_singular = sqrt(vnorm2) < f0;
  c128 qc = cmake(0.0, 0.0);  // component r of the LQPQM solution
  // The Newton iteration of a regular problem (solve_equation, lqpqm.py:112-200; the same
  // arithmetic in every lane of the group): `steps` steps from the Cardano start value.  solve =
  // false: the convergence bits of the states before steps 0 .. `steps`; true: component r of the
  // solution.  A fused sweep runs it twice, around the mixture's vote -- nothing of it stays live
  // across the wait (keeping vt / phi~ / |v~|^2 there cost 600 spilled VGPRs).
  unsigned long long bits = 0ull;
  auto newton = [&](int steps, bool solve) {
    c128 vt[8];  // sigma^H v (column sums over the lanes)
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const c128 t = cmulc(v, sg[l]);  // v_r conj(sigma_rl)
      vt[l] = cmake(sum8(t.x), sum8(t.y));
    }
    double ph[8], w2[8];
    double pmax = 0.0, v2max = 0.0;
    bool first = true;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const bool valid = l < N && l != S;
      const bool keep = valid && (phi[l] * cabs2(vt[l]) >= f0);
      ph[l] = keep ? phi[l] : 0.0;
      w2[l] = keep ? cabs2(vt[l]) : 0.0;
      if (valid && (first || ph[l] > pmax)) {
        pmax = ph[l];
        v2max = w2[l];
        first = false;
      }
    }
    const double pm = apply_floor(pmax, floor_kind, eps);
    const double inv = 1.0 / pm;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      ph[l] *= inv;
      w2[l] *= inv * inv;
    }
    const double zn = z * inv;
    const double A = -(v2max * inv * inv + 2.0 + zn), Bc = 1.0 + 2.0 * zn, Cc = -zn;
    double lamb = largest_cubic_root(A, Bc, Cc);
    if (!(lamb > 1.0)) lamb = 1.0 + f0;
    lamb = fmax(lamb, zn);
    for (int it = 0; it <= steps; ++it) {
      double s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        if (l < N && l != S) {
          const double dl = lamb - ph[l];
          s2 += ph[l] * w2[l] / (dl * dl);
          s3 += ph[l] * ph[l] * w2[l] / (dl * dl * dl);
        }
      }
      const double f = lamb * lamb * s2 - lamb + zn;
      if (!solve && fabs(f) <= f0) bits |= 1ull << it;
      if (it == steps) break;
      const double df = -2.0 * lamb * s3 - 1.0;
      const double mu = lamb - f / df;
      lamb = mu > 1.0 ? mu : 0.5 * (1.0 + lamb);
    }
    if (solve) {
      lamb *= pm;
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        if (l < N && l != S) {
          const double gl = phi[l] / (lamb - phi[l]);
          cfma(qc, sg[l], cscale(vt[l], gl));
        }
      }
    }
  };
  if (is_singular) {
    // v = 0: scale * (last row of the eigenvector matrix in ascending-eigenvalue column order), see
    // lqpqm2 in ipa_kernels.hip
    double pmax = -1.7976931348623157e308;
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (l < N && l != S) pmax = fmax(pmax, phi[l]);
    const double lamb = fmax(z, pmax);
    const double scale = sqrt(fmax((lamb - z) / pmax, 0.0));
    const int last = S == N - 1 ? N - 2 : N - 1;
    const int mypos = r < S ? r : r - 1;  // my position among the rest indices
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const c128 val = cscale(shfl8(sg[l], last), scale);
      int rank = 0;
#pragma unroll
      for (int m = 0; m < 8; ++m)
        rank += (m < N && m != S && (phi[m] < phi[l] || (phi[m] == phi[l] && m < l))) ? 1 : 0;
      if (l < N && l != S && rest && rank == mypos) qc = val;
    }
  } else {
    newton(max_iter, MODE != NEWTON_FUSED);
  }
  if (MODE == NEWTON_FUSED) {
    // the mixture's vote: one AND per wave, then every workgroup of the mixture meets
    const bool voter = live && r == 0 && !is_singular;
    unsigned long long all = ~0ull;
    for (int it = 0; it <= max_iter; ++it)
      if (__ballot(voter && !((bits >> it) & 1ull)) != 0ull) all &= ~(1ull << it);
    if (lane == 0) __hip_atomic_fetch_and(word, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned *counter = (unsigned *)(newton_ws + (long long)B * N + (long long)blockIdx.y * N + S);
      vote_order();  // the AND has been performed
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1ll << 23)) {
          atomicAdd(&g_ipa_rows_barrier_timeouts, 1);
          break;
        }
      }
      vote_order();
      *vote_word = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned long long w = *vote_word;
    int agreed = max_iter;
    for (int k = 0; k < max_iter; ++k)
      if ((w >> k) & 1ull) {
        agreed = k;
        break;
      }
    if (blockIdx.x == 0 && threadIdx.x == 0 && agreed == max_iter && !((w >> max_iter) & 1ull) &&
        not_converged)
      atomicAdd(not_converged, 1);
    if (!is_singular) newton(agreed, true);
    __syncthreads();  // (vote_word is rewritten by the next source step)
  }

  // ---- q = q_check / a_sqrt - b / a;  q~ = e_S - E conj(q);  p = U_S^-1 q~ / floor(sqrt(q~^H U_S^-1 q~))
  const double a_r = seld(a, r);
  const c128 q = rest ? cmake(qc.x / as_r - br.x / a_r, qc.y / as_r - br.y / a_r) : cmake(0.0, 0.0);
  const c128 qt = r == S ? cmake(1.0, 0.0) : (rest ? cmake(-q.x, q.y) : cmake(0.0, 0.0));
  if (literal) {  // singly floored inverse
    rebuild(pl, r < N ? 1.0 / lamf : 0.0, X, r, uinv);
    wsync();
  } else if (floor_kind == SSSPY_FLOOR_ADD) {
    c128 t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) t[c] = us[c];
    put(t, r, r < N ? cmake(sel(us, r).x + eps, 0.0) : cmake(1.0, 0.0));
    const bool ok = chol_inverse_rows(t, X, r, uinv);
    if (!ok && info && live && r == 0) atomicAdd(info, 1);
  }
  c128 uq = cmake(0.0, 0.0), qtall[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    qtall[c] = shfl8(qt, c);
    cfma(uq, uinv[c], qtall[c]);
  }
  if (r >= N) uq = cmake(0.0, 0.0);
  const double quq = sum8(qt.x * uq.x + qt.y * uq.y);
  const double den = apply_floor(sqrt(fmax(quq, 0.0)), floor_kind, eps);
  c128 prow[8], gcol[8];  // row S of G_S; column S of G_S (entry S unused)
  const c128 gown = rest ? cmake(q.x, -q.y) : cmake(0.0, 0.0);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const c128 u = shfl8(uq, c);
    prow[c] = cmake(u.x / den, -u.y / den);
    gcol[c] = shfl8(gown, c);
  }
  c128 *Gb = G + bin * (long long)(N * N);
  if (!Vchain) {
    if (live && r < N) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < N) {
          c128 gv = cmake(r == c ? 1.0 : 0.0, 0.0);
          if (r == S) gv = prow[c];
          else if (c == S) gv = gown;
          Gb[r * N + c] = gv;
        }
    }
    return;
  }
  // ---- chained sweep: V_m <- G_S V_m G_S^H for every weight set, G <- G_S G
  c128 *Vb = Vchain + bin * (long long)(N * N * N);
#pragma unroll 1
  for (int m = 0; m <= N; ++m) {  // (m == N: the accumulated transform, left product only)
    const bool is_g = m == N;
    if (is_g && chain_first) {
      if (live && r < N) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < N) {
            c128 gv = cmake(r == c ? 1.0 : 0.0, 0.0);
            if (r == S) gv = prow[c];
            else if (c == S) gv = gown;
            Gb[r * N + c] = gv;
          }
      }
      break;
    }
    c128 *Mb = is_g ? Gb : Vb + m * N * N;
    c128 mr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) mr[c] = (r < N && c < N) ? Mb[r * N + c] : cmake(0.0, 0.0);
    // left: rows r != S gain g_r M[S]; row S becomes sum_k conj(p_k) M[k]
    wsync();
    store_row(X, r, mr);
    wsync();
    c128 srow[8];
    mul_rows(prow, X, srow);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      c128 t = mr[c];
      cfma(t, gown, X[S * LD + c]);
      mr[c] = r == S ? srow[c] : t;
    }
    wsync();
    if (!is_g) {
      // right: M G_S^H (column S = sum_k M[r][k] conj(prow[k]); the others gain M[r][S] conj(g_c))
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < 8; ++k) cfma(acc, mr[k], cconj(prow[k]));
      const c128 ms = sel(mr, S);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        c128 t = mr[c];
        cfma(t, ms, cconj(gcol[c]));  // (gcol[S] = 0)
        mr[c] = t;
      }
      put(mr, S, acc);
    }
    if (live && r < N) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < N) Mb[r * N + c] = mr[c];
    }
  }
  // the next source step re-reads what this one stored (other lanes' rows)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wsync();
}

// MODE NEWTON_FUSED / ROWS_FUSED_FIXED: with / without the mixtures' Newton votes.
// grid (ceil(F / 32), B), 256 threads = 32 bins x 8 lanes; N == 8 (lanes beyond N would idle)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_ipa_rows(c128 *Vc, c128 *__restrict__ G, int F, int N,
                                                     int normalization, int max_iter,
                                                     int floor_kind, double eps, int *info,
                                                     unsigned long long *newton_ws, int B,
                                                     int *not_converged) {
  __shared__ c128 slots[BINS * SLOT];
  __shared__ unsigned long long vote_word;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 7, g = wave * 8 + (lane >> 3);
  c128 *X = slots + g * SLOT;
  const int local = blockIdx.x * BINS + g;
  const bool live = local < F;
  // idle groups shadow the last bin of the mixture, never store or vote
  const long long bin = (long long)blockIdx.y * F + (live ? local : F - 1);
  c128 *Ub = Vc + bin * (long long)(N * N * N);
#define SSSPY_ROWS_STEP(S_)                                                                       \
  if (S_ < N)                                                                                     \
    ipa_rows_step<MODE, S_>(Ub, G, Vc, X, &vote_word, bin, live, r, lane, N, normalization,       \
                            max_iter, floor_kind, eps, info, newton_ws, B, not_converged);
  SSSPY_ROWS_STEP(0) SSSPY_ROWS_STEP(1) SSSPY_ROWS_STEP(2) SSSPY_ROWS_STEP(3)
  SSSPY_ROWS_STEP(4) SSSPY_ROWS_STEP(5) SSSPY_ROWS_STEP(6) SSSPY_ROWS_STEP(7)
#undef SSSPY_ROWS_STEP
}

}  // namespace

// Round 5, 16 x 1025 bins, us per launch (probe / apply), 8 lanes | a lane per bin: 8 sources 247 /
// 368 | 285 / 472; an ILRMA-IPA iteration 7.0 | 8.05 ms.  The kernel is bound by its chains of
// dependent instructions (pivot -> reciprocal -> broadcast -> update, 56 Jacobi rounds) with two
// waves per SIMD to interleave, so the padded 8 x 8 loses where the lane-per-bin form still fits the
// register file.  Round 6, whole sweep in one launch, 8 mixtures of 513 x 256, GaussILRMA /
// AuxLaplaceIVA iteration, lane per bin against rows: 5 sources 0.42 / 0.34 against 0.73 / 0.63 ms,
// 6 sources 0.77 / 0.65 against 0.96 / 0.78, 7 sources 1.45 / 1.30 against 1.26 / 1.04 (the chained
// lane-per-bin sweep of 7 sources spilled 17 000 VGPRs).  So: 7 and 8 sources here, 5 and 6 a lane
// per bin.
bool ipa_rows_wanted(int N) { return N >= 7; }

// the whole sweep of 7 / 8 sources: votes = the mixtures' Newton votes are held (else max_iter steps
// everywhere); ws: prepared by the caller (k_ipa_sweep_prepare)
int ipa_rows_sweep(bool votes, void *Vc, void *G, int B, int F, int N, int normalization,
                   int max_iter, int floor_kind, double eps, int *info, unsigned long long *ws,
                   int *not_converged, hipStream_t st) {
  const dim3 grid((unsigned)((F + BINS - 1) / BINS), (unsigned)B), block(256);
  if (votes)
    hipLaunchKernelGGL((k_ipa_rows<NEWTON_FUSED>), grid, block, 0, st, (c128 *)Vc, (c128 *)G, F, N,
                       normalization, max_iter, floor_kind, eps, info, ws, B, not_converged);
  else
    hipLaunchKernelGGL((k_ipa_rows<ROWS_FUSED_FIXED>), grid, block, 0, st, (c128 *)Vc, (c128 *)G, F,
                       N, normalization, max_iter, floor_kind, eps, info, ws, B, not_converged);
  return check_launch("k_ipa_rows");
}

int ipa_rows_barrier_timeouts() {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ipa_rows_barrier_timeouts), sizeof(int), 0,
                          hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return v;
}

}  // namespace ssspy
