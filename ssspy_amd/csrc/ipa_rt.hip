// IPA with the source count at run time (9 <= N <= SSSPY_RT_MAX_SOURCES = 16): the reference takes
// n_sources from input.shape without a limit (ssspy/bss/ilrma.py:180, iva.py:152), and IP1 / IP2 /
// ISS1 / ISS2 have run at 9..16 sources since round 4 (wide_n.hip) -- IPA was the one spatial
// algorithm that stopped at 8.  Correct, not tuned: one lane owns one bin and keeps its N x N working
// set in private (scratch) memory with run-time loops; the statements are those of ipa_source_step
// (ipa_kernels.hip), one by one, with the same sweep structure as k_ipa_sweep_fused -- the N source
// steps of a bin chained in one launch on the per-bin statistics, the Newton vote of a mixture met
// inside it.
//
// replaces (for N > 8): ssspy/bss/_update_spatial_model.py:398-513 (update_by_ipa), :611-645
//   (_psd_inv), ssspy/linalg/lqpqm.py:13-352 (lqpqm2, solve_equation), ssspy/linalg/cubic.py.
#include "common.hpp"
#include "hermitian.hpp"
#include "ipa_common.hpp"
#include "rt_dense.hpp"
#include "rt_hermitian.hpp"
#include "ssspy_amd.h"

namespace ssspy {

namespace {

constexpr int RN = SSSPY_RT_MAX_SOURCES;  // leading dimension of the private arrays is the run-time N

__device__ int g_ipa_rt_barrier_timeouts;

// the mixture-wide Newton vote of a source step (SweepVote of ipa_kernels.hip)
struct SweepVoteRt {
  unsigned long long *word;  // of this (mixture, source step), all ones before the sweep
  unsigned *counter;         // its arrival counter, zero before the sweep
  int nblocks, max_iter, *not_converged;
  __device__ __forceinline__ int operator()(unsigned long long bits, bool votes) const {
    unsigned long long all = ~0ull;
    for (int it = 0; it <= max_iter; ++it)
      if (__ballot(votes && !((bits >> it) & 1ull)) != 0ull) all &= ~(1ull << it);
    unsigned long long w = 0ull;
    if ((threadIdx.x & 63) == 0) {
      __hip_atomic_fetch_and(word, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vote_order();  // the AND has been performed
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
             (unsigned)nblocks) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1ll << 23)) {
          atomicAdd(&g_ipa_rt_barrier_timeouts, 1);
          break;
        }
      }
      vote_order();
      w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    w = __shfl(w, 0, 64);  // (one wave per workgroup)
    int steps = max_iter;
    for (int k = 0; k < max_iter; ++k)
      if ((w >> k) & 1ull) {
        steps = k;
        break;
      }
    if (blockIdx.x == 0 && threadIdx.x == 0 && steps == max_iter && !((w >> max_iter) & 1ull) &&
        not_converged)
      atomicAdd(not_converged, 1);
    return steps;
  }
};
struct NoVoteRt {
  __device__ __forceinline__ int operator()(unsigned long long, bool) const { return 0; }
};

// y = argmin of the LQPQM (type 2), H (L x L) Hermitian, v (L): lqpqm2 of ipa_kernels.hip with the
// NEWTON_FIXED (Vote = NoVoteRt) and NEWTON_FUSED (SweepVoteRt) modes.  sigma: scratch L x L.
template <bool FUSED, class Vote>
__device__ void rt_lqpqm2(c128 *H, c128 *sigma, const c128 *v, double z, int L, int floor_kind,
                          double eps, int max_iter, c128 *y, Vote vote,
                          int singular_override = -1) {
  rt_jacobi(H, sigma, L);
  double phi[RN], ph[RN], w2[RN];
  c128 vt[RN];
  for (int l = 0; l < L; ++l) phi[l] = H[l * L + l].x;
  const double f0 = floor_of_zero(floor_kind, eps);
  double vnorm2 = 0.0;
  for (int l = 0; l < L; ++l) vnorm2 += cabs2(v[l]);
  // v = 0 (see lqpqm2: the reference's literal indexing; the override: the caller's singular_fn)
  const bool is_singular = singular_override < 0 ? sqrt(vnorm2) < f0 : singular_override != 0;
  auto singular_solution = [&]() {
    double pmax = phi[0];
    for (int l = 1; l < L; ++l) pmax = fmax(pmax, phi[l]);
    const double lamb = fmax(z, pmax);
    const double scale = sqrt(fmax((lamb - z) / pmax, 0.0));
    for (int a = 0; a < L; ++a) y[a] = cmake(0.0, 0.0);
    for (int l = 0; l < L; ++l) {
      int rank = 0;  // position of phi[l] in ascending order (ties by index)
      for (int m = 0; m < L; ++m) rank += (phi[m] < phi[l] || (phi[m] == phi[l] && m < l)) ? 1 : 0;
      y[rank] = cscale(sigma[(L - 1) * L + l], scale);
    }
  };
  // (FUSED: a singular lane walks on with the others so that the wave meets at one vote)
  if (is_singular && !FUSED) {
    singular_solution();
    return;
  }
  for (int l = 0; l < L; ++l) {
    c128 s = cmake(0.0, 0.0);
    for (int a = 0; a < L; ++a) {
      const c128 t = cmulc(v[a], sigma[a * L + l]);  // v_a conj(sigma_al)
      s.x += t.x;
      s.y += t.y;
    }
    vt[l] = s;
  }
  double pmax = 0.0, v2max = 0.0;
  bool first = true;
  for (int l = 0; l < L; ++l) {
    const bool keep = phi[l] * cabs2(vt[l]) >= f0;
    ph[l] = keep ? phi[l] : 0.0;
    w2[l] = keep ? cabs2(vt[l]) : 0.0;
    if (first || ph[l] > pmax) {
      pmax = ph[l];
      v2max = w2[l];
      first = false;
    }
  }
  const double pm = apply_floor(pmax, floor_kind, eps);
  const double inv = 1.0 / pm;
  for (int l = 0; l < L; ++l) {
    ph[l] *= inv;
    w2[l] *= inv * inv;
  }
  const double zn = z * inv;
  const double A = -(v2max * inv * inv + 2.0 + zn), Bc = 1.0 + 2.0 * zn, Cc = -zn;
  double lamb = largest_cubic_root(A, Bc, Cc);
  if (!(lamb > 1.0)) lamb = 1.0 + f0;
  lamb = fmax(lamb, zn);
  const double lamb0 = lamb;
  const int steps = max_iter;
  unsigned long long bits = 0ull;
  for (int it = 0; it <= steps; ++it) {
    double s2 = 0.0, s3 = 0.0;
    for (int l = 0; l < L; ++l) {
      const double dl = lamb - ph[l];
      s2 += ph[l] * w2[l] / (dl * dl);
      s3 += ph[l] * ph[l] * w2[l] / (dl * dl * dl);
    }
    const double f = lamb * lamb * s2 - lamb + zn;
    if (fabs(f) <= f0) bits |= 1ull << it;
    if (it == steps) break;
    const double df = -2.0 * lamb * s3 - 1.0;
    const double mu = lamb - f / df;
    lamb = mu > 1.0 ? mu : 0.5 * (1.0 + lamb);
  }
  if (FUSED) {
    const int agreed = vote(bits, !is_singular);
    if (is_singular) {
      singular_solution();
      return;
    }
    if (agreed != steps) {  // (the mixture converged early: the reference stopped there)
      lamb = lamb0;
      for (int it = 0; it < agreed; ++it) {
        double s2 = 0.0, s3 = 0.0;
        for (int l = 0; l < L; ++l) {
          const double dl = lamb - ph[l];
          s2 += ph[l] * w2[l] / (dl * dl);
          s3 += ph[l] * ph[l] * w2[l] / (dl * dl * dl);
        }
        const double f = lamb * lamb * s2 - lamb + zn;
        const double df = -2.0 * lamb * s3 - 1.0;
        const double mu = lamb - f / df;
        lamb = mu > 1.0 ? mu : 0.5 * (1.0 + lamb);
      }
    }
  }
  lamb *= pm;
  for (int a = 0; a < L; ++a) y[a] = cmake(0.0, 0.0);
  for (int l = 0; l < L; ++l) {
    const double g = phi[l] / (lamb - phi[l]);
    const c128 coef = cscale(vt[l], g);
    for (int a = 0; a < L; ++a) cfma(y[a], sigma[a * L + l], coef);
  }
}

__device__ __forceinline__ int rest_rt(int S, int m) { return m < S ? m : m + 1; }

// One source step of a bin on the chained statistics (ipa_source_step with Vchain == Vc).
// M, P, Uinv, X1, X2: N x N scratch of the lane.
template <bool FUSED, class Vote>
__device__ void rt_ipa_step(c128 *Vc, c128 *G, long long bin, bool live, int N, int S,
                            int normalization, int max_iter, int floor_kind, double eps, int *info,
                            bool chain_first, c128 *M, c128 *P, c128 *Uinv, c128 *X1, c128 *X2,
                            Vote vote) {
  const int L = N - 1;
  c128 *Ub = Vc + bin * (long long)(N * N * N);
  double lam[RN], a[RN], as[RN], w[RN];
  c128 b[RN], d[RN], rhs[RN], v[RN], qc[RN], q[RN], qt[RN], Uq[RN], prow[RN], gcol[RN], srow[RN];
  // a_m = Re to_psd(U[m])[S][S], b_m = to_psd(U[m])[S][m] for the other sources
  for (int mm = 0; mm < L; ++mm) {
    const int m = rest_rt(S, mm);
    for (int e = 0; e < N * N; ++e) M[e] = Ub[m * N * N + e];
    rt_hermitize(M, N);
    double ass = M[S * N + S].x;
    c128 bsm = M[S * N + m];
    bool idle = floor_kind != SSSPY_FLOOR_MAX;
    if (floor_kind == SSSPY_FLOOR_ADD) ass += eps;
    if (floor_kind == SSSPY_FLOOR_MAX) idle = rt_shifted_pd(M, P, N, eps);
    if (!idle) {  // (the sweeps end on a test over the lanes that are here: __all sees the active ones)
      rt_psd_eigen(M, P, lam, N, floor_kind, eps);
      ass = 0.0;
      bsm = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) {
        ass = fma(lam[k], cabs2(P[S * N + k]), ass);
        const c128 t = cmulc(P[S * N + k], P[m * N + k]);
        bsm.x = fma(lam[k], t.x, bsm.x);
        bsm.y = fma(lam[k], t.y, bsm.y);
      }
    }
    a[mm] = ass;
    b[mm] = bsm;
  }
  // U_S: to_psd, then _psd_inv floors the floored eigenvalues again
  for (int e = 0; e < N * N; ++e) M[e] = Ub[S * N * N + e];
  rt_hermitize(M, N);
  bool literal = floor_kind == SSSPY_FLOOR_MAX && !rt_shifted_pd(M, P, N, eps);
  if (!literal) {
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c)
        P[r * N + c] = (r == c && floor_kind == SSSPY_FLOOR_ADD)
                           ? cmake(M[r * N + c].x + 2.0 * eps, 0.0)
                           : M[r * N + c];
    literal = !rt_chol_inverse(P, Uinv, X1, N);
  }
  if (literal) {
    rt_psd_eigen(M, P, lam, N, floor_kind, eps);
    for (int k = 0; k < N; ++k) w[k] = 1.0 / apply_floor(lam[k], floor_kind, eps);
    rt_rebuild(P, w, Uinv, N);
  }
  // C = conj(Uinv)[rest][rest] (X1: L x L, kept; X2: the copy the solver destroys), d, z
  for (int r = 0; r < L; ++r) {
    for (int c = 0; c < L; ++c) {
      X1[r * L + c] = cconj(Uinv[rest_rt(S, r) * N + rest_rt(S, c)]);
      X2[r * L + c] = X1[r * L + c];
    }
    d[r] = cconj(Uinv[rest_rt(S, r) * N + S]);
    rhs[r] = d[r];
  }
  const bool ok = rt_lu_solve(X2, rhs, L, 1);
  if (!ok && info && live) atomicAdd(info, 1);
  double dCd = 0.0;
  for (int r = 0; r < L; ++r) dCd += d[r].x * rhs[r].x + d[r].y * rhs[r].y;
  double z = Uinv[S * N + S].x - dCd;
  for (int r = 0; r < L; ++r) as[r] = sqrt(a[r]);
  // H (X2, L x L) and v
  double tr = 0.0;
  for (int r = 0; r < L; ++r) {
    for (int c = 0; c < L; ++c) {
      const double sc = 1.0 / (as[r] * as[c]);
      X2[r * L + c] = cscale(X1[r * L + c], sc);
    }
    tr += X2[r * L + r].x;
    v[r] = cmake(-b[r].x / as[r] - as[r] * rhs[r].x, -b[r].y / as[r] - as[r] * rhs[r].y);
  }
  if (normalization) {
    const double it = 1.0 / tr;
    for (int e = 0; e < L * L; ++e) X2[e] = cscale(X2[e], it);
    z *= it;
  }
  rt_hermitize(X2, L);
  rt_lqpqm2<FUSED, Vote>(X2, X1, v, z, L, floor_kind, eps, max_iter, qc, vote);
  // q = q_check / a_sqrt - b / a ; q~ = e_S - E conj(q)
  for (int r = 0; r < L; ++r)
    q[r] = cmake(qc[r].x / as[r] - b[r].x / a[r], qc[r].y / as[r] - b[r].y / a[r]);
  for (int m = 0; m < N; ++m) qt[m] = cmake(m == S ? 1.0 : 0.0, 0.0);
  for (int r = 0; r < L; ++r) qt[rest_rt(S, r)] = cmake(-q[r].x, q[r].y);
  // Uq = U_S^-1 q~ (single floor), p = Uq / floor(sqrt(max(q~^H Uq, 0)))
  if (literal) {
    for (int k = 0; k < N; ++k) w[k] = 1.0 / lam[k];
    rt_rebuild(P, w, Uinv, N);
  } else if (floor_kind == SSSPY_FLOOR_ADD) {
    // (M still holds the Hermitian part of U_S: the singly floored inverse is (A + eps I)^-1)
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c)
        P[r * N + c] = r == c ? cmake(M[r * N + c].x + eps, 0.0) : M[r * N + c];
    if (!rt_chol_inverse(P, Uinv, X1, N) && info && live) atomicAdd(info, 1);
  }
  double quq = 0.0;
  for (int r = 0; r < N; ++r) {
    c128 s = cmake(0.0, 0.0);
    for (int c = 0; c < N; ++c) cfma(s, Uinv[r * N + c], qt[c]);
    Uq[r] = s;
    quq += qt[r].x * s.x + qt[r].y * s.y;
  }
  const double den = apply_floor(sqrt(fmax(quq, 0.0)), floor_kind, eps);
  if (!live) return;  // (nothing below votes)
  // chained: V_m <- G_S V_m G_S^H for every weight set m, G <- G_S G
  for (int c = 0; c < N; ++c) {
    prow[c] = cmake(Uq[c].x / den, -Uq[c].y / den);
    gcol[c] = cmake(0.0, 0.0);
  }
  for (int r = 0; r < L; ++r) gcol[rest_rt(S, r)] = cmake(q[r].x, -q[r].y);
  for (int m = 0; m < N; ++m) {
    c128 *Vm = Ub + m * N * N;
    for (int e = 0; e < N * N; ++e) M[e] = Vm[e];
    // left: M <- G_S M
    for (int c = 0; c < N; ++c) {
      c128 acc = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(acc, prow[k], M[k * N + c]);
      srow[c] = acc;
    }
    for (int r = 0; r < N; ++r)
      if (r != S)
        for (int c = 0; c < N; ++c) cfma(M[r * N + c], gcol[r], M[S * N + c]);
    for (int c = 0; c < N; ++c) M[S * N + c] = srow[c];
    // right: M <- M G_S^H
    for (int r = 0; r < N; ++r) {
      c128 acc = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(acc, M[r * N + k], cconj(prow[k]));
      const c128 ms = M[r * N + S];
      for (int c = 0; c < N; ++c)
        if (c != S) cfma(M[r * N + c], ms, cconj(gcol[c]));
      M[r * N + S] = acc;
    }
    for (int e = 0; e < N * N; ++e) Vm[e] = M[e];
  }
  c128 *Gb = G + bin * (long long)(N * N);
  if (chain_first) {
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) {
        c128 g = cmake(r == c ? 1.0 : 0.0, 0.0);
        if (r == S) g = prow[c];
        else if (c == S) g = gcol[r];
        Gb[r * N + c] = g;
      }
  } else {
    for (int e = 0; e < N * N; ++e) M[e] = Gb[e];
    for (int c = 0; c < N; ++c) {
      c128 acc = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(acc, prow[k], M[k * N + c]);
      srow[c] = acc;
    }
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) {
        c128 g = M[r * N + c];
        if (r != S) cfma(g, gcol[r], M[S * N + c]);
        else g = srow[c];
        Gb[r * N + c] = g;
      }
  }
}

// grid: (ceil(F / 64), B), one wave per workgroup; ws: B N vote words, then B N arrival counters
template <bool FUSED>
__global__ __launch_bounds__(64) void k_ipa_sweep_rt(c128 *Vc, c128 *__restrict__ G, int F, int N,
                                                    int normalization, int max_iter,
                                                    int floor_kind, double eps, int *info,
                                                    unsigned long long *ws, int B,
                                                    int *not_converged) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool live = i < F;
  const long long bin = (long long)blockIdx.y * F + (live ? i : F - 1);
  c128 M[RN * RN], P[RN * RN], Uinv[RN * RN], X1[RN * RN], X2[RN * RN];
  for (int S = 0; S < N; ++S) {
    if (FUSED) {
      const int slot = blockIdx.y * N + S;
      SweepVoteRt vote{ws + slot, (unsigned *)(ws + (long long)B * N + slot), (int)gridDim.x,
                       max_iter, not_converged};
      rt_ipa_step<true, SweepVoteRt>(Vc, G, bin, live, N, S, normalization, max_iter, floor_kind,
                                     eps, info, S == 0, M, P, Uinv, X1, X2, vote);
    } else {
      rt_ipa_step<false, NoVoteRt>(Vc, G, bin, live, N, S, normalization, max_iter, floor_kind, eps,
                                   info, S == 0, M, P, Uinv, X1, X2, NoVoteRt());
    }
  }
}

// ---- the standalone lqpqm2 at 8 <= L <= 15 (ssspy_lqpqm2 / _masked; k_lqpqm2 of ipa_kernels.hip):
// the probe pass ANDs the convergence bits of all problems into one word (the lanes of a wave that
// are at the vote elect the first of them), k_newton_steps turns it into the reference's step count,
// the apply pass repeats that many steps from the start value.
struct ProbeVoteRt {
  unsigned long long *word;
  int max_iter;
  __device__ __forceinline__ int operator()(unsigned long long bits, bool votes) const {
    unsigned long long all = ~0ull;
    for (int it = 0; it <= max_iter; ++it)
      if (__ballot(votes && !((bits >> it) & 1ull)) != 0ull) all &= ~(1ull << it);
    const unsigned long long here = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)here) - 1) atomicAnd(word, all);
    return max_iter;
  }
};
struct ApplyVoteRt {
  const unsigned long long *word;  // the step count (k_newton_steps)
  __device__ __forceinline__ int operator()(unsigned long long, bool) const { return (int)*word; }
};

template <int MODE>
__global__ __launch_bounds__(64) void k_lqpqm2_rt(const c128 *__restrict__ H,
                                                  const c128 *__restrict__ v,
                                                  const double *__restrict__ z, c128 *y, long long n,
                                                  int L, int max_iter, int floor_kind, double eps,
                                                  unsigned long long *word,
                                                  const int *__restrict__ singular) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const long long idx = live ? i : n - 1;  // (every lane is in rt_jacobi's wave votes)
  c128 Hm[RN * RN], sigma[RN * RN], vv[RN], yy[RN];
  for (int r = 0; r < L; ++r) {
    vv[r] = v[idx * L + r];
    for (int c = 0; c < L; ++c) Hm[r * L + c] = H[(idx * L + r) * L + c];
  }
  rt_hermitize(Hm, L);
  const int so = singular ? singular[idx] : -1;
  if (MODE == NEWTON_FIXED) {
    rt_lqpqm2<false, NoVoteRt>(Hm, sigma, vv, z[idx], L, floor_kind, eps, max_iter, yy, NoVoteRt(),
                               so);
  } else if (MODE == NEWTON_PROBE) {
    // (a lane past the end repeats the last problem: its bits are that problem's)
    rt_lqpqm2<true, ProbeVoteRt>(Hm, sigma, vv, z[idx], L, floor_kind, eps, max_iter, yy,
                                 ProbeVoteRt{word, max_iter}, so);
    return;
  } else {
    rt_lqpqm2<true, ApplyVoteRt>(Hm, sigma, vv, z[idx], L, floor_kind, eps, max_iter, yy,
                                 ApplyVoteRt{word}, so);
  }
  if (live)
    for (int r = 0; r < L; ++r) y[idx * L + r] = yy[r];
}

}  // namespace

// one pass of the standalone lqpqm2 (mode: NEWTON_FIXED / NEWTON_PROBE / NEWTON_APPLY)
int lqpqm2_rt(int mode, const void *H, const void *v, const double *z, void *y, long long n, int L,
              int max_iter, int floor_kind, double eps, unsigned long long *word,
              const int *singular, hipStream_t st) {
  const dim3 grid((unsigned)((n + 63) / 64)), block(64);
#define LQ_RT(MODE_)                                                                              \
  hipLaunchKernelGGL(k_lqpqm2_rt<MODE_>, grid, block, 0, st, (const c128 *)H, (const c128 *)v, z, \
                     (c128 *)y, n, L, max_iter, floor_kind, eps, word, singular)
  if (mode == NEWTON_FIXED) LQ_RT(NEWTON_FIXED);
  else if (mode == NEWTON_PROBE) LQ_RT(NEWTON_PROBE);
  else LQ_RT(NEWTON_APPLY);
#undef LQ_RT
  return check_launch("k_lqpqm2_rt");
}

// the whole sweep of 9..16 sources (ws prepared by the caller: k_ipa_sweep_prepare)
int ipa_rt_sweep(bool votes, void *Vc, void *G, int B, int F, int N, int normalization,
                 int max_iter, int floor_kind, double eps, int *info, unsigned long long *ws,
                 int *not_converged, hipStream_t st) {
  const dim3 grid((unsigned)((F + 63) / 64), (unsigned)B), block(64);
  if (votes)
    hipLaunchKernelGGL(k_ipa_sweep_rt<true>, grid, block, 0, st, (c128 *)Vc, (c128 *)G, F, N,
                       normalization, max_iter, floor_kind, eps, info, ws, B, not_converged);
  else
    hipLaunchKernelGGL(k_ipa_sweep_rt<false>, grid, block, 0, st, (c128 *)Vc, (c128 *)G, F, N,
                       normalization, max_iter, floor_kind, eps, info, ws, B, not_converged);
  return check_launch("k_ipa_sweep_rt");
}

int ipa_rt_barrier_timeouts() {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ipa_rt_barrier_timeouts), sizeof(int), 0,
                          hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return v;
}

}  // namespace ssspy
