// Iterative projection with adjustment (IPA): the per-bin update matrix of one source step.
//
// One lane owns one bin.  Given the weighted covariances of the CURRENT separated spectrogram,
// U[n] = mean_j varphi_nj y_j y_j^H for every weight set n, the lane floors them (to_psd), forms
// the (N-1)-dimensional log-quadratically penalised quadratic problem of source s, solves it
// (Hermitian Jacobi eigen-decomposition, Cardano start value, Newton steps on the secular
// equation), and writes the N x N matrix G with  y <- G y:
//   row s of G = p^H,  G[m][s] = conj(q_m) for m != s,  identity elsewhere.
// The reference runs  weighted_covariance -> this -> separate  once per source; here -- round 5 --
// weighted_covariance runs once, the N source steps chained on the per-bin statistics
// (V_m <- G V_m G^H), one separate with the accumulated transform (ssspy_ipa_sweep): 3 passes over the
// spectrogram per sweep instead of 3 N.
//
// replaces: ssspy/bss/_update_spatial_model.py:398-513 (update_by_ipa), :611-645 (_psd_inv),
//           ssspy/linalg/lqpqm.py:13-352 (lqpqm2, solve_equation, _find_largest_root),
//           ssspy/linalg/cubic.py (polar complex cube root).
// The Newton loop of the reference runs over all bins of a mixture at once and stops at the first
// step at which EVERY (non-singular) bin has converged (lqpqm.py:196-213) -- a reduction across bins
// in the middle of a per-bin computation.  Every bin's trajectory is independent of the others for
// as long as the loop runs, so the step count is found by a PROBE pass: the same kernel up to the
// Newton loop, run for all max_iter steps, each bin leaving the bit mask "converged at step k" (and
// after the last step) AND-ed into one 64-bit word per mixture; k_newton_steps turns the word into
// the number of steps the reference would have made (and counts the mixtures that never converged:
// the reference's UserWarning); the real pass then makes exactly that many steps.  Without the word
// (newton_ws == NULL, or max_iter > 62) every bin runs max_iter steps.
// One difference from the reference remains: the ||v|| ~ 0 branch returns a scaled eigenvector whose
// phase is the decomposition's.
#include "common.hpp"
#include "hermitian.hpp"
#include "ipa_common.hpp"
#include "smallmat.hpp"
#include "ssspy_amd.h"

namespace ssspy {

// y = argmin of the LQPQM (type 2) with H = sigma diag(phi) sigma^H.  ref: lqpqm.py:13-110
// mode: NEWTON_FIXED max_iter steps; NEWTON_PROBE max_iter steps, convergence bits AND-ed into *word
// (nothing else is produced); NEWTON_APPLY the number of steps found in *word by k_newton_steps
// NEWTON_FUSED (round 6): probe and apply in one kernel.  After the max_iter probing steps the lane
// hands its convergence bits to `vote`, which returns the number of steps the reference makes for
// the lane's mixture (a reduction over all bins of the mixture: k_ipa_sweep_fused waits for the
// other workgroups of the mixture there), and the lane repeats that many steps from the start
// value.  Every lane of the wave must reach the vote: singular problems hand in `false`.
struct NoVote {
  __device__ __forceinline__ int operator()(unsigned long long, bool) const { return 0; }
};

template <int L, class Vote = NoVote>
__device__ __forceinline__ void lqpqm2(c128 (&H)[L][L], const c128 (&v)[L], double z,
                                       int floor_kind, double eps, int max_iter, c128 (&y)[L],
                                       int mode = NEWTON_FIXED, unsigned long long *word = nullptr,
                                       int singular_override = -1, Vote vote = Vote()) {
  c128 sigma[L][L];
  jacobi_eigh<L>(H, sigma);
  double phi[L];
#pragma unroll
  for (int l = 0; l < L; ++l) phi[l] = H[l][l].x;
  const double f0 = floor_of_zero(floor_kind, eps);
  double vnorm2 = 0.0;
#pragma unroll
  for (int l = 0; l < L; ++l) vnorm2 += cabs2(v[l]);
  // singular_override: the caller evaluated the reference's singular_fn on ||v|| itself (None:
  // x == 0, or any callable; lqpqm.py:61-78); -1: the default "x < flooring_fn(0)"
  const bool is_singular = singular_override < 0 ? sqrt(vnorm2) < f0 : singular_override != 0;
  // v = 0.  The reference returns scale * sigma[:, -1] taken on the (n_bins, L, L) eigenvector
  // array (lqpqm.py:84-93), i.e. the LAST ROW of each eigenvector matrix in ascending-eigenvalue
  // column order, not its last column: component a is the last entry of the eigenvector of the
  // a-th smallest eigenvalue (every entry with that eigenvector's arbitrary phase).  Followed
  // literally -- the moduli are what parity can pin.
  auto singular_solution = [&]() {
    double pmax = phi[0];
#pragma unroll
    for (int l = 1; l < L; ++l) pmax = fmax(pmax, phi[l]);
    const double lamb = fmax(z, pmax);
    const double scale = sqrt(fmax((lamb - z) / pmax, 0.0));
#pragma unroll
    for (int a = 0; a < L; ++a) y[a] = cmake(0.0, 0.0);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      int rank = 0;  // position of phi[l] in ascending order (ties by index)
#pragma unroll
      for (int m = 0; m < L; ++m) rank += (phi[m] < phi[l] || (phi[m] == phi[l] && m < l)) ? 1 : 0;
      const c128 val = cscale(sigma[L - 1][l], scale);
#pragma unroll
      for (int a = 0; a < L; ++a)
        if (a == rank) y[a] = val;
    }
  };
  // NEWTON_FUSED: a singular lane walks through the Newton arithmetic with the others (its numbers
  // are not used) so that the wave meets at ONE vote -- with a vote of its own in this branch, a
  // wave whose lane 0 is singular handed in the singular lanes' empty ballot and lost the rest's.
  if (is_singular && mode != NEWTON_FUSED) {
    singular_solution();
    return;
  }
  c128 vt[L];  // sigma^H v
#pragma unroll
  for (int l = 0; l < L; ++l) {
    c128 s = cmake(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < L; ++a) {
      const c128 t = cmulc(v[a], sigma[a][l]);  // v_a conj(sigma_al)
      s.x += t.x;
      s.y += t.y;
    }
    vt[l] = s;
  }
  // ---- solve_equation (normalization=True), ref: lqpqm.py:112-200
  double ph[L], w2[L];  // masked, normalised phi and |v|^2
  double pmax = 0.0, v2max = 0.0;
  bool first = true;
#pragma unroll
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
#pragma unroll
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
  const int steps = mode == NEWTON_APPLY ? (int)*word : max_iter;
  unsigned long long bits = 0ull;
  for (int it = 0; it <= steps; ++it) {
    double s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const double dl = lamb - ph[l];
      s2 += ph[l] * w2[l] / (dl * dl);
      s3 += ph[l] * ph[l] * w2[l] / (dl * dl * dl);
    }
    const double f = lamb * lamb * s2 - lamb + zn;
    if (fabs(f) <= f0) bits |= 1ull << it;  // (bit `steps`: the state after the last step)
    if (it == steps) break;
    const double df = -2.0 * lamb * s3 - 1.0;
    const double mu = lamb - f / df;
    lamb = mu > 1.0 ? mu : 0.5 * (1.0 + lamb);
  }
  if (mode == NEWTON_FUSED) {
    const int agreed = vote(bits, !is_singular);
    if (is_singular) {
      singular_solution();
      return;
    }
    if (agreed != steps) {  // (the mixture converged early: the reference stopped there)
      lamb = lamb0;
      for (int it = 0; it < agreed; ++it) {
        double s2 = 0.0, s3 = 0.0;
#pragma unroll
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
  if (mode == NEWTON_PROBE) {
    // One atomic per (wave, mixture), not one per bin: 1025 atomics on one word took 130 of the
    // probe launch's 205 us (32 mixtures of configs[1]; the apply launch: 75 us).  The lanes still
    // here (singular problems have left: they do not vote) that share a word elect a leader; bit k of
    // the group is set when no member has it clear.
    unsigned long long todo = __ballot(1);
    const unsigned long long mine_word = (unsigned long long)word;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const unsigned long long lw = __shfl(mine_word, leader, 64);
      const bool mine = mine_word == lw;
      const unsigned long long group = __ballot(mine);
      unsigned long long all = 0ull;
      for (int it = 0; it <= steps; ++it)
        if (__ballot(mine && !((bits >> it) & 1ull)) == 0ull) all |= 1ull << it;
      if ((int)(threadIdx.x & 63) == leader) atomicAnd(word, all);
      todo &= ~group;
    }
    return;
  }
  lamb *= pm;
  // y = sigma (phi * vt / (lamb - phi)) with the unmasked phi, vt
#pragma unroll
  for (int a = 0; a < L; ++a) y[a] = cmake(0.0, 0.0);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const double g = phi[l] / (lamb - phi[l]);
    const c128 coef = cscale(vt[l], g);
#pragma unroll
    for (int a = 0; a < L; ++a) cfma(y[a], sigma[a][l], coef);
  }
}

// lam_min(A) > shift for a Hermitian A, by the pivots of the Cholesky factorisation of A - shift I
// (W: working copy; A is left intact)
template <int N>
__device__ __forceinline__ bool shifted_positive_definite(const c128 (&A)[N][N], c128 (&W)[N][N],
                                                          double shift) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    double d = A[c][c].x - shift;
#pragma unroll
    for (int k = 0; k < c; ++k) d -= cabs2(W[c][k]);
    ok = ok && (d > 0.0);
    const double il = 1.0 / sqrt(d > 0.0 ? d : 1.0);
#pragma unroll
    for (int r = c + 1; r < N; ++r) {
      c128 sum = A[r][c];
#pragma unroll
      for (int k = 0; k < c; ++k) cfms(sum, W[r][k], cconj(W[c][k]));
      W[r][c] = cscale(sum, il);
    }
  }
  return ok;
}

// index of the m-th source other than S
template <int S>
__device__ __forceinline__ constexpr int rest_index(int m) {
  return m < S ? m : m + 1;
}

// One source step of a bin (the body of the fused sweep).
// `live`: false in the lanes a fused sweep keeps behind the last bin of its mixture (they walk a
// copy of that bin so that the wave reaches every vote, and store nothing)
template <int N, int S, int MODE, class Vote = NoVote>
__device__ __forceinline__ void ipa_source_step(const c128 *Vc, c128 *__restrict__ G,
                                                long long bin, bool live, int normalization,
                                                int max_iter, int floor_kind, double eps, int *info,
                                                unsigned long long *word, c128 *Vchain,
                                                int chain_first, Vote vote = Vote()) {
  constexpr int L = N - 1;
  const c128 *Ub = Vc + bin * (long long)(N * N * N);
  c128 M[N][N], P[N][N];
  double lam[N];
  // Round 5: to_psd without an eigen-decomposition where its floor provably does nothing.  The
  // reference's to_psd is P diag(floor(lam)) P^H of the Hermitian part A (special/psd.py:54-71): A
  // itself under the identity floor, A + eps I under the add floor, and A under the max floor
  // whenever lam_min(A) > eps -- which a Cholesky factorisation of A - eps I decides (positive
  // pivots <=> positive definite) at a thirtieth of the cost of the Jacobi sweeps.  N + 1 eigen-
  // decompositions per source step become none in the common case (the LQPQM problem keeps its own);
  // a lane whose test fails takes the literal route (`literal`).
  // a_m = Re to_psd(U[m])[S][S], b_m = to_psd(U[m])[S][m] for the other sources
  double a[L];
  c128 b[L];
#pragma unroll 1
  for (int mm = 0; mm < L; ++mm) {
    const int m = rest_index<S>(mm);
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) M[r][c] = Ub[(m * N + r) * N + c];
    hermitize<N>(M);
    double ass = 0.0;
    c128 bsm = cmake(0.0, 0.0);
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) {
        if (r == S && c == S) ass = M[r][c].x;
        if (r == S && c == m) bsm = M[r][c];
      }
    bool idle = floor_kind != SSSPY_FLOOR_MAX;
    if (floor_kind == SSSPY_FLOOR_ADD) ass += eps;
    if (floor_kind == SSSPY_FLOOR_MAX) idle = shifted_positive_definite<N>(M, P, eps);
    if (!idle) {  // (M is intact: the test works on a copy)
      psd_eigen<N>(M, P, lam, floor_kind, eps);
      ass = 0.0;
      bsm = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        ass = fma(lam[k], cabs2(P[S][k]), ass);
        const c128 t = cmulc(P[S][k], P[m][k]);
        bsm.x = fma(lam[k], t.x, bsm.x);
        bsm.y = fma(lam[k], t.y, bsm.y);
      }
    }
    a[mm] = ass;
    b[mm] = bsm;
  }
  // U_S: to_psd, then _psd_inv floors the floored eigenvalues again (identity / max floor idle:
  // U_S^-1 by Cholesky; add floor: (A + 2 eps I)^-1 here and (A + eps I)^-1 for the final solve)
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c) M[r][c] = Ub[(S * N + r) * N + c];
  hermitize<N>(M);
  c128 Uinv[N][N];
  bool literal = floor_kind == SSSPY_FLOOR_MAX && !shifted_positive_definite<N>(M, P, eps);
  if (!literal) {
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c)
        P[r][c] = (r == c && floor_kind == SSSPY_FLOOR_ADD) ? cmake(M[r][c].x + 2.0 * eps, 0.0)
                                                            : M[r][c];
    double ld;
    literal = !chol_inverse<N>(P, Uinv, ld);  // (not positive definite: the eigen route copes)
  }
  double w[N];
  if (literal) {
    psd_eigen<N>(M, P, lam, floor_kind, eps);
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = 1.0 / apply_floor(lam[k], floor_kind, eps);
    herm_rebuild<N>(P, w, Uinv);
  }
  // C = conj(Uinv)[rest][rest], d = conj(Uinv)[rest][S]
  Mat<L> C;
  c128 Cm[L][L], d[L], rhs[L][1];
#pragma unroll
  for (int r = 0; r < L; ++r) {
#pragma unroll
    for (int c = 0; c < L; ++c) {
      Cm[r][c] = cconj(Uinv[rest_index<S>(r)][rest_index<S>(c)]);
      C.a[r][c] = Cm[r][c];
    }
    d[r] = cconj(Uinv[rest_index<S>(r)][S]);
    rhs[r][0] = d[r];
  }
  const bool ok = lu_forward<L, 1>(C, rhs);
  lu_backward<L, 1>(C, rhs);
  if (MODE != NEWTON_PROBE && !ok && info && live) atomicAdd(info, 1);
  double dCd = 0.0;
#pragma unroll
  for (int r = 0; r < L; ++r) dCd += d[r].x * rhs[r][0].x + d[r].y * rhs[r][0].y;
  double z = Uinv[S][S].x - dCd;
  double as[L];
#pragma unroll
  for (int r = 0; r < L; ++r) as[r] = sqrt(a[r]);
  c128 H[L][L], v[L];
  double tr = 0.0;
#pragma unroll
  for (int r = 0; r < L; ++r) {
#pragma unroll
    for (int c = 0; c < L; ++c) {
      const double sc = 1.0 / (as[r] * as[c]);
      H[r][c] = cscale(Cm[r][c], sc);
    }
    tr += H[r][r].x;
    // v = -b / a_sqrt - a_sqrt * Cd
    v[r] = cmake(-b[r].x / as[r] - as[r] * rhs[r][0].x, -b[r].y / as[r] - as[r] * rhs[r][0].y);
  }
  if (normalization) {
    const double it = 1.0 / tr;
#pragma unroll
    for (int r = 0; r < L; ++r)
#pragma unroll
      for (int c = 0; c < L; ++c) H[r][c] = cscale(H[r][c], it);
    z *= it;
  }
  hermitize<L>(H);
  c128 qc[L];
  lqpqm2<L, Vote>(H, v, z, floor_kind, eps, max_iter, qc, MODE, word, -1, vote);
  if (MODE == NEWTON_PROBE) return;
  // q = q_check / a_sqrt - b / a ; q~ = e_S - E conj(q)
  c128 q[L], qt[N];
#pragma unroll
  for (int r = 0; r < L; ++r)
    q[r] = cmake(qc[r].x / as[r] - b[r].x / a[r], qc[r].y / as[r] - b[r].y / a[r]);
#pragma unroll
  for (int m = 0; m < N; ++m) qt[m] = cmake(m == S ? 1.0 : 0.0, 0.0);
#pragma unroll
  for (int r = 0; r < L; ++r) qt[rest_index<S>(r)] = cmake(-q[r].x, q[r].y);
  // Uq = U_S^-1 q~ (single floor), p = Uq / floor(sqrt(max(q~^H Uq, 0)))
  if (literal) {
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = 1.0 / lam[k];
    herm_rebuild<N>(P, w, Uinv);
  } else if (floor_kind == SSSPY_FLOOR_ADD) {
    // (M still holds the Hermitian part of U_S: the singly floored inverse is (A + eps I)^-1)
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c)
        P[r][c] = r == c ? cmake(M[r][c].x + eps, 0.0) : M[r][c];
    double ld;
    if (!chol_inverse<N>(P, Uinv, ld) && info && live) atomicAdd(info, 1);
  }  // (identity / idle max floor: Uinv is U_S^-1 already)
  c128 Uq[N];
  double quq = 0.0;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    c128 s = cmake(0.0, 0.0);
#pragma unroll
    for (int c = 0; c < N; ++c) cfma(s, Uinv[r][c], qt[c]);
    Uq[r] = s;
    quq += qt[r].x * s.x + qt[r].y * s.y;
  }
  const double den = apply_floor(sqrt(fmax(quq, 0.0)), floor_kind, eps);
  if (!live) return;  // (nothing below votes)
  c128 *Gb = G + bin * (long long)(N * N);
  if (!Vchain) {
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 g = cmake(r == c ? 1.0 : 0.0, 0.0);
        if (r == S) g = cmake(Uq[c].x / den, -Uq[c].y / den);  // conj(p_c)
        Gb[r * N + c] = g;
      }
#pragma unroll
    for (int r = 0; r < L; ++r) Gb[rest_index<S>(r) * N + S] = cmake(q[r].x, -q[r].y);
    return;
  }
  // ---- chained sweep (ssspy_ipa_sweep): the step's update y <- G_S y is not applied to the
  // spectrogram; the statistics follow it, V_m <- G_S V_m G_S^H for every weight set m (what the
  // reference recomputes from the updated spectrogram, _update_spatial_model.py:442-445), and the
  // accumulated transform G <- G_S G.  G_S = I except row S (= p^H) and column S (= conj(q)):
  //   (G_S A)[r] = A[r] + g_r A[S] (r != S),  (G_S A)[S] = sum_c conj(p_c) A[c].
  c128 prow[N], gcol[N];  // row S of G_S; column S of G_S (entry S unused)
#pragma unroll
  for (int c = 0; c < N; ++c) {
    prow[c] = cmake(Uq[c].x / den, -Uq[c].y / den);
    gcol[c] = cmake(0.0, 0.0);
  }
#pragma unroll
  for (int r = 0; r < L; ++r) gcol[rest_index<S>(r)] = cmake(q[r].x, -q[r].y);
  c128 *Vb = Vchain + bin * (long long)(N * N * N);
  for (int m = 0; m < N; ++m) {  // (rolled: one matrix live)
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) M[r][c] = Vb[(m * N + r) * N + c];
    // left: M <- G_S M
    c128 srow[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) cfma(acc, prow[k], M[k][c]);
      srow[c] = acc;
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
      if (r != S) {
#pragma unroll
        for (int c = 0; c < N; ++c) cfma(M[r][c], gcol[r], M[S][c]);
      }
#pragma unroll
    for (int c = 0; c < N; ++c) M[S][c] = srow[c];
    // right: M <- M G_S^H   (G_S^H: column S = conj(row S of G_S) = p, row S = q^T elsewhere)
#pragma unroll
    for (int r = 0; r < N; ++r) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) cfma(acc, M[r][k], cconj(prow[k]));
      const c128 ms = M[r][S];
#pragma unroll
      for (int c = 0; c < N; ++c)
        if (c != S) cfma(M[r][c], ms, cconj(gcol[c]));
      M[r][S] = acc;
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) Vb[(m * N + r) * N + c] = M[r][c];
  }
  // G <- G_S G (the first step: G = G_S)
  if (chain_first) {
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 g = cmake(r == c ? 1.0 : 0.0, 0.0);
        if (r == S) g = prow[c];
        else if (c == S) g = gcol[r];
        Gb[r * N + c] = g;
      }
  } else {
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) M[r][c] = Gb[r * N + c];
    c128 srow[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) cfma(acc, prow[k], M[k][c]);
      srow[c] = acc;
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 g = M[r][c];
        if (r != S) cfma(g, gcol[r], M[S][c]);
        else g = srow[c];
        Gb[r * N + c] = g;
      }
  }
}

// ---- the whole sweep of up to 6 sources in ONE launch (round 6).  One lane per bin; one wave per
// SIMD (the N x N working set of the larger source counts wants the whole 512-entry register file;
// the grid has only B F lanes anyway).  Rounds 4-5 spent four launches
// per source step: a memset of the vote words, the probe (everything up to the Newton iteration,
// 50 us at 32 mixtures of configs[1]), the step count, and the apply, which recomputes the probe's
// state (31 us) -- 16 launches and 0.33 ms per iteration at 4 sources.  Here a workgroup (one wave,
// 64 bins of ONE mixture) walks the N source steps; at each Newton vote it ANDs its wave's
// convergence bits into the (mixture, step) word and waits for the mixture's other workgroups
// (arrival counter, agent-scope release / acquire), then carries on with the state it holds.
// Forward progress: a mixture's workgroups have consecutive ids, workgroups are dispatched in
// order, and the chip holds far more than one mixture's ceil(F / 64) of them, so the waiting
// workgroups of the frontier mixture are always joined by the rest; the wait is bounded anyway
// (g_ipa_barrier_timeouts, read by ssspy_debug_barrier_timeouts).
__device__ int g_ipa_barrier_timeouts;

struct SweepVote {
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
        if (++spins > (1ll << 23)) {  // (seconds: the launch is broken, not slow)
          atomicAdd(&g_ipa_barrier_timeouts, 1);
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

// ws: B N vote words, then B N arrival counters (64-bit slots); prepared by k_ipa_sweep_prepare
__global__ void k_ipa_sweep_prepare(unsigned long long *ws, int count) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) {
    ws[e] = ~0ull;
    ws[count + e] = 0ull;
  }
}

template <int N, int S, int MODE>
__device__ __forceinline__ void ipa_sweep_steps(c128 *Vc, c128 *G, long long bin, bool live,
                                                int normalization, int max_iter, int floor_kind,
                                                double eps, int *info, unsigned long long *ws,
                                                int nblocks, int B, int *not_converged) {
  if constexpr (S < N) {
    const int slot = blockIdx.y * N + S;
    SweepVote vote{ws + slot, (unsigned *)(ws + (long long)B * N + slot), nblocks, max_iter,
                   not_converged};
    ipa_source_step<N, S, MODE, SweepVote>(Vc, G, bin, live, normalization, max_iter, floor_kind,
                                           eps, info, nullptr, Vc, S == 0, vote);
    ipa_sweep_steps<N, S + 1, MODE>(Vc, G, bin, live, normalization, max_iter, floor_kind, eps,
                                    info, ws, nblocks, B, not_converged);
  }
}

// grid: (ceil(F / 64), B), one wave per workgroup; MODE: NEWTON_FUSED or NEWTON_FIXED (no votes)
template <int N, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ipa_sweep_fused(
    c128 *Vc, c128 *__restrict__ G, int F, int normalization, int max_iter, int floor_kind,
    double eps, int *info, unsigned long long *ws, int B, int *not_converged) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool live = i < F;
  const long long bin = (long long)blockIdx.y * F + (live ? i : F - 1);
  ipa_sweep_steps<N, 0, MODE>(Vc, G, bin, live, normalization, max_iter, floor_kind, eps, info, ws,
                              (int)gridDim.x, B, not_converged);
}

// standalone LQPQM2 (ssspy.linalg.lqpqm2): H (n, L, L), v (n, L), z (n) -> y (n, L)
// word <- the number of Newton steps the reference makes: the first step at which every problem of
// the group had converged (it stops BEFORE that step's update), else max_iter; groups that had not
// converged after the last step either are counted in *not_converged.  grid: 1 block, thread = group
__global__ void k_newton_steps(unsigned long long *words, int ngroups, int max_iter,
                               int *not_converged) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const unsigned long long w = words[g];
  int steps = max_iter;
  for (int k = 0; k < max_iter; ++k)
    if ((w >> k) & 1ull) {
      steps = k;
      break;
    }
  if (steps == max_iter && !((w >> max_iter) & 1ull) && not_converged) atomicAdd(not_converged, 1);
  words[g] = (unsigned long long)steps;
}

template <int L, int MODE>
__global__ __launch_bounds__(64) void k_lqpqm2(const c128 *__restrict__ H, const c128 *__restrict__ v,
                                               const double *__restrict__ z, c128 *y, long long n,
                                               int max_iter, int floor_kind, double eps,
                                               unsigned long long *newton_ws,
                                               const int *__restrict__ singular) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  c128 Hm[L][L], vv[L], yy[L];
#pragma unroll
  for (int r = 0; r < L; ++r) {
    vv[r] = v[idx * L + r];
#pragma unroll
    for (int c = 0; c < L; ++c) Hm[r][c] = H[(idx * L + r) * L + c];
  }
  hermitize<L>(Hm);
  lqpqm2<L>(Hm, vv, z[idx], floor_kind, eps, max_iter, yy, MODE, newton_ws,
            singular ? singular[idx] : -1);
  if (MODE == NEWTON_PROBE) return;
#pragma unroll
  for (int r = 0; r < L; ++r) y[idx * L + r] = yy[r];
}

// all-ones words, probe pass, words -> step counts
static int newton_prepare(unsigned long long *ws, int ngroups, hipStream_t st) {
  hipError_t e = hipMemsetAsync(ws, 0xFF, (size_t)ngroups * sizeof(unsigned long long), st);
  return e == hipSuccess ? SSSPY_OK : fail(SSSPY_ERR_HIP, hipGetErrorString(e));
}
static int newton_finish(unsigned long long *ws, int ngroups, int max_iter, int *not_converged,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_newton_steps, dim3((ngroups + 255) / 256), dim3(256), 0, st, ws, ngroups,
                     max_iter, not_converged);
  return check_launch("k_newton_steps");
}

// ipa_rows.hip: the sweep of 7 / 8 sources with a bin on 8 lanes
bool ipa_rows_wanted(int N);
int ipa_rows_sweep(bool votes, void *Vc, void *G, int B, int F, int N, int normalization,
                   int max_iter, int floor_kind, double eps, int *info, unsigned long long *ws,
                   int *not_converged, hipStream_t st);
int ipa_rows_barrier_timeouts();
// ipa_rt.hip: 9..16 sources with the source count at run time (a lane per bin, scratch-resident)
int ipa_rt_sweep(bool votes, void *Vc, void *G, int B, int F, int N, int normalization,
                 int max_iter, int floor_kind, double eps, int *info, unsigned long long *ws,
                 int *not_converged, hipStream_t st);
int ipa_rt_barrier_timeouts();
int lqpqm2_rt(int mode, const void *H, const void *v, const double *z, void *y, long long n, int L,
              int max_iter, int floor_kind, double eps, unsigned long long *word,
              const int *singular, hipStream_t st);

}  // namespace ssspy

using namespace ssspy;

extern "C" int ssspy_ipa_sweep(void *Vc, void *G, int B, int F, int N, int normalization,
                               int max_iter, int floor_kind, double floor_eps, int *info,
                               void *newton_ws, int *not_converged, void *stream) {
  SSSPY_REQUIRE(Vc && G && B > 0 && F > 0, "ipa_sweep: bad argument");
  SSSPY_REQUIRE(max_iter >= 0, "ipa_sweep: max_iter must be non-negative");
  if (N < 2 || N > SSSPY_RT_MAX_SOURCES)
    return fail(SSSPY_ERR_UNSUPPORTED, "IPA is built for n_sources in [2, 16]");
  // one launch for the whole sweep: a lane per bin up to 6 sources (k_ipa_sweep_fused), a bin on 8
  // lanes at 7 and 8 (k_ipa_rows)
  hipStream_t st = as_stream(stream);
  const bool votes = newton_ws && max_iter >= 1 && max_iter <= 62;
  unsigned long long *ws = (unsigned long long *)newton_ws;
  if (votes) {
    hipLaunchKernelGGL(k_ipa_sweep_prepare, dim3((B * N + 255) / 256), dim3(256), 0, st, ws, B * N);
    const int rc = check_launch("k_ipa_sweep_prepare");
    if (rc) return rc;
  }
  if (N > SSSPY_MAX_SOURCES)
    return ipa_rt_sweep(votes, Vc, G, B, F, N, normalization, max_iter, floor_kind, floor_eps, info,
                        ws, not_converged, st);
  if (ipa_rows_wanted(N))
    return ipa_rows_sweep(votes, Vc, G, B, F, N, normalization, max_iter, floor_kind, floor_eps,
                          info, ws, not_converged, st);
  const dim3 grid((F + 63) / 64, B), block(64);
#define IPA_FUSED(N_)                                                                              \
  if (N == N_) {                                                                                   \
    if (votes)                                                                                     \
      hipLaunchKernelGGL((k_ipa_sweep_fused<N_, NEWTON_FUSED>), grid, block, 0, st, (c128 *)Vc,    \
                         (c128 *)G, F, normalization, max_iter, floor_kind, floor_eps, info, ws, B, \
                         not_converged);                                                           \
    else                                                                                           \
      hipLaunchKernelGGL((k_ipa_sweep_fused<N_, NEWTON_FIXED>), grid, block, 0, st, (c128 *)Vc,    \
                         (c128 *)G, F, normalization, max_iter, floor_kind, floor_eps, info, ws, B, \
                         not_converged);                                                           \
  }
  IPA_FUSED(2) IPA_FUSED(3) IPA_FUSED(4) IPA_FUSED(5) IPA_FUSED(6)
#undef IPA_FUSED
  return check_launch("k_ipa_sweep_fused");
}

extern "C" size_t ssspy_ipa_sweep_newton_words(int B, int N) {
  return (B > 0 && N > 0) ? (size_t)2 * B * N : 0;
}

extern "C" int ssspy_debug_barrier_timeouts(void) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ipa_barrier_timeouts), sizeof(int), 0,
                          hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  const int rows = ipa_rows_barrier_timeouts(), rt = ipa_rt_barrier_timeouts();
  return (rows < 0 || rt < 0) ? -1 : v + rows + rt;
}

static int lqpqm2_launch(const void *H, const void *v, const double *z, void *y, long long n, int L,
                         int max_iter, int floor_kind, double floor_eps, void *newton_ws,
                         int *not_converged, const int *singular, void *stream) {
  SSSPY_REQUIRE(H && v && z && y && n > 0 && max_iter >= 0, "lqpqm2: bad argument");
  if (L < 1 || L > SSSPY_RT_MAX_SOURCES - 1)
    return fail(SSSPY_ERR_UNSUPPORTED, "lqpqm2: dimension must be in [1, 15]");
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  hipStream_t st = as_stream(stream);
  unsigned long long *ws = (unsigned long long *)newton_ws;
  const bool exact = ws && max_iter >= 1 && max_iter <= 62;
  if (L > 7) {  // the dimension at run time (ipa_rt.hip)
    if (!exact)
      return lqpqm2_rt(NEWTON_FIXED, H, v, z, y, n, L, max_iter, floor_kind, floor_eps, ws, singular,
                       st);
    int rc = newton_prepare(ws, 1, st);
    if (rc) return rc;
    rc = lqpqm2_rt(NEWTON_PROBE, H, v, z, y, n, L, max_iter, floor_kind, floor_eps, ws, singular, st);
    if (rc) return rc;
    rc = newton_finish(ws, 1, max_iter, not_converged, st);
    if (rc) return rc;
    return lqpqm2_rt(NEWTON_APPLY, H, v, z, y, n, L, max_iter, floor_kind, floor_eps, ws, singular,
                     st);
  }
#define LQ_LAUNCH(L_, MODE_)                                                                     \
  hipLaunchKernelGGL((k_lqpqm2<L_, MODE_>), grid, block, 0, st, (const c128 *)H, (const c128 *)v, \
                     z, (c128 *)y, n, max_iter, floor_kind, floor_eps, ws, singular)
#define LQ_CASE(L_)                                                                              \
  if (L == L_) {                                                                                 \
    if (!exact) {                                                                                \
      LQ_LAUNCH(L_, NEWTON_FIXED);                                                               \
      return check_launch("k_lqpqm2");                                                           \
    }                                                                                            \
    int rc = newton_prepare(ws, 1, st);                                                          \
    if (rc) return rc;                                                                           \
    LQ_LAUNCH(L_, NEWTON_PROBE);                                                                 \
    rc = check_launch("k_lqpqm2 (probe)");                                                       \
    if (rc) return rc;                                                                           \
    rc = newton_finish(ws, 1, max_iter, not_converged, st);                                      \
    if (rc) return rc;                                                                           \
    LQ_LAUNCH(L_, NEWTON_APPLY);                                                                 \
    return check_launch("k_lqpqm2");                                                             \
  }
  LQ_CASE(1) LQ_CASE(2) LQ_CASE(3) LQ_CASE(4) LQ_CASE(5) LQ_CASE(6) LQ_CASE(7)
#undef LQ_CASE
#undef LQ_LAUNCH
  return SSSPY_ERR_UNSUPPORTED;
}

extern "C" int ssspy_lqpqm2(const void *H, const void *v, const double *z, void *y, long long n,
                            int L, int max_iter, int floor_kind, double floor_eps, void *newton_ws,
                            int *not_converged, void *stream) {
  return lqpqm2_launch(H, v, z, y, n, L, max_iter, floor_kind, floor_eps, newton_ws, not_converged,
                       nullptr, stream);
}

extern "C" int ssspy_lqpqm2_masked(const void *H, const void *v, const double *z, void *y,
                                   long long n, int L, int max_iter, int floor_kind,
                                   double floor_eps, void *newton_ws, int *not_converged,
                                   const int *singular, void *stream) {
  SSSPY_REQUIRE(singular, "lqpqm2_masked: the singular mask is required");
  return lqpqm2_launch(H, v, z, y, n, L, max_iter, floor_kind, floor_eps, newton_ws, not_converged,
                       singular, stream);
}
