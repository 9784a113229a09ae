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
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
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
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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

// (A + A^H) / 2 in place
__device__ void rt_hermitize(c128 *A, int N) {
  for (int a = 0; a < N; ++a) {
    A[a * N + a] = cmake(A[a * N + a].x, 0.0);
    for (int b = a + 1; b < N; ++b) {
      const c128 z = cmake(0.5 * (A[a * N + b].x + A[b * N + a].x),
                           0.5 * (A[a * N + b].y - A[b * N + a].y));
      A[a * N + b] = z;
      A[b * N + a] = cconj(z);
    }
  }
}

// lam_min(A) > shift by the pivots of the Cholesky factorisation of A - shift I (W: working copy)
__device__ bool rt_shifted_pd(const c128 *A, c128 *W, int N, double shift) {
  bool ok = true;
  for (int c = 0; c < N; ++c) {
    double d = A[c * N + c].x - shift;
    for (int k = 0; k < c; ++k) d -= cabs2(W[c * N + k]);
    ok = ok && (d > 0.0);
    const double il = 1.0 / sqrt(d > 0.0 ? d : 1.0);
    for (int r = c + 1; r < N; ++r) {
      c128 sum = A[r * N + c];
      for (int k = 0; k < c; ++k) cfms(sum, W[r * N + k], cconj(W[c * N + k]));
      W[r * N + c] = cscale(sum, il);
    }
  }
  return ok;
}

// cyclic complex Jacobi (the sweeps of jacobi_eigh, hermitian.hpp): A = P diag(lam) P^H, lam on the
// diagonal of A.  (The sweep loop ends when every lane that is in the call has converged.)
__device__ void rt_jacobi(c128 *A, c128 *P, int N) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) P[r * N + c] = cmake(r == c ? 1.0 : 0.0, 0.0);
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < N; ++p) {
      diag = fma(A[p * N + p].x, A[p * N + p].x, diag);
      for (int q = p + 1; q < N; ++q) off += cabs2(A[p * N + q]);
    }
    if (__all(off <= 1e-34 * diag)) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double app = A[p * N + p].x, aqq = A[q * N + q].x;
        const JacobiRot rot = jacobi_rot(A[p * N + q], app, aqq);
        const double cs = rot.cs;
        const c128 su = rot.su, sub = cconj(rot.su);
        for (int k = 0; k < N; ++k) {
          if (k != p && k != q) {
            const c128 akp = A[k * N + p], akq = A[k * N + q];
            c128 nkp = cmake(cs * akp.x, cs * akp.y);
            cfms(nkp, sub, akq);
            c128 nkq = cmake(cs * akq.x, cs * akq.y);
            cfma(nkq, su, akp);
            A[k * N + p] = nkp;
            A[p * N + k] = cconj(nkp);
            A[k * N + q] = nkq;
            A[q * N + k] = cconj(nkq);
          }
        }
        A[p * N + p] = cmake(app - rot.tm, 0.0);
        A[q * N + q] = cmake(aqq + rot.tm, 0.0);
        A[p * N + q] = cmake(0.0, 0.0);
        A[q * N + p] = cmake(0.0, 0.0);
        for (int k = 0; k < N; ++k) {
          const c128 vkp = P[k * N + p], vkq = P[k * N + q];
          c128 nkp = cmake(cs * vkp.x, cs * vkp.y);
          cfms(nkp, sub, vkq);
          c128 nkq = cmake(cs * vkq.x, cs * vkq.y);
          cfma(nkq, su, vkp);
          P[k * N + p] = nkp;
          P[k * N + q] = nkq;
        }
      }
  }
}

// to_psd: Hermitise, eigen-decompose, floor the eigenvalues (psd_eigen, hermitian.hpp)
__device__ void rt_psd_eigen(c128 *A, c128 *P, double *lam, int N, int floor_kind, double eps) {
  rt_hermitize(A, N);
  rt_jacobi(A, P, N);
  for (int k = 0; k < N; ++k) lam[k] = apply_floor(A[k * N + k].x, floor_kind, eps);
}

// Out = P diag(w) P^H (exactly Hermitian)
__device__ void rt_rebuild(const c128 *P, const double *w, c128 *Out, int N) {
  for (int a = 0; a < N; ++a)
    for (int b = a; b < N; ++b) {
      c128 s = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) {
        const c128 t = cmulc(P[a * N + k], P[b * N + k]);
        s.x = fma(w[k], t.x, s.x);
        s.y = fma(w[k], t.y, s.y);
      }
      if (a == b) s.y = 0.0;
      Out[a * N + b] = s;
      Out[b * N + a] = cconj(s);
    }
}

// Inverse of a Hermitian positive definite matrix by Cholesky (chol_inverse, hermitian.hpp): A is
// destroyed (its lower triangle becomes L), Li is scratch (L^-1).  False: a pivot was not positive.
__device__ bool rt_chol_inverse(c128 *A, c128 *Inv, c128 *Li, int N) {
  bool ok = true;
  for (int c = 0; c < N; ++c) {
    double d = A[c * N + c].x;
    for (int k = 0; k < c; ++k) d -= cabs2(A[c * N + k]);
    ok = ok && (d > 0.0);
    const double dd = d > 0.0 ? d : 1.0;
    const double l = sqrt(dd), il = 1.0 / l;
    A[c * N + c] = cmake(l, 0.0);
    for (int r = c + 1; r < N; ++r) {
      c128 s = A[r * N + c];
      for (int k = 0; k < c; ++k) cfms(s, A[r * N + k], cconj(A[c * N + k]));
      A[r * N + c] = cscale(s, il);
    }
  }
  for (int c = 0; c < N; ++c) {
    for (int r = 0; r < N; ++r) Li[r * N + c] = cmake(0.0, 0.0);
    Li[c * N + c] = cmake(1.0 / A[c * N + c].x, 0.0);
    for (int r = c + 1; r < N; ++r) {
      c128 s = cmake(0.0, 0.0);
      for (int k = c; k < r; ++k) cfms(s, A[r * N + k], Li[k * N + c]);
      Li[r * N + c] = cscale(s, 1.0 / A[r * N + r].x);
    }
  }
  for (int a = 0; a < N; ++a)
    for (int b = a; b < N; ++b) {
      c128 s = cmake(0.0, 0.0);
      for (int k = b; k < N; ++k) {
        const c128 t = cmulc(Li[k * N + b], Li[k * N + a]);  // conj(Li[k][a]) Li[k][b]
        s.x += t.x;
        s.y += t.y;
      }
      if (a == b) s.y = 0.0;
      Inv[a * N + b] = s;
      Inv[b * N + a] = cconj(s);
    }
  return ok;
}

// y = argmin of the LQPQM (type 2), H (L x L) Hermitian, v (L): lqpqm2 of ipa_kernels.hip with the
// NEWTON_FIXED (Vote = NoVoteRt) and NEWTON_FUSED (SweepVoteRt) modes.  sigma: scratch L x L.
template <bool FUSED, class Vote>
__device__ void rt_lqpqm2(c128 *H, c128 *sigma, const c128 *v, double z, int L, int floor_kind,
                          double eps, int max_iter, c128 *y, Vote vote) {
  rt_jacobi(H, sigma, L);
  double phi[RN], ph[RN], w2[RN];
  c128 vt[RN];
  for (int l = 0; l < L; ++l) phi[l] = H[l * L + l].x;
  const double f0 = floor_of_zero(floor_kind, eps);
  double vnorm2 = 0.0;
  for (int l = 0; l < L; ++l) vnorm2 += cabs2(v[l]);
  if (sqrt(vnorm2) < f0) {  // v = 0 (see lqpqm2: the reference's literal indexing)
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
    if (FUSED) vote(0ull, false);
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
    const int agreed = vote(bits, true);
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

}  // namespace

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
