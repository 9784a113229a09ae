// Pairwise spatial updates: IP2 (two demixing rows at a time) and ISS2 (two steering vectors at a
// time), both built on a generalised 2x2 Hermitian eigenproblem per bin and pair.
// One lane owns one (mixture, bin); pairs are walked sequentially inside the lane.
#include "common.hpp"
#include "eigh2.hpp"
#include "rt_dense.hpp"
#include "smallmat.hpp"

namespace ssspy {

struct PairList {
  int count;
  int first[SSSPY_MAX_PAIRS];
  int second[SSSPY_MAX_PAIRS];
};

// row `idx` (runtime) of a register matrix without dynamic indexing
template <int N>
__device__ __forceinline__ void get_row(const Mat<N> &M, int idx, c128 (&row)[N]) {
#pragma unroll
  for (int c = 0; c < N; ++c) row[c] = cmake(0.0, 0.0);
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c)
      if (r == idx) row[c] = M.a[r][c];
}
template <int N>
__device__ __forceinline__ void set_row(Mat<N> &M, int idx, const c128 (&row)[N]) {
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int c = 0; c < N; ++c)
      if (r == idx) M.a[r][c] = row[c];
}

// P = (W U)^-1 [e_m e_n]  (N x 2);  PUP = P^H U P  (2 x 2)
template <int N>
__device__ __forceinline__ bool ip2_half(const Mat<N> &Wm, const Mat<N> &Um, int m, int n,
                                         c128 (&P)[N][2], c128 (&PUP)[2][2]) {
  Mat<N> A;
  matmul<N>(A, Wm, Um);
#pragma unroll
  for (int r = 0; r < N; ++r) {
    P[r][0] = cmake(r == m ? 1.0 : 0.0, 0.0);
    P[r][1] = cmake(r == n ? 1.0 : 0.0, 0.0);
  }
  const bool ok = lu_forward<N, 2>(A, P);
  lu_backward<N, 2>(A, P);
  c128 UP[N][2];
#pragma unroll
  for (int a = 0; a < N; ++a)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int b = 0; b < N; ++b) cfma(acc, Um.a[a][b], P[b][k]);
      UP[a][k] = acc;
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int a = 0; a < N; ++a) {  // conj(P[a][j]) * UP[a][k]
        acc.x += P[a][j].x * UP[a][k].x + P[a][j].y * UP[a][k].y;
        acc.y += P[a][j].x * UP[a][k].y - P[a][j].y * UP[a][k].x;
      }
      PUP[j][k] = acc;
    }
  return ok;
}

// ref: ssspy/bss/_update_spatial_model.py:81-143 (update_by_ip2), :317-395 (one pair).
// U holds N covariances per bin indexed by source (pair_only == 0), or the 2 covariances of the
// single pair being updated (pair_only != 0).
template <int N>
__global__ __launch_bounds__(64) void k_ip2(c128 *W, const c128 *__restrict__ U, long long nbins,
                                            int pair_only, PairList pairs, int floor_kind,
                                            double eps, int *info, double *denom) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Wm;
  load_mat<N>(Wm, W + idx * (N * N));
  bool ok = true;
#pragma unroll 1
  for (int p = 0; p < pairs.count; ++p) {
    const int m = pairs.first[p], n = pairs.second[p];
    Mat<N> Um, Un;
    const int u_sets = pair_only ? 2 : N;
    load_mat<N>(Um, U + (idx * u_sets + (pair_only ? 0 : m)) * (N * N));
    load_mat<N>(Un, U + (idx * u_sets + (pair_only ? 1 : n)) * (N * N));
    c128 Pm[N][2], Pn[N][2], Gm[2][2], Gn[2][2];
    ok = ip2_half<N>(Wm, Um, m, n, Pm, Gm) && ok;
    ok = ip2_half<N>(Wm, Un, m, n, Pn, Gn) && ok;
    double lamb[2];
    c128 z[2][2];
    ok = eigh2_type1(Gm, Gn, lamb, z) && ok;
    // the reference reverses the (ascending) eigenvector order: h_m <- larger eigenvalue
    c128 hm[2] = {z[0][1], z[1][1]}, hn[2] = {z[0][0], z[1][0]};
    double qm = quad2(hm, Gm), qn = quad2(hn, Gn);
    qm = qm < 0.0 ? 0.0 : qm;
    qn = qn < 0.0 ? 0.0 : qn;
    // denom (one pair per launch): the flooring callable runs on the host -- leave the rows
    // unnormalised, hand out sqrt(max(q, 0)) of both members, ssspy_scale_filter_row divides later
    const double dm = denom ? 1.0 : apply_floor(sqrt(qm), floor_kind, eps);
    const double dn = denom ? 1.0 : apply_floor(sqrt(qn), floor_kind, eps);
    if (denom) {
      denom[idx * 2 + 0] = sqrt(qm);
      denom[idx * 2 + 1] = sqrt(qn);
    }
    c128 wm[N], wn[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
      c128 a = cmul(Pm[r][0], hm[0]);
      cfma(a, Pm[r][1], hm[1]);
      c128 b = cmul(Pn[r][0], hn[0]);
      cfma(b, Pn[r][1], hn[1]);
      wm[r] = cmake(a.x / dm, -a.y / dm);  // conj(w_m)
      wn[r] = cmake(b.x / dn, -b.y / dn);
    }
    set_row<N>(Wm, m, wm);
    set_row<N>(Wm, n, wn);
  }
  store_mat<N>(Wm, W + idx * (N * N));
  if (!ok && info) atomicAdd(info, 1);
}

// ISS2 on per-bin statistics.  Vc[s] = (1/T) sum_j varphi_s y y^H of the CURRENT Y (N matrices per
// bin); with G the transform accumulated so far the statistics of the updated Y are G Vc[s] G^H,
// of which a pair step needs the 2x2 block on the pair for every s and, for the other sources, the
// pair's column against s.  ref: ssspy/bss/_update_spatial_model.py:197-314.
template <int N>
__global__ __launch_bounds__(64) void k_iss2_transform(const c128 *__restrict__ Vc, c128 *G,
                                                       long long nbins, PairList pairs,
                                                       int floor_kind, double eps, int *info,
                                                       double *denom, int accumulate) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<N> Gm;
  if (accumulate) load_mat<N>(Gm, G + idx * (N * N));  // continue from the transform so far
  else set_identity<N>(Gm);
  const c128 *V0 = Vc + idx * (long long)(N * N * N);
  bool ok = true;
#pragma unroll 1
  for (int p = 0; p < pairs.count; ++p) {
    const int p0 = pairs.first[p], p1 = pairs.second[p];
    c128 g0[N], g1[N];
    get_row<N>(Gm, p0, g0);
    get_row<N>(Gm, p1, g1);
    Mat<N> Gnew = Gm;
    c128 Gmain[2][2][2];  // [k][a][b]: block of source p_k on the pair
#pragma unroll 1
    for (int s = 0; s < N; ++s) {
      const c128 *Vs = V0 + s * N * N;
      c128 gs[N];
      get_row<N>(Gm, s, gs);
      // t_b = Vs g_b^H for b in {p0, p1, s}
      c128 t0[N], t1[N], ts[N];
#pragma unroll
      for (int a = 0; a < N; ++a) {
        c128 a0 = cmake(0.0, 0.0), a1 = a0, a2 = a0;
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const c128 u = Vs[a * N + d];
          a0 = cadd(a0, cmulc(u, g0[d]));
          a1 = cadd(a1, cmulc(u, g1[d]));
          a2 = cadd(a2, cmulc(u, gs[d]));
        }
        t0[a] = a0;
        t1[a] = a1;
        ts[a] = a2;
      }
      c128 C[2][2], Fv[2];
      C[0][0] = C[0][1] = C[1][0] = C[1][1] = Fv[0] = Fv[1] = cmake(0.0, 0.0);
#pragma unroll
      for (int a = 0; a < N; ++a) {
        cfma(C[0][0], g0[a], t0[a]);
        cfma(C[0][1], g0[a], t1[a]);
        cfma(C[1][0], g1[a], t0[a]);
        cfma(C[1][1], g1[a], t1[a]);
        cfma(Fv[0], g0[a], ts[a]);
        cfma(Fv[1], g1[a], ts[a]);
      }
      if (s == p0 || s == p1) {
        const int k = (s == p0) ? 0 : 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (k == 0) Gmain[0][a][b] = C[a][b];
            if (k == 1) Gmain[1][a][b] = C[a][b];
          }
      } else {
        // Q = -inv2(C) Fv ; row_s += conj(Q0) g0 + conj(Q1) g1
        const c128 det = csub(cmul(C[0][0], C[1][1]), cmul(C[0][1], C[1][0]));
        const c128 idet = crecip(det);
        const c128 q0 = cmul(idet, csub(cmul(C[0][1], Fv[1]), cmul(C[1][1], Fv[0])));
        const c128 q1 = cmul(idet, csub(cmul(C[1][0], Fv[0]), cmul(C[0][0], Fv[1])));
        c128 row[N];
#pragma unroll
        for (int c = 0; c < N; ++c) {
          c128 v = gs[c];
          v = cadd(v, cmul(cconj(q0), g0[c]));
          v = cadd(v, cmul(cconj(q1), g1[c]));
          row[c] = v;
        }
        set_row<N>(Gnew, s, row);
      }
    }
    double lamb[2];
    c128 z[2][2];
    ok = eigh2_type1(Gmain[0], Gmain[1], lamb, z) && ok;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const c128 h[2] = {z[0][k], z[1][k]};
      double q = (k == 0) ? quad2(h, Gmain[0]) : quad2(h, Gmain[1]);
      q = q < 0.0 ? 0.0 : q;
      // (denom: host-side flooring callable, see k_ip2)
      const double dk = denom ? 1.0 : apply_floor(sqrt(q), floor_kind, eps);
      if (denom) denom[idx * 2 + k] = sqrt(q);
      c128 row[N];
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 v = cmul(cconj(h[0]), g0[c]);
        v = cadd(v, cmul(cconj(h[1]), g1[c]));
        row[c] = cmake(v.x / dk, v.y / dk);
      }
      set_row<N>(Gnew, k == 0 ? p0 : p1, row);
    }
    Gm = Gnew;
  }
  store_mat<N>(Gm, G + idx * (N * N));
  if (!ok && info) atomicAdd(info, 1);
}

// ---- the same two updates with the source count at run time (9 <= N <= SSSPY_RT_MAX_SOURCES):
// loops instead of unrolled code, the bin's matrices in the lane's private memory (round 4; the
// reference has no limit on n_sources).  Arithmetic as above.
// P = (W U)^-1 [e_m e_n] (N x 2), PUP = P^H U P (2 x 2)
__device__ inline bool ip2_half_rt(const c128 *Wm, const c128 *__restrict__ Um, int N, int m, int n,
                                   c128 *A, c128 *P, c128 (&PUP)[2][2]) {
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c) {
      c128 acc = cmake(0.0, 0.0);
      for (int k = 0; k < N; ++k) cfma(acc, Wm[r * N + k], Um[k * N + c]);
      A[r * N + c] = acc;
    }
  for (int r = 0; r < N; ++r) {
    P[r * 2 + 0] = cmake(r == m ? 1.0 : 0.0, 0.0);
    P[r * 2 + 1] = cmake(r == n ? 1.0 : 0.0, 0.0);
  }
  const bool ok = rt_lu_solve(A, P, N, 2);
  // UP = U P in A's first 2 N slots (A is free now)
  c128 *UP = A;
  for (int a = 0; a < N; ++a)
    for (int k = 0; k < 2; ++k) {
      c128 acc = cmake(0.0, 0.0);
      for (int b = 0; b < N; ++b) cfma(acc, Um[a * N + b], P[b * 2 + k]);
      UP[a * 2 + k] = acc;
    }
  for (int j = 0; j < 2; ++j)
    for (int k = 0; k < 2; ++k) {
      c128 acc = cmake(0.0, 0.0);
      for (int a = 0; a < N; ++a) {  // conj(P[a][j]) * UP[a][k]
        acc.x += P[a * 2 + j].x * UP[a * 2 + k].x + P[a * 2 + j].y * UP[a * 2 + k].y;
        acc.y += P[a * 2 + j].x * UP[a * 2 + k].y - P[a * 2 + j].y * UP[a * 2 + k].x;
      }
      PUP[j][k] = acc;
    }
  return ok;
}

__global__ __launch_bounds__(64) void k_ip2_rt(c128 *W, const c128 *__restrict__ U, long long nbins,
                                               int N, int pair_only, PairList pairs, int floor_kind,
                                               double eps, int *info, double *denom) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 Wm[RTN * RTN], A[RTN * RTN], Pm[RTN * 2], Pn[RTN * 2];
  for (int e = 0; e < N * N; ++e) Wm[e] = W[idx * (N * N) + e];
  bool ok = true;
  for (int p = 0; p < pairs.count; ++p) {
    const int m = pairs.first[p], n = pairs.second[p];
    const int u_sets = pair_only ? 2 : N;
    const c128 *__restrict__ Um = U + (idx * u_sets + (pair_only ? 0 : m)) * (long long)(N * N);
    const c128 *__restrict__ Un = U + (idx * u_sets + (pair_only ? 1 : n)) * (long long)(N * N);
    c128 Gm[2][2], Gn[2][2];
    ok = ip2_half_rt(Wm, Um, N, m, n, A, Pm, Gm) && ok;
    ok = ip2_half_rt(Wm, Un, N, m, n, A, Pn, Gn) && ok;
    double lamb[2];
    c128 z[2][2];
    ok = eigh2_type1(Gm, Gn, lamb, z) && ok;
    c128 hm[2] = {z[0][1], z[1][1]}, hn[2] = {z[0][0], z[1][0]};
    double qm = quad2(hm, Gm), qn = quad2(hn, Gn);
    qm = qm < 0.0 ? 0.0 : qm;
    qn = qn < 0.0 ? 0.0 : qn;
    const double dm = denom ? 1.0 : apply_floor(sqrt(qm), floor_kind, eps);
    const double dn = denom ? 1.0 : apply_floor(sqrt(qn), floor_kind, eps);
    if (denom) {
      denom[idx * 2 + 0] = sqrt(qm);
      denom[idx * 2 + 1] = sqrt(qn);
    }
    for (int r = 0; r < N; ++r) {
      c128 a = cmul(Pm[r * 2 + 0], hm[0]);
      cfma(a, Pm[r * 2 + 1], hm[1]);
      c128 b = cmul(Pn[r * 2 + 0], hn[0]);
      cfma(b, Pn[r * 2 + 1], hn[1]);
      Wm[m * N + r] = cmake(a.x / dm, -a.y / dm);  // conj(w_m)
      Wm[n * N + r] = cmake(b.x / dn, -b.y / dn);
    }
  }
  for (int e = 0; e < N * N; ++e) W[idx * (N * N) + e] = Wm[e];
  if (!ok && info) atomicAdd(info, 1);
}

__global__ __launch_bounds__(64) void k_iss2_transform_rt(const c128 *__restrict__ Vc, c128 *G,
                                                          long long nbins, int N, PairList pairs,
                                                          int floor_kind, double eps, int *info,
                                                          double *denom, int accumulate) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  c128 Gm[RTN * RTN], Gnew[RTN * RTN], t0[RTN], t1[RTN], ts[RTN];
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < N; ++c)
      Gm[r * N + c] = accumulate ? G[idx * (N * N) + r * N + c] : cmake(r == c ? 1.0 : 0.0, 0.0);
  const c128 *__restrict__ V0 = Vc + idx * (long long)(N * N) * N;
  bool ok = true;
  for (int p = 0; p < pairs.count; ++p) {
    const int p0 = pairs.first[p], p1 = pairs.second[p];
    const c128 *g0 = Gm + p0 * N, *g1 = Gm + p1 * N;
    for (int e = 0; e < N * N; ++e) Gnew[e] = Gm[e];
    c128 Gmain[2][2][2];
    for (int s = 0; s < N; ++s) {
      const c128 *__restrict__ Vs = V0 + (long long)s * N * N;
      const c128 *gs = Gm + s * N;
      for (int a = 0; a < N; ++a) {
        c128 a0 = cmake(0.0, 0.0), a1 = a0, a2 = a0;
        for (int d = 0; d < N; ++d) {
          const c128 u = Vs[a * N + d];
          a0 = cadd(a0, cmulc(u, g0[d]));
          a1 = cadd(a1, cmulc(u, g1[d]));
          a2 = cadd(a2, cmulc(u, gs[d]));
        }
        t0[a] = a0;
        t1[a] = a1;
        ts[a] = a2;
      }
      c128 C[2][2], Fv[2];
      C[0][0] = C[0][1] = C[1][0] = C[1][1] = Fv[0] = Fv[1] = cmake(0.0, 0.0);
      for (int a = 0; a < N; ++a) {
        cfma(C[0][0], g0[a], t0[a]);
        cfma(C[0][1], g0[a], t1[a]);
        cfma(C[1][0], g1[a], t0[a]);
        cfma(C[1][1], g1[a], t1[a]);
        cfma(Fv[0], g0[a], ts[a]);
        cfma(Fv[1], g1[a], ts[a]);
      }
      if (s == p0 || s == p1) {
        const int k = (s == p0) ? 0 : 1;
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 2; ++b) Gmain[k][a][b] = C[a][b];
      } else {
        const c128 det = csub(cmul(C[0][0], C[1][1]), cmul(C[0][1], C[1][0]));
        const c128 idet = crecip(det);
        const c128 q0 = cmul(idet, csub(cmul(C[0][1], Fv[1]), cmul(C[1][1], Fv[0])));
        const c128 q1 = cmul(idet, csub(cmul(C[1][0], Fv[0]), cmul(C[0][0], Fv[1])));
        for (int c = 0; c < N; ++c) {
          c128 v = gs[c];
          v = cadd(v, cmul(cconj(q0), g0[c]));
          v = cadd(v, cmul(cconj(q1), g1[c]));
          Gnew[s * N + c] = v;
        }
      }
    }
    double lamb[2];
    c128 z[2][2];
    ok = eigh2_type1(Gmain[0], Gmain[1], lamb, z) && ok;
    for (int k = 0; k < 2; ++k) {
      const c128 h[2] = {z[0][k], z[1][k]};
      double q = (k == 0) ? quad2(h, Gmain[0]) : quad2(h, Gmain[1]);
      q = q < 0.0 ? 0.0 : q;
      const double dk = denom ? 1.0 : apply_floor(sqrt(q), floor_kind, eps);
      if (denom) denom[idx * 2 + k] = sqrt(q);
      for (int c = 0; c < N; ++c) {
        c128 v = cmul(cconj(h[0]), g0[c]);
        v = cadd(v, cmul(cconj(h[1]), g1[c]));
        Gnew[(k == 0 ? p0 : p1) * N + c] = cmake(v.x / dk, v.y / dk);
      }
    }
    for (int e = 0; e < N * N; ++e) Gm[e] = Gnew[e];
  }
  for (int e = 0; e < N * N; ++e) G[idx * (N * N) + e] = Gm[e];
  if (!ok && info) atomicAdd(info, 1);
}

static bool pair_rt_sources(int N) { return N > SSSPY_MAX_SOURCES && N <= SSSPY_RT_MAX_SOURCES; }

static int fill_pairs(PairList &pl, const int *pairs, int n_pairs, int N) {
  if (n_pairs < 1 || n_pairs > SSSPY_MAX_PAIRS)
    return fail(SSSPY_ERR_BADARG, "pair list must hold between 1 and SSSPY_MAX_PAIRS pairs");
  pl.count = n_pairs;
  for (int p = 0; p < n_pairs; ++p) {
    const int m = pairs[2 * p], n = pairs[2 * p + 1];
    if (m < 0 || m >= N || n < 0 || n >= N || m == n)
      return fail(SSSPY_ERR_BADARG, "pair indices must be distinct and in [0, n_sources)");
    pl.first[p] = m;
    pl.second[p] = n;
  }
  return SSSPY_OK;
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

int ssspy_update_by_ip2(void *W, const void *U, int pair_only, const int *pairs, int n_pairs, int B,
                        int F, int N, int floor_kind, double floor_eps, int *info, void *stream) {
  SSSPY_REQUIRE(W && U && pairs && B > 0 && F > 0, "update_by_ip2: bad argument");
  SSSPY_REQUIRE(!pair_only || n_pairs == 1,
                "update_by_ip2: a pair-only U (B,F,2,N,N) serves exactly one pair");
  PairList pl;
  int rc = fill_pairs(pl, pairs, n_pairs, N);
  if (rc) return rc;
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  if (pair_rt_sources(N)) {
    hipLaunchKernelGGL(k_ip2_rt, grid, block, 0, as_stream(stream), (c128 *)W, (const c128 *)U, nbins,
                       N, pair_only, pl, floor_kind, floor_eps, info, (double *)nullptr);
    return check_launch("k_ip2_rt");
  }
  DISPATCH_N(N, hipLaunchKernelGGL((k_ip2<NN>), grid, block, 0, as_stream(stream), (c128 *)W,
                                   (const c128 *)U, nbins, pair_only, pl, floor_kind, floor_eps, info,
                                   (double *)nullptr));
  return check_launch("k_ip2");
}

int ssspy_update_by_ip2_deferred(void *W, const void *U, int pair_only, const int *pair, int B,
                                 int F, int N, double *denom, int *info, void *stream) {
  SSSPY_REQUIRE(W && U && pair && denom && B > 0 && F > 0, "update_by_ip2_deferred: bad argument");
  PairList pl;
  int rc = fill_pairs(pl, pair, 1, N);
  if (rc) return rc;
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  if (pair_rt_sources(N)) {
    hipLaunchKernelGGL(k_ip2_rt, grid, block, 0, as_stream(stream), (c128 *)W, (const c128 *)U, nbins,
                       N, pair_only, pl, SSSPY_FLOOR_NONE, 0.0, info, denom);
    return check_launch("k_ip2_rt (deferred)");
  }
  DISPATCH_N(N, hipLaunchKernelGGL((k_ip2<NN>), grid, block, 0, as_stream(stream), (c128 *)W,
                                   (const c128 *)U, nbins, pair_only, pl, SSSPY_FLOOR_NONE, 0.0, info,
                                   denom));
  return check_launch("k_ip2 (deferred)");
}

int ssspy_iss2_transform(const void *Vc, void *G, const int *pairs, int n_pairs, int B, int F,
                         int N, int floor_kind, double floor_eps, int *info, void *stream) {
  SSSPY_REQUIRE(Vc && G && pairs && B > 0 && F > 0, "iss2_transform: bad argument");
  PairList pl;
  int rc = fill_pairs(pl, pairs, n_pairs, N);
  if (rc) return rc;
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  if (pair_rt_sources(N)) {
    hipLaunchKernelGGL(k_iss2_transform_rt, grid, block, 0, as_stream(stream), (const c128 *)Vc,
                       (c128 *)G, nbins, N, pl, floor_kind, floor_eps, info, (double *)nullptr, 0);
    return check_launch("k_iss2_transform_rt");
  }
  DISPATCH_N(N, hipLaunchKernelGGL((k_iss2_transform<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)Vc, (c128 *)G, nbins, pl, floor_kind, floor_eps,
                                   info, (double *)nullptr, 0));
  return check_launch("k_iss2_transform");
}

int ssspy_iss2_transform_deferred(const void *Vc, void *G, const int *pair, int accumulate, int B,
                                  int F, int N, double *denom, int *info, void *stream) {
  SSSPY_REQUIRE(Vc && G && pair && denom && B > 0 && F > 0, "iss2_transform_deferred: bad argument");
  PairList pl;
  int rc = fill_pairs(pl, pair, 1, N);
  if (rc) return rc;
  const long long nbins = (long long)B * F;
  dim3 grid((unsigned)((nbins + 63) / 64)), block(64);
  if (pair_rt_sources(N)) {
    hipLaunchKernelGGL(k_iss2_transform_rt, grid, block, 0, as_stream(stream), (const c128 *)Vc,
                       (c128 *)G, nbins, N, pl, SSSPY_FLOOR_NONE, 0.0, info, denom,
                       accumulate ? 1 : 0);
    return check_launch("k_iss2_transform_rt (deferred)");
  }
  DISPATCH_N(N, hipLaunchKernelGGL((k_iss2_transform<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)Vc, (c128 *)G, nbins, pl, SSSPY_FLOOR_NONE, 0.0,
                                   info, denom, accumulate ? 1 : 0));
  return check_launch("k_iss2_transform (deferred)");
}

}  // extern "C"
