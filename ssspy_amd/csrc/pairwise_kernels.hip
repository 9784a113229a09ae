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
// ---- the same update with a bin spread over G = 8 lanes, lane r owning row r of W (round 5; the
// layout of k_ip1_rows, spatial_kernels.hip).  The one-lane kernel above keeps W, two covariances and
// the LU working set of a bin in one lane's registers: 110 / 402 / 760 / 1438 spilled VGPRs at 5 / 6
// / 7 / 8 sources and 1.68 ms per launch for 16 x 1025 bins of 8 sources -- 44 % of a GaussILRMA-IP2
// iteration (profiles/r05_legs8_*).  Here a lane holds ONE row of W and of A = W U; the covariance
// entries are read where they are used (the lanes of a bin ask for the same address), the LU solve
// with the two right-hand sides e_m, e_n is row-distributed (pivot search by three exchange steps,
// the pivot row broadcast, back substitution broadcasting one pair of unknowns per step) and leaves
// the whole N x 2 solution in every lane; P^H U P is a sum over the rows in row order.
template <int N, int G>
__device__ __forceinline__ bool ip2_half_rows(const c128 (&Wr)[N], const c128 *__restrict__ Um,
                                              int r, int rr, bool row, int m, int n,
                                              c128 (&P)[N][2], c128 (&PUP)[2][2]) {
  c128 a[N];
#pragma unroll
  for (int c = 0; c < N; ++c) a[c] = cmake(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int c = 0; c < N; ++c) cfma(a[c], Wr[k], Um[k * N + c]);
  c128 rhs[2] = {cmake(r == m ? 1.0 : 0.0, 0.0), cmake(r == n ? 1.0 : 0.0, 0.0)};
  bool ok = true;
  int order = row ? -1 : N;  // elimination step at which my row became the pivot row
  int plane[N];              // lane of the group that owns pivot k (uniform within the group)
  c128 pinv[N];              // 1 / pivot k (crecip_fast), reused by the back substitution
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const bool cand = order < 0;
    double bv = cand ? cabs1(a[k]) : -1.0;
    int bl = r;
#pragma unroll
    for (int s = 1; s < G; s <<= 1) {
      const double ov = __shfl_xor(bv, s, G);
      const int ol = __shfl_xor(bl, s, G);
      const bool take = ov > bv || (ov == bv && ol < bl);
      bv = take ? ov : bv;
      bl = take ? ol : bl;
    }
    plane[k] = bl;
    if (r == bl) order = k;
    c128 prow[N];
#pragma unroll
    for (int c = k; c < N; ++c) prow[c] = cmake(__shfl(a[c].x, bl, G), __shfl(a[c].y, bl, G));
    const c128 prhs0 = cmake(__shfl(rhs[0].x, bl, G), __shfl(rhs[0].y, bl, G));
    const c128 prhs1 = cmake(__shfl(rhs[1].x, bl, G), __shfl(rhs[1].y, bl, G));
    const c128 piv = prow[k];
    ok = ok && (piv.x != 0.0 || piv.y != 0.0);
    pinv[k] = crecip_fast(piv);
    if (order < 0) {  // still unused: eliminate column k
      const c128 f = cmul(a[k], pinv[k]);
#pragma unroll
      for (int c = k + 1; c < N; ++c) cfms(a[c], f, prow[c]);
      cfms(rhs[0], f, prhs0);
      cfms(rhs[1], f, prhs1);
    }
  }
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    // the owner of pivot k has folded the unknowns above k into its right-hand sides already
    const c128 m0 = cmul(rhs[0], pinv[k]), m1 = cmul(rhs[1], pinv[k]);
    P[k][0] = cmake(__shfl(m0.x, plane[k], G), __shfl(m0.y, plane[k], G));
    P[k][1] = cmake(__shfl(m1.x, plane[k], G), __shfl(m1.y, plane[k], G));
    if (order < k) {
      cfms(rhs[0], a[k], P[k][0]);
      cfms(rhs[1], a[k], P[k][1]);
    }
  }
  // (U P)[row rr], then P^H (U P): lane rr contributes conj(P[rr][j]) (U P)[rr][k], added in row order
  c128 up[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)};
#pragma unroll
  for (int b = 0; b < N; ++b) {
    const c128 u = Um[rr * N + b];
    cfma(up[0], u, P[b][0]);
    cfma(up[1], u, P[b][1]);
  }
  c128 pmine[2] = {cmake(0.0, 0.0), cmake(0.0, 0.0)};
#pragma unroll
  for (int c = 0; c < N; ++c)
    if (c == rr) {
      pmine[0] = P[c][0];
      pmine[1] = P[c][1];
    }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double tx = row ? pmine[j].x * up[k].x + pmine[j].y * up[k].y : 0.0;
      const double ty = row ? pmine[j].x * up[k].y - pmine[j].y * up[k].x : 0.0;
      double sx = 0.0, sy = 0.0;
#pragma unroll
      for (int c = 0; c < N; ++c) {
        sx += __shfl(tx, c, G);
        sy += __shfl(ty, c, G);
      }
      PUP[j][k] = cmake(sx, sy);
    }
  return ok;
}

template <int N, int G>
__global__ __launch_bounds__(256) void k_ip2_rows(c128 *W, const c128 *__restrict__ U,
                                                  long long nbins, int pair_only, PairList pairs,
                                                  int floor_kind, double eps, int *info,
                                                  double *denom) {
  static_assert(G == 8 && N <= G, "one lane per row, groups of 8");
  const int r = threadIdx.x % G;
  const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const bool live = idx < nbins;
  const long long id = live ? idx : nbins - 1;  // idle groups shadow the last bin, never store
  const bool row = r < N;
  const int rr = row ? r : N - 1;
  c128 Wr[N];
#pragma unroll
  for (int c = 0; c < N; ++c) Wr[c] = W[id * (N * N) + rr * N + c];
  bool ok = true;
  const int u_sets = pair_only ? 2 : N;
#pragma unroll 1
  for (int p = 0; p < pairs.count; ++p) {
    const int m = pairs.first[p], n = pairs.second[p];
    const c128 *Um = U + (id * u_sets + (pair_only ? 0 : m)) * (N * N);
    const c128 *Un = U + (id * u_sets + (pair_only ? 1 : n)) * (N * N);
    c128 Pm[N][2], Pn[N][2], Gm[2][2], Gn[2][2];
    ok = ip2_half_rows<N, G>(Wr, Um, r, rr, row, m, n, Pm, Gm) && ok;
    ok = ip2_half_rows<N, G>(Wr, Un, r, rr, row, m, n, Pn, Gn) && ok;
    double lamb[2];
    c128 z[2][2];
    ok = eigh2_type1(Gm, Gn, lamb, z) && ok;
    c128 hm[2] = {z[0][1], z[1][1]}, hn[2] = {z[0][0], z[1][0]};
    double qm = quad2(hm, Gm), qn = quad2(hn, Gn);
    qm = qm < 0.0 ? 0.0 : qm;
    qn = qn < 0.0 ? 0.0 : qn;
    const double dm = denom ? 1.0 : apply_floor(sqrt(qm), floor_kind, eps);
    const double dn = denom ? 1.0 : apply_floor(sqrt(qn), floor_kind, eps);
    if (denom && live && r == 0) {
      denom[idx * 2 + 0] = sqrt(qm);
      denom[idx * 2 + 1] = sqrt(qn);
    }
    // every lane holds both solutions; rows m and n of W take conj(P h) / d
    if (r == m || r == n) {
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 v;
        if (r == m) {
          v = cmul(Pm[c][0], hm[0]);
          cfma(v, Pm[c][1], hm[1]);
          Wr[c] = cmake(v.x / dm, -v.y / dm);
        } else {
          v = cmul(Pn[c][0], hn[0]);
          cfma(v, Pn[c][1], hn[1]);
          Wr[c] = cmake(v.x / dn, -v.y / dn);
        }
      }
    }
  }
  if (live && row) {
#pragma unroll
    for (int c = 0; c < N; ++c) W[idx * (N * N) + r * N + c] = Wr[c];
    if (!ok && info && r == 0) atomicAdd(info, 1);
  }
}

// ---- ISS2 transform with a bin on GL lanes, lane s = source s = row s of G (round 5).  A lane
// per bin (the kernel of rounds 2-5, removed in round 6) ran 513 waves at 32 x 1025 bins -- half a wave per SIMD, every wave a serial walk
// over N sources per pair with N^2 uncoalesced loads each (88 us at 4 sources; 256 AGPRs + scratch
// at 8).  Here source s's statistics stay in lane s's registers (N <= 4) or are re-read by it
// (N > 4), the pair's rows and 2 x 2 blocks travel by lane shuffles, and every lane repeats the
// 2 x 2 eigenproblem.
template <int N, int GL, bool HOLD>
__global__ __launch_bounds__(256) void k_iss2_transform_rows(const c128 *__restrict__ Vc, c128 *G,
                                                             long long nbins, PairList pairs,
                                                             int floor_kind, double eps, int *info,
                                                             double *denom, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int s = lane & (GL - 1), base = lane & ~(GL - 1);
  const long long idx =
      ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / GL) + lane / GL;
  const bool live = idx < nbins;
  const long long bin = live ? idx : nbins - 1;
  const int sr = s < N ? s : N - 1;  // (idle lanes of a 3-source group shadow the last source)
  c128 gs[N];
#pragma unroll
  for (int c = 0; c < N; ++c)
    gs[c] = accumulate ? G[bin * (N * N) + sr * N + c] : cmake(c == sr ? 1.0 : 0.0, 0.0);
  const c128 *Vs = Vc + (bin * N + sr) * (long long)(N * N);
  c128 Vh[HOLD ? N : 1][HOLD ? N : 1];
  if (HOLD) {
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
      for (int d = 0; d < N; ++d) Vh[HOLD ? a : 0][HOLD ? d : 0] = Vs[a * N + d];
  }
  bool ok = true;
#pragma unroll 1
  for (int p = 0; p < pairs.count; ++p) {
    const int p0 = pairs.first[p], p1 = pairs.second[p];
    c128 g0[N], g1[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
      g0[c] = cmake(__shfl(gs[c].x, base + p0), __shfl(gs[c].y, base + p0));
      g1[c] = cmake(__shfl(gs[c].x, base + p1), __shfl(gs[c].y, base + p1));
    }
    c128 t0[N], t1[N], ts[N];
#pragma unroll
    for (int a = 0; a < N; ++a) {
      c128 a0 = cmake(0.0, 0.0), a1 = a0, a2 = a0;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const c128 u = HOLD ? Vh[HOLD ? a : 0][HOLD ? d : 0] : Vs[a * N + d];
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
    c128 Gmain[2][2][2];  // [k][a][b]: block of source p_k on the pair, from its lane
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        Gmain[0][a][b] = cmake(__shfl(C[a][b].x, base + p0), __shfl(C[a][b].y, base + p0));
        Gmain[1][a][b] = cmake(__shfl(C[a][b].x, base + p1), __shfl(C[a][b].y, base + p1));
      }
    double lamb[2];
    c128 z[2][2];
    ok = eigh2_type1(Gmain[0], Gmain[1], lamb, z) && ok;
    c128 row[N];
    if (s == p0 || s == p1) {
      const int k = (s == p0) ? 0 : 1;
      const c128 h[2] = {k == 0 ? z[0][0] : z[0][1], k == 0 ? z[1][0] : z[1][1]};
      double q = (k == 0) ? quad2(h, Gmain[0]) : quad2(h, Gmain[1]);
      q = q < 0.0 ? 0.0 : q;
      const double dk = denom ? 1.0 : apply_floor(sqrt(q), floor_kind, eps);
      if (denom && live) denom[idx * 2 + k] = sqrt(q);
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 v = cmul(cconj(h[0]), g0[c]);
        v = cadd(v, cmul(cconj(h[1]), g1[c]));
        row[c] = cmake(v.x / dk, v.y / dk);
      }
    } else {
      const c128 det = csub(cmul(C[0][0], C[1][1]), cmul(C[0][1], C[1][0]));
      const c128 idet = crecip(det);
      const c128 q0 = cmul(idet, csub(cmul(C[0][1], Fv[1]), cmul(C[1][1], Fv[0])));
      const c128 q1 = cmul(idet, csub(cmul(C[1][0], Fv[0]), cmul(C[0][0], Fv[1])));
#pragma unroll
      for (int c = 0; c < N; ++c) {
        c128 v = gs[c];
        v = cadd(v, cmul(cconj(q0), g0[c]));
        v = cadd(v, cmul(cconj(q1), g1[c]));
        row[c] = v;
      }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) gs[c] = row[c];
  }
  if (live && s < N) {
#pragma unroll
    for (int c = 0; c < N; ++c) G[idx * (N * N) + s * N + c] = gs[c];
    if (!ok && info && s == 0) atomicAdd(info, 1);
  }
}

template <int N>
static int launch_iss2_rows(const void *Vc, void *G, long long nbins, const PairList &pl,
                            int floor_kind, double eps, int *info, double *denom, int accumulate,
                            hipStream_t st) {
  constexpr int GL = N <= 2 ? 2 : (N <= 4 ? 4 : 8);
  constexpr bool HOLD = N <= 4;
  const long long per_block = 4 * (64 / GL);
  hipLaunchKernelGGL((k_iss2_transform_rows<N, GL, HOLD>),
                     dim3((unsigned)((nbins + per_block - 1) / per_block)), dim3(256), 0, st,
                     (const c128 *)Vc, (c128 *)G, nbins, pl, floor_kind, eps, info, denom,
                     accumulate);
  return check_launch("k_iss2_transform_rows");
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

// 5..8 sources: a bin on 8 lanes (k_ip2_rows)
static int launch_ip2_rows(void *W, const void *U, long long nbins, int N, int pair_only,
                           const PairList &pl, int floor_kind, double eps, int *info, double *denom,
                           hipStream_t st) {
  dim3 grid((unsigned)((nbins * 8 + 255) / 256)), block(256);
  switch (N) {
    case 5: hipLaunchKernelGGL((k_ip2_rows<5, 8>), grid, block, 0, st, (c128 *)W, (const c128 *)U, nbins, pair_only, pl, floor_kind, eps, info, denom); break;
    case 6: hipLaunchKernelGGL((k_ip2_rows<6, 8>), grid, block, 0, st, (c128 *)W, (const c128 *)U, nbins, pair_only, pl, floor_kind, eps, info, denom); break;
    case 7: hipLaunchKernelGGL((k_ip2_rows<7, 8>), grid, block, 0, st, (c128 *)W, (const c128 *)U, nbins, pair_only, pl, floor_kind, eps, info, denom); break;
    case 8: hipLaunchKernelGGL((k_ip2_rows<8, 8>), grid, block, 0, st, (c128 *)W, (const c128 *)U, nbins, pair_only, pl, floor_kind, eps, info, denom); break;
    default: return fail(SSSPY_ERR_INTERNAL, "k_ip2_rows: 5..8 sources");
  }
  return check_launch("k_ip2_rows");
}

// (per-N kernels of up to 4 sources: above, the row-distributed or the run-time-N forms take over)
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
  if (N > 4) return launch_ip2_rows(W, U, nbins, N, pair_only, pl, floor_kind, floor_eps, info,
                                    nullptr, as_stream(stream));
  DISPATCH_N4(N, hipLaunchKernelGGL((k_ip2<NN>), grid, block, 0, as_stream(stream), (c128 *)W,
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
  if (N > 4) return launch_ip2_rows(W, U, nbins, N, pair_only, pl, SSSPY_FLOOR_NONE, 0.0, info, denom,
                                    as_stream(stream));
  DISPATCH_N4(N, hipLaunchKernelGGL((k_ip2<NN>), grid, block, 0, as_stream(stream), (c128 *)W,
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
  DISPATCH_N(N, return launch_iss2_rows<NN>(Vc, G, nbins, pl, floor_kind, floor_eps, info, nullptr,
                                            0, as_stream(stream)));
  return SSSPY_OK;
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
  DISPATCH_N(N, return launch_iss2_rows<NN>(Vc, G, nbins, pl, SSSPY_FLOOR_NONE, 0.0, info, denom,
                                            accumulate ? 1 : 0, as_stream(stream)));
  return SSSPY_OK;
}

}  // extern "C"
