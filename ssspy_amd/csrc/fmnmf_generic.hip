// FastGaussMNMF, general shapes: any n_channels M in [2, 8] and n_sources N in [1, 8].
//
// The MFMA-tile kernels of mnmf_kernels.hip keep the diagonaliser rows, the spatial rows and the
// GEMM1 output of every source in registers and are compiled for N, M <= 4.  Beyond that the unit of
// work here is "one lane = one (bin, frame) point" (the GaussMNMF design, gmnmf_kernels.hip): the
// lane forms lambda_n, |q_m^H x|^2 and R~_m = sum_n lambda_n d_nm and leaves
//   traces   A_n = sum_m d_nm |q_m^H x|^2 / R~_m^2,  Bt_n = sum_m d_nm / R~_m      (basis, activation)
//   weights  1 / R~_m                                                           (diagonaliser: IP1)
//   loss     sum_m |q_m^H x|^2 / R~_m + log R~_m
// in HBM; small kernels contract them with the activation (over frames) or the basis (over bins), and
// the spatial update folds its N x M sums per bin through LDS.  Per iteration that is five passes over
// X instead of four plus the (N, F, T) traces: a correct general path, not a tuned one.
//
// replaces: ssspy/bss/mnmf.py:1278-1303 (update_once), :1305-1417, :1449-1514, :1635-1675, :632-678,
//           :1219-1261 (loss), :1174-1217 (Wiener filter) for shapes outside N, M <= 4.
#include "common.hpp"
#include "hermitian.hpp"
#include "smallmat.hpp"
#include "ssspy_amd.h"

namespace ssspy {

int ip1_with_power(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N,
                   int floor_kind, double floor_eps, int *info, hipStream_t st);
int row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st);

namespace fmg {

constexpr int NMAX = SSSPY_MAX_SOURCES;
enum { MODE_TRACES = 0, MODE_WEIGHTS = 1, MODE_LOSS = 2 };

// Q, D and basis rows of one bin in LDS: Qs[M*M], Ds[N*M], Ts[N*K]
template <int M>
__device__ __forceinline__ void stage_bin(c128 *Qs, double *Ds, double *Ts,
                                          const c128 *__restrict__ Q,
                                          const double *__restrict__ Dsp,
                                          const double *__restrict__ basis, int b, int N, int F,
                                          int K, int i) {
  for (int e = threadIdx.x; e < M * M; e += blockDim.x) Qs[e] = Q[((long long)b * F + i) * (M * M) + e];
  for (int e = threadIdx.x; e < N * M; e += blockDim.x) Ds[e] = Dsp[((long long)b * F + i) * (N * M) + e];
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int n = e / K, k = e % K;
    Ts[e] = basis[(((long long)b * N + n) * F + i) * K + k];
  }
  __syncthreads();
}

static inline size_t bin_smem(int N, int M, int K) {
  return (size_t)M * M * sizeof(c128) + ((size_t)N * M + (size_t)N * K) * sizeof(double);
}

// per-point terms: lam[n], qx2[m] = |(Q x)_m|^2, rc[m] = R~_m
template <int M>
__device__ __forceinline__ void point_terms(double (&lam)[NMAX], double (&qx2)[M], double (&rc)[M],
                                            const c128 *__restrict__ Xb,
                                            const double *__restrict__ act_b, const c128 *Qs,
                                            const double *Ds, const double *Ts, int N, int F, int T,
                                            int K, int i, int j) {
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    double l = 0.0;
    if (n < N)
      for (int k = 0; k < K; ++k) l = fma(Ts[n * K + k], act_b[((long long)n * K + k) * T + j], l);
    lam[n] = l;
  }
  c128 x[M];
#pragma unroll
  for (int m = 0; m < M; ++m) x[m] = Xb[((long long)m * F + i) * T + j];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    c128 y = cmake(0.0, 0.0);
#pragma unroll
    for (int a = 0; a < M; ++a) cfma(y, Qs[m * M + a], x[a]);
    qx2[m] = cabs2(y);
    double r = 0.0;
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) r = fma(lam[n], Ds[n * M + m], r);
    rc[m] = r;
  }
}

// grid: (ceil(T/128), F, B), 128 threads (lanes along frames)
template <int M, int MODE>
__global__ __launch_bounds__(128) void k_points(const c128 *__restrict__ X,
                                                const c128 *__restrict__ Q,
                                                const double *__restrict__ Dsp,
                                                const double *__restrict__ basis,
                                                const double *__restrict__ act,
                                                double *__restrict__ out0,
                                                double *__restrict__ out1, int N, int F, int T,
                                                int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Qs = reinterpret_cast<c128 *>(smem);
  double *Ds = reinterpret_cast<double *>(Qs + M * M);
  double *Ts = Ds + N * M;
  __shared__ double scratch[2];
  const int i = blockIdx.y, b = blockIdx.z;
  stage_bin<M>(Qs, Ds, Ts, Q, Dsp, basis, b, N, F, K, i);
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = j < T;
  double lam[NMAX], qx2[M], rc[M];
  point_terms<M>(lam, qx2, rc, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Qs, Ds,
                 Ts, N, F, T, K, i, valid ? j : T - 1);
  if (MODE == MODE_TRACES) {
    if (!valid) return;
    double g[M], h[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      g[m] = 1.0 / rc[m];
      h[m] = qx2[m] * g[m] * g[m];
    }
    for (int n = 0; n < N; ++n) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int m = 0; m < M; ++m) {
        sa = fma(Ds[n * M + m], h[m], sa);
        sb = fma(Ds[n * M + m], g[m], sb);
      }
      const long long o = (((long long)b * N + n) * F + i) * T + j;
      out0[o] = sa;
      out1[o] = sb;
    }
  } else if (MODE == MODE_WEIGHTS) {
    if (!valid) return;
#pragma unroll
    for (int m = 0; m < M; ++m) out0[(((long long)b * M + m) * F + i) * T + j] = 1.0 / rc[m];
  } else {
    double term = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) term += qx2[m] / rc[m] + log(rc[m]);
    term = wave_sum(valid ? term : 0.0);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = term;
    __syncthreads();
    // one slot per (frame block, bin) of the mixture, [slot][B]; fmnmf_generic_loss folds them
    if (threadIdx.x == 0)
      out0[((long long)blockIdx.y * gridDim.x + blockIdx.x) * gridDim.z + b] =
          (scratch[0] + scratch[1]) / (double)T;
  }
}

// basis[b,n,i,k] <- floor(basis * sqrt(sum_j V A / sum_j V Bt)).  grid: (F, N, B), 256 threads
__global__ __launch_bounds__(256) void k_basis(double *basis, const double *__restrict__ act,
                                               const double *__restrict__ A,
                                               const double *__restrict__ Bt, int N, int F, int T,
                                               int K, int floor_kind, double eps) {
  const int i = blockIdx.x, n = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row = (((long long)b * N + n) * F + i) * T;
  for (int k = wave; k < K; k += 4) {
    const double *v = act + (((long long)b * N + n) * K + k) * T;
    double sn = 0.0, sd = 0.0;
    for (int j = lane; j < T; j += 64) {
      const double vv = v[j];
      sn = fma(vv, A[row + j], sn);
      sd = fma(vv, Bt[row + j], sd);
    }
    sn = wave_sum(sn);
    sd = wave_sum(sd);
    if (lane == 0) {
      const long long o = (((long long)b * N + n) * F + i) * K + k;
      basis[o] = apply_floor(basis[o] * sqrt(sn / sd), floor_kind, eps);
    }
  }
}

// act[b,n,k,j] <- floor(act * sqrt(sum_i T A / sum_i T Bt)): lanes along frames, each lane walks all
// bins for its (n, k) -- deterministic, no atomics.  grid: (ceil(T/256), K, N*B)
__global__ __launch_bounds__(256) void k_activation(const double *__restrict__ basis, double *act,
                                                    const double *__restrict__ A,
                                                    const double *__restrict__ Bt, int N, int F,
                                                    int T, int K, int floor_kind, double eps) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  const int n = blockIdx.z % N, b = blockIdx.z / N;
  if (j >= T) return;
  const double *tb = basis + ((long long)b * N + n) * F * K + k;
  const long long base = ((long long)b * N + n) * F * T + j;
  double sn = 0.0, sd = 0.0;
  for (int i = 0; i < F; ++i) {
    const double t = tb[(long long)i * K];
    sn = fma(t, A[base + (long long)i * T], sn);
    sd = fma(t, Bt[base + (long long)i * T], sd);
  }
  double *dst = act + (((long long)b * N + n) * K + k) * T + j;
  *dst = apply_floor((*dst) * sqrt(sn / sd), floor_kind, eps);
}

// d_inm <- d_inm sqrt(sum_j lam_n h_m / sum_j lam_n g_m) (no floor).  grid: (F, B), 64 threads: lanes
// take frames in chunks of 64, the per-point (lam, g, h) go through LDS and thread e folds the chunk
// into its (n, m) sums.
constexpr int PB = 64;
template <int M>
__global__ __launch_bounds__(PB) void k_spatial(const c128 *__restrict__ X,
                                                const c128 *__restrict__ Q, double *Dsp,
                                                const double *__restrict__ basis,
                                                const double *__restrict__ act, int N, int F, int T,
                                                int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *Qs = reinterpret_cast<c128 *>(smem);
  double *Ds = reinterpret_cast<double *>(Qs + M * M);
  double *Ts = Ds + N * M;
  constexpr int ROW = NMAX + 2 * M;  // lam[NMAX], g[M], h[M] per point
  double *pts = Ts + N * K;          // [PB][ROW]
  const int i = blockIdx.x, b = blockIdx.y;
  stage_bin<M>(Qs, Ds, Ts, Q, Dsp, basis, b, N, F, K, i);
  constexpr int SLOTS = (NMAX * M + PB - 1) / PB;  // (n, m) pairs per thread
  double an[SLOTS], ad[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) an[s] = ad[s] = 0.0;
  for (int j0 = 0; j0 < T; j0 += PB) {
    const int j = j0 + threadIdx.x;
    double lam[NMAX], qx2[M], rc[M];
    point_terms<M>(lam, qx2, rc, X + (long long)b * M * F * T, act + (long long)b * N * K * T, Qs,
                   Ds, Ts, N, F, T, K, i, j < T ? j : T - 1);
    double *mine = pts + threadIdx.x * ROW;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) mine[n] = j < T ? lam[n] : 0.0;  // frames beyond T add nothing
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const double g = 1.0 / rc[m];
      mine[NMAX + m] = g;
      mine[NMAX + M + m] = qx2[m] * g * g;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int idx = threadIdx.x + PB * s;
      if (idx < N * M) {
        const int n = idx / M, m = idx % M;
        double vn = an[s], vd = ad[s];
        for (int p = 0; p < PB; ++p) {
          const double l = pts[p * ROW + n];
          vn = fma(l, pts[p * ROW + NMAX + M + m], vn);
          vd = fma(l, pts[p * ROW + NMAX + m], vd);
        }
        an[s] = vn;
        ad[s] = vd;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int idx = threadIdx.x + PB * s;
    if (idx < N * M) {
      double *dst = Dsp + ((long long)b * F + i) * (N * M) + idx;
      *dst = sqrt(an[s] / ad[s]) * Ds[idx];
    }
  }
}

// psi_m = floor(sqrt(mean_i q[i][m])); Q[:,m,:] /= psi_m; D[:,:,m] /= psi_m^2.  grid (ceil(F/64), B)
__global__ __launch_bounds__(256) void k_norm_scale(c128 *Q, double *Dsp,
                                                    const double *__restrict__ qbuf, int N, int M,
                                                    int F, int floor_kind, double eps) {
  __shared__ double scratch[4];
  __shared__ double psi[NMAX];
  const int b = blockIdx.y;
  const double *qb = qbuf + (long long)b * F * M;
  for (int m = 0; m < M; ++m) {
    double local = 0.0;
    for (int i = threadIdx.x; i < F; i += blockDim.x) local += qb[(long long)i * M + m];
    const double total = block_sum(local, scratch);
    if (threadIdx.x == 0) {
      double v = total / (double)F;
      v = v < 0.0 ? 0.0 : v;
      psi[m] = apply_floor(sqrt(v), floor_kind, eps);
    }
  }
  __syncthreads();
  const int i0 = blockIdx.x * 64;
  const int nb = min(64, F - i0);
  c128 *Qb = Q + ((long long)b * F + i0) * M * M;
  for (int e = threadIdx.x; e < nb * M * M; e += blockDim.x) {
    const int m = (e / M) % M;
    const c128 v = Qb[e];
    Qb[e] = cmake(v.x / psi[m], v.y / psi[m]);
  }
  double *Db = Dsp + ((long long)b * F + i0) * N * M;
  for (int e = threadIdx.x; e < nb * N * M; e += blockDim.x) {
    const int m = e % M;
    Db[e] = Db[e] / (psi[m] * psi[m]);
  }
}

template <int M>
__global__ __launch_bounds__(64) void k_qinv(const c128 *__restrict__ Q, c128 *Qinv,
                                             long long nbins, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nbins) return;
  Mat<M> A, Inv;
  load_mat<M>(A, Q + idx * (M * M));
  const bool ok = invert<M>(A, Inv);
  store_mat<M>(Inv, Qinv + idx * (M * M));
  if (!ok && info) atomicAdd(info, 1);
}

// Multichannel Wiener filter (ref: ssspy/bss/mnmf.py:1174-1217): grid (F, B), lanes along frames.
// R = Q~ diag(rc) Q~^H, Q~ = Q^-1; closed form R^-1 = Q^H diag(1/rc) Q when the eigenvalue floor is
// provably idle (lambda_min(R) >= min rc / ||Q||_F^2 > eps), else the Jacobi eigen-floor.
// Round 5: two launches of it.  REPAIR == false holds the closed form alone (no M x M working set: 5-8
// channels no longer spill 530-2400 registers per lane on the path every call takes) and flags the
// bins that have a point where the floor may act; REPAIR == true is the old kernel restricted to the
// flagged bins (both forms per point, as before).
// EIG (an arbitrary flooring callable, evaluated on the host -- round 5): 1 = every point takes the
// literal route up to its eigen-decomposition and leaves the eigenvalues (ascending, as numpy.linalg.
// eigh hands them to the callable: lam_io (B, F, T, M)) and eigenvectors (P_io (B, F, T, M, M)) in HBM;
// 2 = the same walk from the floored eigenvalues and the stored eigenvectors to Y.
template <int M, bool REPAIR, int EIG = 0>
__global__ __launch_bounds__(128) void k_separate(const c128 *__restrict__ X,
                                                  const c128 *__restrict__ Q,
                                                  const c128 *__restrict__ Qinv,
                                                  const double *__restrict__ Dsp,
                                                  const double *__restrict__ basis,
                                                  const double *__restrict__ act, c128 *Y, int N,
                                                  int F, int T, int K, int ref, int floor_kind,
                                                  double eps, int *redo, double *lam_io = nullptr,
                                                  c128 *P_io = nullptr) {
  const int i = blockIdx.x, b = blockIdx.y;
  if (REPAIR && EIG == 0 && !redo[(long long)b * F + i]) return;
  __shared__ c128 qt[M * M];
  __shared__ c128 qsrc[M * M];
  __shared__ double dd[NMAX * M];
  if (threadIdx.x < M * M) {
    qt[threadIdx.x] = Qinv[((long long)b * F + i) * (M * M) + threadIdx.x];
    qsrc[threadIdx.x] = Q[((long long)b * F + i) * (M * M) + threadIdx.x];
  }
  for (int e = threadIdx.x; e < N * M; e += blockDim.x)
    dd[e] = Dsp[((long long)b * F + i) * (N * M) + e];
  __syncthreads();
  double qf2 = 0.0;
#pragma unroll
  for (int e = 0; e < M * M; ++e) qf2 += cabs2(qsrc[e]);
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    double lam[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      double r = 0.0;
      if (n < N) {
        const double *tr = basis + (((long long)b * N + n) * F + i) * K;
        const double *Vn = act + ((long long)b * N + n) * K * T;
        for (int k = 0; k < K; ++k) r = fma(tr[k], Vn[(long long)k * T + j], r);
      }
      lam[n] = r;
    }
    double rc[M];
    double rcmin = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      double r = 0.0;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) r = fma(lam[n], dd[n * M + m], r);
      rc[m] = r;
      rcmin = m == 0 ? r : (r < rcmin ? r : rcmin);
    }
    c128 x[M], z[M];
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] = X[(((long long)b * M + m) * F + i) * T + j];
    c128 sm[M];  // s_m = (Q~^H R^-1 x)_m, then scaled by q~[ref][m]
    const bool closed = EIG == 0 && floor_kind != SSSPY_FLOOR_ADD && rcmin > eps * qf2 * 1.0000001;
    if (!REPAIR && !closed) {
      redo[(long long)b * F + i] = 1;  // (every writer stores the same value)
      continue;
    }
    if (closed) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        c128 y = cmake(0.0, 0.0);
#pragma unroll
        for (int a = 0; a < M; ++a) cfma(y, qsrc[m * M + a], x[a]);
        const double g = 1.0 / rc[m];
        sm[m] = cmake(y.x * g, y.y * g);
      }
    } else if constexpr (REPAIR) {
      c128 A[M][M], P[M][M];
#pragma unroll
      for (int a = 0; a < M; ++a)
#pragma unroll
        for (int c2 = a; c2 < M; ++c2) {
          c128 s = cmake(0.0, 0.0);
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const c128 zz = cmulc(qt[a * M + m], qt[c2 * M + m]);
            s.x = fma(rc[m], zz.x, s.x);
            s.y = fma(rc[m], zz.y, s.y);
          }
          if (a == c2) s.y = 0.0;
          A[a][c2] = s;
          A[c2][a] = cconj(s);
        }
      double evs[M];
      const long long point = ((long long)b * F + i) * T + j;
      if constexpr (EIG == 2) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
          evs[k] = lam_io[point * M + k];
#pragma unroll
          for (int a = 0; a < M; ++a) P[a][k] = P_io[(point * M + a) * M + k];
        }
      } else {
        jacobi_eigh<M>(A, P);
#pragma unroll
        for (int k = 0; k < M; ++k) evs[k] = apply_floor(A[k][k].x, floor_kind, eps);
      }
      if constexpr (EIG == 1) {
        // ascending order without dynamic indexing (rank of every eigenvalue, ties by index)
#pragma unroll
        for (int k = 0; k < M; ++k) {
          int rank = 0;
#pragma unroll
          for (int l = 0; l < M; ++l)
            rank += (A[l][l].x < A[k][k].x || (A[l][l].x == A[k][k].x && l < k)) ? 1 : 0;
          lam_io[point * M + rank] = A[k][k].x;
#pragma unroll
          for (int a = 0; a < M; ++a) P_io[(point * M + a) * M + rank] = P[a][k];
        }
        continue;
      }
#pragma unroll
      for (int a = 0; a < M; ++a) z[a] = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) {
        c128 proj = cmake(0.0, 0.0);  // p_k^H x
#pragma unroll
        for (int a = 0; a < M; ++a) {
          const c128 pk = P[a][k];
          proj.x += pk.x * x[a].x + pk.y * x[a].y;
          proj.y += pk.x * x[a].y - pk.y * x[a].x;
        }
        const double ev = evs[k];
        proj = cmake(proj.x / ev, proj.y / ev);
#pragma unroll
        for (int a = 0; a < M; ++a) cfma(z[a], P[a][k], proj);
      }
#pragma unroll
      for (int m = 0; m < M; ++m) {
        c128 s = cmake(0.0, 0.0);
#pragma unroll
        for (int c2 = 0; c2 < M; ++c2) {
          const c128 qv = qt[c2 * M + m];
          s.x += qv.x * z[c2].x + qv.y * z[c2].y;
          s.y += qv.x * z[c2].y - qv.y * z[c2].x;
        }
        sm[m] = s;
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) sm[m] = cmul(qt[ref * M + m], sm[m]);
    for (int n = 0; n < N; ++n) {
      c128 y = cmake(0.0, 0.0);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const double wgt = lam[n] * dd[n * M + m];
        y.x = fma(wgt, sm[m].x, y.x);
        y.y = fma(wgt, sm[m].y, y.y);
      }
      Y[(((long long)b * N + n) * F + i) * T + j] = y;
    }
  }
}

#define FMG_DISPATCH_M(M_, CALL)                                                           \
  switch (M_) {                                                                            \
    case 2: { constexpr int MM = 2; CALL; } break;                                         \
    case 3: { constexpr int MM = 3; CALL; } break;                                         \
    case 4: { constexpr int MM = 4; CALL; } break;                                         \
    case 5: { constexpr int MM = 5; CALL; } break;                                         \
    case 6: { constexpr int MM = 6; CALL; } break;                                         \
    case 7: { constexpr int MM = 7; CALL; } break;                                         \
    case 8: { constexpr int MM = 8; CALL; } break;                                         \
    default: return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_channels must be in [2, 8]"); \
  }

template <int MODE>
static int launch_points(const void *X, const void *Q, const double *D, const double *basis,
                         const double *act, double *out0, double *out1, int B, int N, int M, int F,
                         int T, int K, hipStream_t st) {
  // point_terms keeps lam[NMAX]: more sources would silently drop out of R~
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  dim3 grid((T + 127) / 128, F, B), block(128);
  const size_t smem = bin_smem(N, M, K);
  FMG_DISPATCH_M(M, hipLaunchKernelGGL((k_points<MM, MODE>), grid, block, smem, st, (const c128 *)X,
                                       (const c128 *)Q, D, basis, act, out0, out1, N, F, T, K));
  return check_launch("fmnmf_generic points");
}

}  // namespace fmg

// ---- entry points used by mnmf_api.hip for shapes outside the MFMA-tile kernels
size_t fmnmf_generic_workspace_doubles(int B, int N, int M, int F, int T) {
  // A, Bt (B,N,F,T) each; the (B,M,F,T) weights reuse A's space when M <= N, else their own
  const size_t pts = (size_t)B * F * T;
  return pts * (2 * (size_t)N) + pts * (size_t)M;
}

int fmnmf_generic_update(const void *X, const void *C, void *Q, double *D, double *basis,
                         double *activation, int B, int N, int M, int F, int T, int K, int steps,
                         int floor_kind, double floor_eps, double *gws, void *U, double *qbuf,
                         int *info, hipStream_t st) {
  using namespace fmg;
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  if (M < 2 || M > 8) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_channels must be in [2, 8]");
  const size_t pts = (size_t)B * F * T;
  double *A = gws, *Bt = gws + pts * N, *Wt = gws + pts * 2 * N;
  int rc = SSSPY_OK;
  if (steps & SSSPY_MNMF_BASIS) {
    rc = launch_points<MODE_TRACES>(X, Q, D, basis, activation, A, Bt, B, N, M, F, T, K, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_basis, dim3(F, N, B), dim3(256), 0, st, basis, (const double *)activation,
                       (const double *)A, (const double *)Bt, N, F, T, K, floor_kind, floor_eps);
    rc = check_launch("fmnmf_generic basis");
    if (rc) return rc;
  }
  if (steps & SSSPY_MNMF_ACTIVATION) {
    rc = launch_points<MODE_TRACES>(X, Q, D, basis, activation, A, Bt, B, N, M, F, T, K, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_activation, dim3((T + 255) / 256, K, N * B), dim3(256), 0, st,
                       (const double *)basis, activation, (const double *)A, (const double *)Bt, N,
                       F, T, K, floor_kind, floor_eps);
    rc = check_launch("fmnmf_generic activation");
    if (rc) return rc;
  }
  bool have_q = false;
  if (steps & SSSPY_MNMF_DIAGONALIZER) {
    rc = launch_points<MODE_WEIGHTS>(X, Q, D, basis, activation, Wt, nullptr, B, N, M, F, T, K, st);
    if (rc) return rc;
    rc = ssspy_weighted_covariance(X, Wt, SSSPY_WEIGHT_BIN_FRAME, U, B, M, M, F, T, (void *)st);
    if (rc) return rc;
    rc = ip1_with_power(Q, U, C, C ? qbuf : nullptr, B, F, M, floor_kind, floor_eps, info, st);
    if (rc) return rc;
    have_q = C != nullptr;
  }
  if (steps & SSSPY_MNMF_SPATIAL) {
    const size_t smem = bin_smem(N, M, K) + (size_t)PB * (NMAX + 2 * M) * sizeof(double);
    FMG_DISPATCH_M(M, hipLaunchKernelGGL((k_spatial<MM>), dim3(F, B), dim3(PB), smem, st,
                                         (const c128 *)X, (const c128 *)Q, D,
                                         (const double *)basis, (const double *)activation, N, F, T,
                                         K));
    rc = check_launch("fmnmf_generic spatial");
    if (rc) return rc;
  }
  if (steps & SSSPY_MNMF_NORMALIZE) {
    if (!have_q) {
      rc = row_power(Q, C, qbuf, B, F, M, st);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k_norm_scale, dim3((F + 63) / 64, B), dim3(256), 0, st, (c128 *)Q, D,
                       (const double *)qbuf, N, M, F, floor_kind, floor_eps);
    rc = check_launch("fmnmf_generic norm_scale");
  }
  return rc;
}

int fmnmf_generic_weights(const void *X, const void *Q, const double *D, const double *basis,
                          const double *act, double *Wt, int B, int N, int M, int F, int T, int K,
                          hipStream_t st) {
  return fmg::launch_points<fmg::MODE_WEIGHTS>(X, Q, D, basis, act, Wt, nullptr, B, N, M, F, T, K, st);
}

size_t fmnmf_generic_loss_ws_bytes(int B, int F, int T) {
  return scalar_slots_bytes(B, ((T + 127) / 128) * F);
}
// out[b] = the data term; loss_ws: fmnmf_generic_loss_ws_bytes() (every block writes its slot)
int fmnmf_generic_loss(const void *X, const void *Q, const double *D, const double *basis,
                       const double *act, double *out, void *loss_ws, int B, int N, int M, int F,
                       int T, int K, hipStream_t st) {
  const int rc = fmg::launch_points<fmg::MODE_LOSS>(X, Q, D, basis, act, (double *)loss_ws, nullptr,
                                                    B, N, M, F, T, K, st);
  return rc ? rc : scalar_slots_fold(loss_ws, B, ((T + 127) / 128) * F, out, 0, st);
}

int fmnmf_generic_separate(const void *X, const void *Q, void *Qinv, const double *D,
                           const double *basis, const double *act, void *Y, int B, int N, int M,
                           int F, int T, int K, int ref, int floor_kind, double eps, int *info,
                           int *redo, hipStream_t st) {
  // redo: B F ints of scratch (bins the closed-form launch hands to the general one)
  using namespace fmg;
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  const long long nbins = (long long)B * F;
  hipError_t e = hipMemsetAsync(redo, 0, (size_t)nbins * sizeof(int), st);
  if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  FMG_DISPATCH_M(M, {
    hipLaunchKernelGGL((k_qinv<MM>), dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                       (const c128 *)Q, (c128 *)Qinv, nbins, info);
    hipLaunchKernelGGL((k_separate<MM, false>), dim3(F, B), dim3(128), 0, st, (const c128 *)X,
                       (const c128 *)Q, (const c128 *)Qinv, D, basis, act, (c128 *)Y, N, F, T, K,
                       ref, floor_kind, eps, redo);
    hipLaunchKernelGGL((k_separate<MM, true>), dim3(F, B), dim3(128), 0, st, (const c128 *)X,
                       (const c128 *)Q, (const c128 *)Qinv, D, basis, act, (c128 *)Y, N, F, T, K,
                       ref, floor_kind, eps, redo);
  });
  return check_launch("fmnmf_generic separate");
}

// The Wiener filter split at the eigenvalue floor of to_psd (an arbitrary flooring callable):
// stage 1 leaves ascending eigenvalues lam (B,F,T,M) and eigenvectors P (B,F,T,M,M); the host floors
// lam; stage 2 finishes.  Any N, M <= 8.
int fmnmf_generic_separate_eig(const void *X, const void *Q, void *Qinv, const double *D,
                               const double *basis, const double *act, void *Y, int B, int N,
                               int M, int F, int T, int K, int ref, int stage, double *lam,
                               void *P, int *info, hipStream_t st) {
  using namespace fmg;
  if (N < 1 || N > NMAX) return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [1, 8]");
  const long long nbins = (long long)B * F;
  FMG_DISPATCH_M(M, {
    if (stage == 1) {
      hipLaunchKernelGGL((k_qinv<MM>), dim3((unsigned)((nbins + 63) / 64)), dim3(64), 0, st,
                         (const c128 *)Q, (c128 *)Qinv, nbins, info);
      hipLaunchKernelGGL((k_separate<MM, true, 1>), dim3(F, B), dim3(128), 0, st, (const c128 *)X,
                         (const c128 *)Q, (const c128 *)Qinv, D, basis, act, (c128 *)Y, N, F, T, K,
                         ref, SSSPY_FLOOR_NONE, 0.0, (int *)nullptr, lam, (c128 *)P);
    } else {
      hipLaunchKernelGGL((k_separate<MM, true, 2>), dim3(F, B), dim3(128), 0, st, (const c128 *)X,
                         (const c128 *)Q, (const c128 *)Qinv, D, basis, act, (c128 *)Y, N, F, T, K,
                         ref, SSSPY_FLOOR_NONE, 0.0, (int *)nullptr, lam, (c128 *)P);
    }
  });
  return check_launch("fmnmf_generic separate (eigen stages)");
}

}  // namespace ssspy
