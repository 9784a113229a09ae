// ILRMA source-model updates for shapes the register-tiled passes do not hold: n_basis above 64 (their
// activation staging and accumulators are sized for <= 64 bases) and more than 8 sources off the
// grouped path.  Round 3 left these to the first-generation kernels (n_basis 80: 25 ms per iteration
// at 32 mixtures of the configs[1] shape against 4.7 ms at 64).  Here the update is what it is on
// paper -- three dense products per source and a few element-wise maps:
//   R  = T V                      (F x T)    a batched GEMM whose epilogue maps R (and |y|^2) to
//   a  = mm numerator factor, b = 1 / R      a and b (or to the covariance pass's weight 1 / R~)
//   basis:      num = a V^T,  den = b V^T    (F x K)   one batched GEMM, both right-hand sides
//   activation: num = T^T a,  den = T^T b    (K x T)   one batched GEMM
//   state <- floor(state * (num / den)^e)              k_mu_update
// with the source count, n_basis and the source model at run time (mm_weights / mm_ratio_pow of
// ilrma_params.hpp cover Gauss, t, GGD, any domain, MM and ME).
// ref: ssspy/bss/ilrma.py:1051-1204 (Gauss), :2470-2607 (t), :3790-3985 (GGD).
#include "common.hpp"
#include "ilrma_params.hpp"
#include "ssspy_amd.h"

namespace ssspy {

// ---------------------------------------------------------------------------- batched fp64 GEMM
// C[g] (M x N) = A[g] (M x Kd) B[g] (Kd x N) on v_mfma_f64_16x16x4.  Operands by strides (element
// (r, c) at base + r * rs + c * cs, one of the two strides is 1); `dual`: batch g uses operand set
// g & 1 (A0 / A1, C0 / C1) and batch index g >> 1 -- the numerator and denominator products share
// the other operand.  Workgroup = 64 x 64 tile of C, 4 waves of 32 x 32 (2 x 2 MFMA tiles),
// Kd walked 16 at a time through LDS ([kd][row] with the row index contiguous, so the operand reads
// of a k-step are conflict-free), double buffered.
struct GemmSide {
  const double *p0, *p1;  // the two operand sets (p1 == p0 unless dual)
  long long batch;        // stride between batches
  long long rs, cs;       // row / column strides
};
constexpr int GT = 64, GK = 16, GLD = GT + 2;

__device__ __forceinline__ void gemm_stage_load(double (&reg)[4], const double *base, long long rs,
                                                long long cs, int row0, int k0, int nrows, int nk,
                                                bool a_side) {
  // a_side: tile is (64 rows x 16 kd), element (row, kd) at base + row * rs + kd * cs
  // b side: tile is (16 kd x 64 cols), element (kd, col) at base + kd * rs + col * cs
  // -> both as "outer (64) x kd (16)" with strides (so, sk)
  const long long so = a_side ? rs : cs, sk = a_side ? cs : rs;
  const int t = threadIdx.x;
  if (sk == 1) {  // kd contiguous: thread = (outer t / 4, kd chunk (t % 4) * 4 .. + 4)
    const int o = t >> 2, kc = (t & 3) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = row0 + o < nrows && k0 + kc + u < nk;
      reg[u] = ok ? base[(long long)(row0 + o) * so + (k0 + kc + u)] : 0.0;
    }
  } else {  // outer contiguous: thread = (kd t / 16, outer chunk (t % 16) * 4 .. + 4)
    const int kk = t >> 4, oc = (t & 15) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = row0 + oc + u < nrows && k0 + kk < nk;
      reg[u] = ok ? base[(long long)(row0 + oc + u) * so + (long long)(k0 + kk) * sk] : 0.0;
    }
  }
}

__device__ __forceinline__ void gemm_stage_store(const double (&reg)[4], double *tile, long long rs,
                                                 long long cs, bool a_side) {
  const long long sk = a_side ? cs : rs;
  const int t = threadIdx.x;
  if (sk == 1) {
    const int o = t >> 2, kc = (t & 3) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[(kc + u) * GLD + o] = reg[u];
  } else {
    const int kk = t >> 4, oc = (t & 15) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[kk * GLD + oc + u] = reg[u];
  }
}

// Epilogues: EPI_STORE C = the product; EPI_AB the product is R = (T V) of one source: with P = |y|^2
// (ypow, or |y|^2 of y) C0 <- the numerator factor a, C1 <- b = 1 / R of the MM update (mm_weights);
// EPI_PHI C0 <- the spatial weight 1 / R~ of the covariance pass (spatial_weight).
// EPI_LOSS: the workgroup's sum of loss_term(P, R) over its tile goes to C0[batch][tile] (one slot per
// workgroup, no atomics; wb_loss_data adds a mixture's slots in a fixed order).
enum { EPI_STORE = 0, EPI_AB = 1, EPI_PHI = 2, EPI_LOSS = 3 };
struct GemmEpi {
  const double *ypow;  // (batches, M, N) f64, or NULL
  const c128 *y;       // (batches, M, N) complex when ypow is NULL (may be NULL for EPI_PHI / Gauss)
  IlrmaDims d;
};

// grid: (ceil(M / 64), ceil(N / 64), batches)
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f64(GemmSide A, GemmSide Bm, double *C0, double *C1,
                                                  long long c_batch, int M, int N, int Kd,
                                                  int dual, GemmEpi epi) {
  __shared__ double As[2][GK * GLD], Bs[2][GK * GLD];
  const int g = blockIdx.z;
  const int set = dual ? (g & 1) : 0;
  const long long bi = dual ? (g >> 1) : g;
  const double *Ab = (set ? A.p1 : A.p0) + bi * A.batch;
  const double *Bb = (set ? Bm.p1 : Bm.p0) + bi * Bm.batch;
  double *Cb = (set ? C1 : C0) + bi * c_batch;
  const int m0 = blockIdx.x * GT, n0 = blockIdx.y * GT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  double ra[4], rb[4];
  const int nst = (Kd + GK - 1) / GK;
  gemm_stage_load(ra, Ab, A.rs, A.cs, m0, 0, M, Kd, true);
  gemm_stage_load(rb, Bb, Bm.rs, Bm.cs, n0, 0, N, Kd, false);
  gemm_stage_store(ra, As[0], A.rs, A.cs, true);
  gemm_stage_store(rb, Bs[0], Bm.rs, Bm.cs, false);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const int cur = s & 1;
    if (s + 1 < nst) {
      gemm_stage_load(ra, Ab, A.rs, A.cs, m0, (s + 1) * GK, M, Kd, true);
      gemm_stage_load(rb, Bb, Bm.rs, Bm.cs, n0, (s + 1) * GK, N, Kd, false);
    }
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      const double *ar = As[cur] + (4 * ks + q) * GLD, *br = Bs[cur] + (4 * ks + q) * GLD;
      const double a0 = ar[wm + c], a1 = ar[wm + 16 + c];
      const double b0 = br[wn + c], b1 = br[wn + 16 + c];
      acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    }
    if (s + 1 < nst) {
      gemm_stage_store(ra, As[cur ^ 1], A.rs, A.cs, true);
      gemm_stage_store(rb, Bs[cur ^ 1], Bm.rs, Bm.cs, false);
    }
    __syncthreads();
  }
  if (EPI == EPI_LOSS) {
    __shared__ double red[4];
    double local = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm + 16 * i + q + 4 * r, col = n0 + wn + 16 * j + c;
          if (row < M && col < N) {
            const long long ge = bi * c_batch + (long long)row * N + col;
            const double P = epi.ypow ? epi.ypow[ge] : cabs2(epi.y[ge]);
            local += loss_term(P, acc[i][j][r], epi.d);
          }
        }
    const double total = block_sum(local, red);
    if (threadIdx.x == 0)
      C0[((long long)bi * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y] = total;
    return;
  }
  // D: row = q + 4 r, col = c of each 16 x 16 tile; C is row-major (N contiguous)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + 16 * i + q + 4 * r, col = n0 + wn + 16 * j + c;
        if (row < M && col < N) {
          const long long e = (long long)row * N + col;
          if (EPI == EPI_STORE) {
            Cb[e] = acc[i][j][r];
          } else {
            const long long ge = bi * c_batch + e;
            const bool need = EPI == EPI_AB || epi.d.model != SSSPY_SOURCE_GAUSS;
            const double P = need ? (epi.ypow ? epi.ypow[ge] : cabs2(epi.y[ge])) : 0.0;
            if (EPI == EPI_AB) {
              double wa, wb;
              mm_weights(P, acc[i][j][r], epi.d, true, wa, wb);
              C0[ge] = wa;
              C1[ge] = wb;
            } else {
              C0[ge] = spatial_weight(P, acc[i][j][r], epi.d);
            }
          }
        }
      }
}

static int launch_gemm(const GemmSide &A, const GemmSide &Bm, double *C0, double *C1,
                       long long c_batch, int M, int N, int Kd, int batches, int dual,
                       hipStream_t st) {
  dim3 grid((M + GT - 1) / GT, (N + GT - 1) / GT, dual ? 2 * batches : batches);
  hipLaunchKernelGGL(k_gemm_f64<EPI_STORE>, grid, dim3(256), 0, st, A, Bm, C0, C1, c_batch, M, N,
                     Kd, dual, GemmEpi{});
  return check_launch("k_gemm_f64");
}

// out[b] = scale * (sum of the mixture's `count` slots), one block per mixture, fixed order
__global__ __launch_bounds__(256) void k_wb_sum_slots(const double *__restrict__ slots, double *out,
                                                      long long count, double scale) {
  __shared__ double scratch[4];
  const int b = blockIdx.x;
  double s = 0.0;
  for (long long e = threadIdx.x; e < count; e += blockDim.x) s += slots[b * count + e];
  const double total = block_sum(s, scratch);
  if (threadIdx.x == 0) out[b] = total * scale;
}

size_t wb_loss_ws_bytes(int B, int N, int F, int T) {
  return (size_t)B * N * ((F + GT - 1) / GT) * ((T + GT - 1) / GT) * sizeof(double);
}

// Data term of the negative log-likelihood, out[b] = sum_{n,i} mean_j loss_term(|y|^2, (T V)_nij), any
// n_basis / source count / model: T V tile by tile on the matrix cores, the terms summed per
// workgroup in the epilogue.  ypow / y: |y|^2 or y (B N, F, T); ws: wb_loss_ws_bytes().
// ref: ssspy/bss/ilrma.py:1946-1965 (Gauss), :3301-3305 (t), :4377-4381 (GGD).
int wb_loss_data(const double *basis, const double *activation, const double *ypow, const void *y,
                 double *out, void *ws, int B, int N, int F, int T, int K, const IlrmaDims &d,
                 hipStream_t st) {
  const GemmSide A{basis, basis, (long long)F * K, (long long)K, 1};
  const GemmSide Bm{activation, activation, (long long)K * T, (long long)T, 1};
  dim3 grid((F + GT - 1) / GT, (T + GT - 1) / GT, B * N);
  const GemmEpi epi{ypow, (const c128 *)y, d};
  hipLaunchKernelGGL(k_gemm_f64<EPI_LOSS>, grid, dim3(256), 0, st, A, Bm, (double *)ws,
                     (double *)nullptr, (long long)F * T, F, T, K, 0, epi);
  int rc = check_launch("k_gemm_f64 (loss)");
  if (rc) return rc;
  hipLaunchKernelGGL(k_wb_sum_slots, dim3(B), dim3(256), 0, st, (const double *)ws, out,
                     (long long)N * grid.x * grid.y, 1.0 / (double)T);
  return check_launch("k_wb_sum_slots");
}

// R = T V per source (F x T, n_basis deep) with the element-wise map in the epilogue:
// mode EPI_AB -> out0 = a, out1 = b; EPI_PHI -> out0 = varphi.  ypow / y: |y|^2 or y, (B N, F, T).
int wb_tv_weights(int mode, const double *basis, const double *activation, const double *ypow,
                  const void *y, double *out0, double *out1, int BN, int F, int T, int K,
                  const IlrmaDims &d, hipStream_t st) {
  const GemmSide A{basis, basis, (long long)F * K, (long long)K, 1};        // T[i][k]
  const GemmSide Bm{activation, activation, (long long)K * T, (long long)T, 1};  // V[k][j]
  dim3 grid((F + GT - 1) / GT, (T + GT - 1) / GT, BN);
  const GemmEpi epi{ypow, (const c128 *)y, d};
  if (mode == EPI_AB)
    hipLaunchKernelGGL(k_gemm_f64<EPI_AB>, grid, dim3(256), 0, st, A, Bm, out0, out1,
                       (long long)F * T, F, T, K, 0, epi);
  else
    hipLaunchKernelGGL(k_gemm_f64<EPI_PHI>, grid, dim3(256), 0, st, A, Bm, out0, out1,
                       (long long)F * T, F, T, K, 0, epi);
  return check_launch("k_gemm_f64 (T V epilogue)");
}

// state <- floor(state * (num / den)^e), element-wise.  ref: ssspy/bss/ilrma.py:1126-1128, :1202-1204
__global__ __launch_bounds__(256) void k_mu_update(double *state, const double *__restrict__ num,
                                                   const double *__restrict__ den, long long count,
                                                   IlrmaDims d) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  state[e] = apply_floor(state[e] * mm_ratio_pow(num[e], den[e], d), d.floor_kind, d.floor_eps);
}

// ---- the two updates.  a, b: (B N F T) each, written by k_ilrma_iss_weight's (a, b) mode;
// nd: 2 x B N max(F, T) K doubles of scratch (num, den)
int wb_update_basis(const double *a, const double *b, double *basis, const double *activation,
                    double *nd, int BN, int F, int T, int K, const IlrmaDims &d, hipStream_t st) {
  double *num = nd, *den = nd + (long long)BN * F * K;
  // num[g] (F x K) = a[g] (F x T) V[g]^T: B operand element (kd = j, col = k) = V[k][j]
  const GemmSide A{a, b, (long long)F * T, (long long)T, 1};
  const GemmSide Bm{activation, activation, (long long)K * T, 1, (long long)T};
  int rc = launch_gemm(A, Bm, num, den, (long long)F * K, F, K, T, BN, 1, st);
  if (rc) return rc;
  const long long count = (long long)BN * F * K;
  hipLaunchKernelGGL(k_mu_update, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, basis,
                     (const double *)num, (const double *)den, count, d);
  return check_launch("k_mu_update (basis)");
}

int wb_update_activation(const double *a, const double *b, const double *basis, double *activation,
                         double *nd, int BN, int F, int T, int K, const IlrmaDims &d,
                         hipStream_t st) {
  double *num = nd, *den = nd + (long long)BN * K * T;
  // num[g] (K x T) = T[g]^T (K x F) a[g] (F x T): A operand element (row = k, kd = i) = T[i][k]
  const GemmSide A{basis, basis, (long long)F * K, 1, (long long)K};
  const GemmSide Bm{a, b, (long long)F * T, (long long)T, 1};
  // (dual picks the operand set by g & 1 on BOTH sides; A's two sets are the same basis)
  int rc = launch_gemm(A, Bm, num, den, (long long)K * T, K, T, F, BN, 1, st);
  if (rc) return rc;
  const long long count = (long long)BN * K * T;
  hipLaunchKernelGGL(k_mu_update, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                     activation, (const double *)num, (const double *)den, count, d);
  return check_launch("k_mu_update (activation)");
}

}  // namespace ssspy
