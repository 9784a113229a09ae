// C-ABI entry points of the FastGaussMNMF path (kernels: mnmf_kernels.hip, one unit per N).
#include "common.hpp"

namespace ssspy {

#define DECL_N(n)                                                                                \
  int mnmf_handover_ok_n##n(int, int, int, int);                                                \
  int mnmf_loss_handover_n##n(const double *, const double *, const double *, const double *,   \
                              const double *, double *, void *, int, int, int, int, int,        \
                              hipStream_t);                                                     \
  int mnmf_loss_handover_slots_n##n(int, int, int);                                             \
  int mnmf_loss_handover_raw_n##n(const double *, const double *, const double *, const double *, \
                                  const double *, double *, long long, int, int, int, int, int, \
                                  hipStream_t);                                                 \
  size_t mnmf_loss_ws_bytes_n##n(int, int);                                                     \
  int mnmf_qx2_n##n(const void *, const void *, double *, double *, int, int, int, int,         \
                    hipStream_t);                                                               \
  int mnmf_basis_n##n(const void *, const void *, const double *, const double *, double *,     \
                      const double *, int, int, int, int, int, int, double, double *,           \
                      const double *, const double *, hipStream_t);                             \
  int mnmf_activation_n##n(const void *, const void *, const double *, const double *,          \
                           const double *, double *, int, int, int, int, int, int,              \
                           const double *, const double *, hipStream_t);                        \
  int mnmf_wcov_n##n(const void *, const double *, const double *, const double *, void *, int, \
                     int, int, int, int, double *, int *, long long *, hipStream_t);            \
  int mnmf_spatial_n##n(const void *, const void *, double *, const double *, const double *,   \
                        int, int, int, int, int, double *, double *, double *, int, int *,      \
                        hipStream_t);                                                           \
  int mnmf_loss_n##n(const void *, const void *, const double *, const double *, const double *, \
                     double *, void *, int, int, int, int, int, hipStream_t);                   \
  int mnmf_norm_scale_n##n(void *, double *, const double *, int, int, int, int, double,        \
                           double *, int, const double *, int, hipStream_t);                    \
  int mnmf_separate_n##n(const void *, const void *, void *, const double *, const double *,    \
                         const double *, void *, int, int, int, int, int, int, int, double,     \
                         int *, int *, hipStream_t);
DECL_N(2) DECL_N(3) DECL_N(4)
#undef DECL_N

#define MNMF_DISPATCH(N_, fn, ...)                                                       \
  switch (N_) {                                                                          \
    case 2: return fn##_n2(__VA_ARGS__);                                                 \
    case 3: return fn##_n3(__VA_ARGS__);                                                 \
    case 4: return fn##_n4(__VA_ARGS__);                                                 \
    default: return fail(SSSPY_ERR_UNSUPPORTED, "FastMNMF: n_sources must be in [2, 4]"); \
  }

int ip1_with_power(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N,
                   int floor_kind, double floor_eps, int *info, hipStream_t st);
int row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st);
bool ip1_small_shape(int B, int F, int N);
int ip1_from_records(void *W, const void *records, int nchunks, int rbins, long long rec_stride,
                     const void *C, double *qbuf, int B, int F, int N, int floor_kind,
                     double floor_eps, int *info, hipStream_t st, double *logdet,
                     long long logdet_stride);

// general shapes (n_sources or n_channels above 4): the point-wise path of fmnmf_generic.hip
size_t fmnmf_generic_workspace_doubles(int B, int N, int M, int F, int T);
int fmnmf_generic_update(const void *X, const void *C, void *Q, double *D, double *basis,
                         double *activation, int B, int N, int M, int F, int T, int K, int steps,
                         int floor_kind, double floor_eps, double *gws, void *U, double *qbuf,
                         int *info, hipStream_t st);
int fmnmf_generic_weights(const void *X, const void *Q, const double *D, const double *basis,
                          const double *act, double *Wt, int B, int N, int M, int F, int T, int K,
                          hipStream_t st);
size_t fmnmf_generic_loss_ws_bytes(int B, int F, int T);
int fmnmf_generic_loss(const void *X, const void *Q, const double *D, const double *basis,
                       const double *act, double *out, void *loss_ws, int B, int N, int M, int F,
                       int T, int K, hipStream_t st);
int fmnmf_generic_separate(const void *X, const void *Q, void *Qinv, const double *D,
                           const double *basis, const double *act, void *Y, int B, int N, int M,
                           int F, int T, int K, int ref, int floor_kind, double eps, int *info,
                           int *redo, hipStream_t st);
int fmnmf_generic_separate_eig(const void *X, const void *Q, void *Qinv, const double *D,
                               const double *basis, const double *act, void *Y, int B, int N,
                               int M, int F, int T, int K, int ref, int stage, double *lam,
                               void *P, int *info, hipStream_t st);

// the MFMA-tile kernels (mnmf_kernels.hip) are compiled for 2..4 sources and channels
static inline bool mnmf_tiled(int N, int M) { return N >= 2 && N <= 4 && M >= 2 && M <= 4; }

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static inline int mnmf_chunks(int B, int F, int T, int K) {
  const long long blocks0 = (long long)B * ((T + 63) / 64) * ((K + 15) / 16);
  const int ntiles = (F + 15) / 16;
  long long want = (512 + blocks0 - 1) / blocks0;
  if (want < 1) want = 1;
  if (want > 16) want = 16;
  if (want > ntiles) want = ntiles;
  return (int)want;
}

struct MnmfWs {
  size_t part, btmp, U, qbuf, qinv, tail, generic, total;
};

// partial sums of the split work items of the bin-major kernels' last scheduling round
// (tail_plan.hpp; at most 512 (item, chunk) slots; mnmf_tail_doubles() in mnmf_kernels.hip)
static inline size_t mnmf_tail_bytes(int N, int M) {
  size_t a = (size_t)N * 64 * 16 * 2, b = (size_t)64 * M * M * M * 2, c = (size_t)64 * N * M * 2;
  size_t m = a > b ? (a > c ? a : c) : (b > c ? b : c);
  return 512 * m * sizeof(double);
}

static inline MnmfWs mnmf_ws(int B, int N, int M, int F, int T, int K) {
  MnmfWs w;
  size_t off = 0;
  w.part = off;
  off += al((size_t)B * mnmf_chunks(B, F, T, K) * N * 2 * K * T * sizeof(double));
  w.btmp = off;
  off += K > 16 ? al((size_t)B * N * F * K * sizeof(double)) : 0;
  w.U = off;
  off += al((size_t)B * F * M * M * M * 2 * sizeof(double));
  w.qbuf = off;
  off += al((size_t)B * F * M * sizeof(double));
  w.qinv = off;
  off += al((size_t)B * F * M * M * 2 * sizeof(double));
  w.tail = off;
  off += al(mnmf_tail_bytes(N, M));
  w.generic = off;
  if (!mnmf_tiled(N, M)) off += al(fmnmf_generic_workspace_doubles(B, N, M, F, T) * sizeof(double));
  w.total = off;
  return w;
}

// V <- floor(V * sqrt(sum_chunks num / sum_chunks den))
__global__ __launch_bounds__(256) void k_mnmf_activation_finalize(double *act,
                                                                  const double *__restrict__ part,
                                                                  int N, int K, int T, int nchunks,
                                                                  int floor_kind, double eps) {
  const int b = blockIdx.z, n = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)K * T) return;
  // chunks in order, eight loads per round trip (ordered_sum, common.hpp)
  double *dst = act + ((long long)b * N + n) * K * T + e;
  const double vold = *dst;
  const double *src = part + ((((long long)b * nchunks) * N + n) * 2) * K * T + e;
  const long long stride = (long long)N * 2 * K * T;
  const double sn = ordered_sum(src, stride, nchunks);
  const double sd = ordered_sum(src + (long long)K * T, stride, nchunks);
  *dst = apply_floor(vold * sqrt(sn / sd), floor_kind, eps);
}

// P / pscale: the |Q x|^2 hand-over (nullptr: the pass reads x and Q)
static int step_basis(const void *X, const void *Q, const double *D, double *basis,
                      const double *act, int B, int N, int M, int F, int T, int K, int fk,
                      double eps, char *ws, const MnmfWs &w, const double *P,
                      const double *pscale, hipStream_t st) {
  double *out = K > 16 ? (double *)(ws + w.btmp) : basis;
  auto run = [&]() -> int {
    MNMF_DISPATCH(N, mnmf_basis, X, Q, D, basis, out, act, B, M, F, T, K, fk, eps,
                  (double *)(ws + w.tail), P, pscale, st);
  };
  int rc = run();
  if (rc) return rc;
  if (out != basis) {
    hipError_t e = hipMemcpyAsync(basis, out, (size_t)B * N * F * K * sizeof(double),
                                  hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  }
  return SSSPY_OK;
}

static int step_activation(const void *X, const void *Q, const double *D, const double *basis,
                           double *act, int B, int N, int M, int F, int T, int K, int fk,
                           double eps, char *ws, const MnmfWs &w, const double *P,
                           const double *pscale, hipStream_t st) {
  const int chunks = mnmf_chunks(B, F, T, K);
  double *part = (double *)(ws + w.part);
  auto run = [&]() -> int {
    MNMF_DISPATCH(N, mnmf_activation, X, Q, D, basis, act, part, chunks, B, M, F, T, K, P, pscale,
                  st);
  };
  int rc = run();
  if (rc) return rc;
  dim3 g2((unsigned)(((long long)K * T + 255) / 256), N, B);
  hipLaunchKernelGGL(k_mnmf_activation_finalize, g2, dim3(256), 0, st, act, part, N, K, T, chunks,
                     fk, eps);
  return check_launch("k_mnmf_activation_finalize");
}

static int handover_ok(int B, int N, int M, int F, int T, int K) {
  if (!mnmf_tiled(N, M)) return 0;
  MNMF_DISPATCH(N, mnmf_handover_ok, B, F, T, K);
}
static int handover_fill(const void *X, const void *Q, double *P, double *pscale, int B, int N,
                         int M, int F, int T, hipStream_t st) {
  MNMF_DISPATCH(N, mnmf_qx2, X, Q, P, pscale, B, M, F, T, st);
}

// One iteration (or a subset of its steps).  `handover` (optional): |Q x|^2 (B, M, F, T) followed
// by its per-(mixture, channel) scale (B, M); *valid says whether it matches Q and x on entry and,
// on return, on exit.
static int fastmnmf_update_impl(const void *X, const void *C, void *Q, double *D, double *basis,
                                double *activation, int B, int N, int M, int F, int T, int K,
                                int steps, int floor_kind, double floor_eps, void *workspace,
                                size_t workspace_bytes, int *info, double *handover, int *valid,
                                hipStream_t st, double *logdet = nullptr,
                                long long logdet_stride = 0) {
  // logdet (optional; with SSSPY_MNMF_DIAGONALIZER among the steps): sum_i log|det Q_i| of the
  // diagonalisers AS THEY COME IN, as ssspy_fastmnmf_deferred_logdet_slots() shares per mixture
  // (logdet[s * logdet_stride + b]; see ssspy_fastmnmf_update_handover_logdet)
  SSSPY_REQUIRE(X && Q && D && basis && activation && B > 0 && F > 0 && T > 0,
                "fastmnmf_update: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "fastmnmf_update: n_basis must be in [1, 65536]");
  const MnmfWs w = mnmf_ws(B, N, M, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "fastmnmf_update: workspace too small");
  SSSPY_REQUIRE(!(steps & SSSPY_MNMF_NORMALIZE) || C, "fastmnmf_update: normalisation needs C");
  char *ws = (char *)workspace;
  int rc = SSSPY_OK;
  if (logdet) {
    SSSPY_REQUIRE(steps & SSSPY_MNMF_DIAGONALIZER, "fastmnmf_update: logdet without the IP1 step");
  }
  if (!mnmf_tiled(N, M)) {
    SSSPY_REQUIRE(!handover, "fastmnmf_update: no hand-over for this shape");
    if (logdet) {
      rc = ssspy_sum_logdet(Q, logdet, B, F, M, (void *)st);
      if (rc) return rc;
    }
    return fmnmf_generic_update(X, C, Q, D, basis, activation, B, N, M, F, T, K, steps, floor_kind,
                                floor_eps, (double *)(ws + w.generic), ws + w.U,
                                (double *)(ws + w.qbuf), info, st);
  }
  double *P = handover, *pscale = handover ? handover + (size_t)B * M * F * T : nullptr;
  if (handover) {
    SSSPY_REQUIRE(valid, "fastmnmf_update: hand-over without its validity flag");
    SSSPY_REQUIRE(handover_ok(B, N, M, F, T, K) == 1, "fastmnmf_update: no hand-over for this shape");
  }
  bool have_p = handover && *valid;
  if (handover && !have_p && (steps & (SSSPY_MNMF_BASIS | SSSPY_MNMF_ACTIVATION))) {
    rc = handover_fill(X, Q, P, pscale, B, N, M, F, T, st);
    if (rc) return rc;
    have_p = true;
  }
  if (valid) *valid = 0;  // until the call is through
  if (steps & SSSPY_MNMF_BASIS) {
    rc = step_basis(X, Q, D, basis, activation, B, N, M, F, T, K, floor_kind, floor_eps, ws, w,
                    have_p ? P : nullptr, pscale, st);
    if (rc) return rc;
  }
  if (steps & SSSPY_MNMF_ACTIVATION) {
    rc = step_activation(X, Q, D, basis, activation, B, N, M, F, T, K, floor_kind, floor_eps, ws, w,
                         have_p ? P : nullptr, pscale, st);
    if (rc) return rc;
  }
  double *qbuf = (double *)(ws + w.qbuf);
  bool have_q = false;
  if (steps & SSSPY_MNMF_DIAGONALIZER) {
    void *U = ws + w.U;
    // a handful of mixtures: every item of the pass is split, and IP1's latency form folds the
    // partial records itself (no fold kernel, U is not materialised)
    int split = 0;
    long long rec = 0;
    const bool small = ip1_small_shape(B, F, M);
    auto run = [&]() -> int {
      MNMF_DISPATCH(N, mnmf_wcov, X, D, basis, activation, U, B, M, F, T, K,
                    (double *)(ws + w.tail), small ? &split : nullptr, &rec, st);
    };
    rc = run();
    if (rc) return rc;
    // IP1 on the M x M diagonaliser with M weighted covariances per bin
    if (split) {
      rc = ip1_from_records(Q, ws + w.tail, split, 64, rec, C, C ? qbuf : nullptr, B, F, M,
                            floor_kind, floor_eps, info, st, logdet, logdet_stride);
    } else {
      if (logdet) {  // (no by-product on this route: the finished sums go to share 0)
        rc = ssspy_sum_logdet(Q, logdet, B, F, M, (void *)st);
        if (rc) return rc;
      }
      rc = ip1_with_power(Q, U, C, C ? qbuf : nullptr, B, F, M, floor_kind, floor_eps, info, st);
    }
    if (rc) return rc;
    have_q = C != nullptr;
    have_p = false;  // Q moved
  }
  bool fresh_scale = false;  // P written below and its scale left to the normalisation
  int spatial_split = 0;     // > 0: the fold of the spatial pass is left to the normalisation too
  if (steps & SSSPY_MNMF_SPATIAL) {
    const bool norm_next = (steps & SSSPY_MNMF_NORMALIZE) != 0;
    fresh_scale = handover != nullptr && norm_next;
    auto run = [&]() -> int {
      MNMF_DISPATCH(N, mnmf_spatial, X, Q, D, basis, activation, B, M, F, T, K,
                    (double *)(ws + w.tail), P, pscale, fresh_scale ? 1 : 0,
                    norm_next ? &spatial_split : (int *)nullptr, st);
    };
    rc = run();
    if (rc) return rc;
    have_p = handover != nullptr;  // written with the Q of this pass, scale 1
  }
  if (steps & SSSPY_MNMF_NORMALIZE) {
    if (!have_q) {
      rc = row_power(Q, C, qbuf, B, F, M, st);
      if (rc) return rc;
    }
    auto run = [&]() -> int {
      MNMF_DISPATCH(N, mnmf_norm_scale, Q, D, qbuf, B, M, F, floor_kind, floor_eps,
                    have_p ? pscale : nullptr, fresh_scale ? 1 : 0,
                    (const double *)(ws + w.tail), spatial_split, st);
    };
    rc = run();
    if (rc) return rc;
  }
  if (valid) *valid = have_p ? 1 : 0;
  return rc;
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

size_t ssspy_fastmnmf_workspace_bytes(int B, int N, int M, int F, int T, int K) {
  if (B <= 0 || N <= 0 || M <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  return mnmf_ws(B, N, M, F, T, K).total;
}

int ssspy_fastmnmf_update(const void *X, const void *C, void *Q, double *D, double *basis,
                          double *activation, int B, int N, int M, int F, int T, int K, int steps,
                          int floor_kind, double floor_eps, void *workspace, size_t workspace_bytes,
                          int *info, void *stream) {
  return fastmnmf_update_impl(X, C, Q, D, basis, activation, B, N, M, F, T, K, steps, floor_kind,
                              floor_eps, workspace, workspace_bytes, info, nullptr, nullptr,
                              as_stream(stream));
}

size_t ssspy_fastmnmf_handover_doubles(int B, int N, int M, int F, int T, int K) {
  if (B <= 0 || N <= 0 || M <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  if (handover_ok(B, N, M, F, T, K) != 1) return 0;
  return (size_t)B * M * F * T + (size_t)B * M;
}

int ssspy_fastmnmf_update_handover(const void *X, const void *C, void *Q, double *D, double *basis,
                                   double *activation, int B, int N, int M, int F, int T, int K,
                                   int steps, int floor_kind, double floor_eps, void *workspace,
                                   size_t workspace_bytes, int *info, double *handover,
                                   int *handover_valid, void *stream) {
  SSSPY_REQUIRE(handover && handover_valid, "fastmnmf_update_handover: bad argument");
  return fastmnmf_update_impl(X, C, Q, D, basis, activation, B, N, M, F, T, K, steps, floor_kind,
                              floor_eps, workspace, workspace_bytes, info, handover, handover_valid,
                              as_stream(stream));
}

int ssspy_fastmnmf_deferred_logdet_slots(int B, int N, int M, int F, int T, int K) {
  if (B <= 0 || N <= 0 || M <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  return (mnmf_tiled(N, M) && ip1_small_shape(B, F, M)) ? (F + 15) / 16 : 1;
}

int ssspy_fastmnmf_update_handover_logdet(const void *X, const void *C, void *Q, double *D,
                                          double *basis, double *activation, int B, int N, int M,
                                          int F, int T, int K, int steps, int floor_kind,
                                          double floor_eps, void *workspace, size_t workspace_bytes,
                                          int *info, double *handover, int *handover_valid,
                                          double *logdet, long long logdet_stride, void *stream) {
  SSSPY_REQUIRE((handover == nullptr) == (handover_valid == nullptr) && logdet &&
                    logdet_stride >= B,
                "fastmnmf_update_handover_logdet: bad argument");
  return fastmnmf_update_impl(X, C, Q, D, basis, activation, B, N, M, F, T, K, steps, floor_kind,
                              floor_eps, workspace, workspace_bytes, info, handover, handover_valid,
                              as_stream(stream), logdet, logdet_stride);
}

int ssspy_fastmnmf_diagonalizer_covariance(const void *X, const double *D, const double *basis,
                                           const double *activation, void *U, int B, int N, int M,
                                           int F, int T, int K, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && D && basis && activation && U && B > 0, "fastmnmf_diagonalizer_covariance: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "fastmnmf_diagonalizer_covariance: bad n_basis");
  if (!mnmf_tiled(N, M))
    return fail(SSSPY_ERR_UNSUPPORTED,
                "fastmnmf_diagonalizer_covariance: beyond 4 sources / channels use "
                "ssspy_fastmnmf_weights + ssspy_weighted_covariance");
  // workspace (optional, ssspy_fastmnmf_workspace_bytes): with it the tuned pass of the fused update
  // (its split items park their partial sums there); without, the generic unsplit kernel -- round 5:
  // the IP2 diagonaliser took the generic one, 491 against 254 us at 32 mixtures of configs[3]
  double *tail = nullptr;
  if (workspace) {
    const MnmfWs w = mnmf_ws(B, N, M, F, T, K);
    SSSPY_REQUIRE(workspace_bytes >= w.total, "fastmnmf_diagonalizer_covariance: workspace too small");
    tail = (double *)((char *)workspace + w.tail);
  }
  MNMF_DISPATCH(N, mnmf_wcov, X, D, basis, activation, U, B, M, F, T, K, tail, (int *)nullptr,
                (long long *)nullptr, as_stream(stream));
}

int ssspy_fastmnmf_weights(const void *X, const void *Q, const double *D, const double *basis,
                           const double *activation, double *weights, int B, int N, int M, int F,
                           int T, int K, void *stream) {
  SSSPY_REQUIRE(X && Q && D && basis && activation && weights && B > 0, "fastmnmf_weights: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "fastmnmf_weights: bad n_basis");
  return fmnmf_generic_weights(X, Q, D, basis, activation, weights, B, N, M, F, T, K,
                               as_stream(stream));
}

// scratch of the loss entry points (per-block / per-wave shares, added up in a fixed order)
static size_t fastmnmf_loss_ws(int B, int N, int M, int F, int T) {
  if (!mnmf_tiled(N, M)) return fmnmf_generic_loss_ws_bytes(B, F, T);
  switch (N) {
    case 2: return mnmf_loss_ws_bytes_n2(B, F);
    case 3: return mnmf_loss_ws_bytes_n3(B, F);
    default: return mnmf_loss_ws_bytes_n4(B, F);
  }
}

size_t ssspy_fastmnmf_loss_workspace_bytes(int B, int N, int M, int F, int T) {
  if (B <= 0 || N <= 0 || M <= 0 || F <= 0 || T <= 0) return 0;
  return fastmnmf_loss_ws(B, N, M, F, T);
}

int ssspy_fastmnmf_loss_data(const void *X, const void *Q, const double *D, const double *basis,
                             const double *activation, double *out, int B, int N, int M, int F,
                             int T, int K, void *workspace, size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && Q && D && basis && activation && out && B > 0, "fastmnmf_loss_data: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "fastmnmf_loss_data: bad n_basis");
  SSSPY_REQUIRE(N >= 1 && N <= SSSPY_MAX_SOURCES && M >= 2 && M <= 8,
                "fastmnmf_loss_data: n_sources in [1, 8], n_channels in [2, 8]");
  SSSPY_REQUIRE(workspace && workspace_bytes >= fastmnmf_loss_ws(B, N, M, F, T),
                "fastmnmf_loss_data: workspace too small (ssspy_fastmnmf_loss_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  if (!mnmf_tiled(N, M))
    return fmnmf_generic_loss(X, Q, D, basis, activation, out, workspace, B, N, M, F, T, K, st);
  MNMF_DISPATCH(N, mnmf_loss, X, Q, D, basis, activation, out, workspace, B, M, F, T, K, st);
}

int ssspy_fastmnmf_loss_data_handover(const double *D, const double *basis,
                                      const double *activation, const double *handover,
                                      double *out, int B, int N, int M, int F, int T, int K,
                                      void *workspace, size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(D && basis && activation && handover && out && B > 0,
                "fastmnmf_loss_data_handover: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "fastmnmf_loss_data_handover: bad n_basis");
  if (handover_ok(B, N, M, F, T, K) != 1)
    return fail(SSSPY_ERR_UNSUPPORTED, "fastmnmf_loss_data_handover: no hand-over for this shape");
  SSSPY_REQUIRE(workspace && workspace_bytes >= fastmnmf_loss_ws(B, N, M, F, T),
                "fastmnmf_loss_data_handover: workspace too small "
                "(ssspy_fastmnmf_loss_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  const double *pscale = handover + (size_t)B * M * F * T;
  MNMF_DISPATCH(N, mnmf_loss_handover, D, basis, activation, handover, pscale, out, workspace, B, M,
                F, T, K, st);
}

int ssspy_fastmnmf_loss_handover_slots(int B, int N, int M, int F, int T, int K) {
  if (B <= 0 || N <= 0 || M <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  if (handover_ok(B, N, M, F, T, K) != 1) return 0;
  MNMF_DISPATCH(N, mnmf_loss_handover_slots, B, F, T);
}

int ssspy_fastmnmf_loss_data_handover_slots(const double *D, const double *basis,
                                            const double *activation, const double *handover,
                                            double *slots, long long slot_stride, int B, int N,
                                            int M, int F, int T, int K, void *stream) {
  SSSPY_REQUIRE(D && basis && activation && handover && slots && B > 0 && slot_stride >= B &&
                    slot_stride < (1ll << 31),
                "fastmnmf_loss_data_handover_slots: bad argument");
  if (handover_ok(B, N, M, F, T, K) != 1)
    return fail(SSSPY_ERR_UNSUPPORTED, "fastmnmf_loss_data_handover: no hand-over for this shape");
  const double *pscale = handover + (size_t)B * M * F * T;
  MNMF_DISPATCH(N, mnmf_loss_handover_raw, D, basis, activation, handover, pscale, slots,
                slot_stride, B, M, F, T, K, as_stream(stream));
}

int ssspy_fastmnmf_separate(const void *X, const void *Q, const double *D, const double *basis,
                            const double *activation, void *Y, int B, int N, int M, int F, int T,
                            int K, int reference_id, int floor_kind, double floor_eps,
                            void *workspace, size_t workspace_bytes, int *info, void *stream) {
  SSSPY_REQUIRE(X && Q && D && basis && activation && Y && B > 0, "fastmnmf_separate: bad argument");
  SSSPY_REQUIRE(reference_id >= 0 && reference_id < M, "fastmnmf_separate: bad reference_id");
  const MnmfWs w = mnmf_ws(B, N, M, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "fastmnmf_separate: workspace too small");
  void *Qinv = (char *)workspace + w.qinv;
  if (!mnmf_tiled(N, M))
    return fmnmf_generic_separate(X, Q, Qinv, D, basis, activation, Y, B, N, M, F, T, K,
                                  reference_id, floor_kind, floor_eps, info,
                                  (int *)((char *)workspace + w.qbuf), as_stream(stream));
  // (the per-bin row powers' scratch is idle here: B F ints of it flag the bins for the general kernel)
  MNMF_DISPATCH(N, mnmf_separate, X, Q, Qinv, D, basis, activation, Y, B, M, F, T, K, reference_id,
                floor_kind, floor_eps, info, (int *)((char *)workspace + w.qbuf), as_stream(stream));
}

int ssspy_fastmnmf_separate_eig(const void *X, const void *Q, const double *D, const double *basis,
                                const double *activation, void *Y, int B, int N, int M, int F,
                                int T, int K, int reference_id, int stage, double *lam, void *P,
                                void *workspace, size_t workspace_bytes, int *info, void *stream) {
  SSSPY_REQUIRE(X && Q && D && basis && activation && lam && P && B > 0,
                "fastmnmf_separate_eig: bad argument");
  SSSPY_REQUIRE((stage == 1 || stage == 2) && (stage == 1 || Y), "fastmnmf_separate_eig: bad stage");
  SSSPY_REQUIRE(reference_id >= 0 && reference_id < M, "fastmnmf_separate_eig: bad reference_id");
  const MnmfWs w = mnmf_ws(B, N, M, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "fastmnmf_separate_eig: workspace too small");
  // (stage 1 leaves Q^-1 in the workspace; stage 2 reads it: the same workspace, untouched between)
  return fmnmf_generic_separate_eig(X, Q, (char *)workspace + w.qinv, D, basis, activation, Y, B, N,
                                    M, F, T, K, reference_id, stage, lam, P, info,
                                    as_stream(stream));
}

}  // extern "C"
