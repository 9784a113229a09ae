// C-ABI entry points of the Gauss-ILRMA path: dispatch over n_sources to the per-N MFMA
// translation units (ilrma_kernels.hip, built with -DSSSPY_N=n) plus the small
// normalisation / weight kernels that do not depend on N at compile time.
#include <cstdlib>

#include "common.hpp"
#include "ilrma_params.hpp"
#include "tail_plan.hpp"
#include "wide_n.hpp"

namespace ssspy {

#define DECL_N(n)                                                                               \
  int ilrma_basis_n##n(const void *, const void *, const double *, double *, const double *,   \
                       IlrmaDims, hipStream_t);                                                \
  int ilrma_activation_n##n(const void *, const void *, const double *, const double *, double *, \
                            int, IlrmaDims, hipStream_t);                                      \
  int ilrma_wcov_n##n(const void *, const void *, const double *, const double *, void *,       \
                      IlrmaDims, hipStream_t);                                                 \
  size_t ilrma_loss_ws_bytes_n##n(int, int);                                                   \
  int ilrma_loss_n##n(const void *, const void *, const double *, const double *, double *,     \
                      void *, IlrmaDims, hipStream_t);
DECL_N(2) DECL_N(3) DECL_N(4) DECL_N(5) DECL_N(6) DECL_N(7) DECL_N(8)
#undef DECL_N

// throughput variants (ilrma_fast.hip): n_basis <= 64 (two / four k tiles above 16 / 32), n_sources <= 4, models
// of fast_model_id()
#define DECL_FAST(n)                                                                           \
  int ilrma_fast_basis_n##n(const void *, const void *, const double *, double *, const double *, \
                            int, int, int, int, int, double, double *, int, double, int,        \
                            double *, void *, int, hipStream_t, long long);                     \
  int ilrma_fast_basis_loss_slots_n##n(int, int, int);                                          \
  size_t ilrma_fast_loss_ws_bytes_n##n(int, int);                                               \
  int ilrma_fast_activation_n##n(const void *, const void *, const double *, const double *,   \
                                 double *, int, int, int, int, int, int, double, int,           \
                                 hipStream_t);                                                  \
  int ilrma_fast_wcov_n##n(const void *, const void *, const double *, const double *, void *, \
                           int, int, int, int, void *, int, double, int, double, hipStream_t,  \
                           int *, int *);                                                      \
  int ilrma_fast_loss_n##n(const void *, const void *, const double *, const double *, double *, \
                           void *, int, int, int, int, int, double, hipStream_t);
DECL_FAST(2) DECL_FAST(3) DECL_FAST(4)
#undef DECL_FAST

// latency variants for a handful of mixtures (ilrma_small.hip): n_basis <= 16, n_sources <= 4
#define DECL_SMALL(n)                                                                           \
  size_t ilrma_small_scratch_n##n(int, int, int, int);                                            \
  int ilrma_small_activation_n##n(const void *, const void *, const double *, double *, int, int, \
                                  int, int, int, double, double *, int, double, int,              \
                                  hipStream_t);                                                   \
  int ilrma_small_ip1_n##n(const void *, int, int, long long, const void *, void *, int, int, int, \
                           double, double *, int *, hipStream_t);                                 \
  int ilrma_small_ip1_logdet_n##n(const void *, int, int, long long, const void *, void *, int,   \
                                  int, int, double, double *, int *, double *, long long,         \
                                  hipStream_t);                                                   \
  int ilrma_small_norm_n##n(void *, double *, const double *, int, int, int, double, int, double, \
                            hipStream_t);
DECL_SMALL(2) DECL_SMALL(3) DECL_SMALL(4)
#undef DECL_SMALL

#define ILRMA_FAST_DISPATCH(N_, fn, ...)             \
  switch (N_) {                                      \
    case 2: return fn##_n2(__VA_ARGS__);             \
    case 3: return fn##_n3(__VA_ARGS__);             \
    default: return fn##_n4(__VA_ARGS__);            \
  }

// the tuned kernels: n_sources <= 4, n_basis <= 16 and one of the models ilrma_fast.hip carries
// (its FM_* ids): Gauss at domain 2 (MM or ME), 1 or any other value in (0, 2), Student-t and GGD
// at domain 2;
// `source_model` may carry the SSSPY_SOURCE_ME flag.  -1: generic kernels.
static inline int fast_model_id(double domain, int source_model) {
  const int base = source_model & 0xff;
  const bool me = (source_model & SSSPY_SOURCE_ME) != 0;
  if (domain == 2.0) {
    if (base == SSSPY_SOURCE_GAUSS) return 0;
    if (base == SSSPY_SOURCE_T) return 1;
    if (base == SSSPY_SOURCE_GGD) return 2;
  }
  if (domain == 1.0 && base == SSSPY_SOURCE_GAUSS && !me) return 3;
  // Gauss at any other domain in (0, 2): the powers R^((p+2)/p), R^(2/p) as exp2(e log2 R)
  if (base == SSSPY_SOURCE_GAUSS && !me && domain > 0.0 && domain < 2.0) return 4;
  return -1;
}
// what the tuned kernels take as their model parameter: dof (t), beta (GGD), the domain (id 4)
static inline double fast_model_param(double domain, int source_model, double model_param) {
  return fast_model_id(domain, source_model) == 4 ? domain : model_param;
}
// (one channel of a mixture must fit the 32-bit offset of a buffer descriptor: F T 16 bytes < 4 GiB)
static inline bool fast_path(int N, int F, int T, int K, double domain,
                             int source_model = SSSPY_SOURCE_GAUSS) {
  static const bool disabled = std::getenv("SSSPY_AMD_NO_FAST") != nullptr;
  return !disabled && fast_model_id(domain, source_model) >= 0 && N >= 2 && N <= 4 && K <= 64 &&
         (long long)F * T * 16 < (1ll << 32);
}
static inline int is_me(int source_model) { return (source_model & SSSPY_SOURCE_ME) ? 1 : 0; }

// The latency kernels (ilrma_small.hip) serve batches whose bin tiles do not fill the chip with the
// throughput kernels' 64-bin work items: B * ceil(F / 16) <= 350, i.e. up to 5 mixtures of 1025 bins --
// round 4: with the cost-based tail plan the throughput kernels win from 6 mixtures on, 29.7 k
// against 27.4 k mixture-iterations/s at 9, benchmarks/batch_sweep.py; it was 640:
// up to 9 mixtures of 1025 bins).
static inline bool small_path(int B, int N, int F, int T, int K, double domain,
                              int source_model = SSSPY_SOURCE_GAUSS) {
  const long long max_items = 350;
  return K <= 16 && fast_path(N, F, T, K, domain, source_model) &&
         (long long)B * ((F + 15) / 16) <= max_items;
}
static inline size_t small_scratch(int B, int N, int F, int T, int K) {
  switch (N) {
    case 2: return ilrma_small_scratch_n2(B, F, T, K);
    case 3: return ilrma_small_scratch_n3(B, F, T, K);
    case 4: return ilrma_small_scratch_n4(B, F, T, K);
    default: return 0;
  }
}

// More than 4 sources on the tuned NMF passes: the multiplicative updates of source n need only
// |y_n|^2 and (T_n, V_n), and (B, N, ...) tensors are (B N / G, G, ...) tensors in memory, so a wide
// mixture is walked as N / G "mixtures" of G sources over the separated spectrogram y = W x (formed
// once by ssspy_separate; the ISS / IPA state is y already).  Returns the group size G (4, 3 or 2)
// or 0 when the shape has none (N <= 4, N = 5 or 7, or the model / n_basis is off the tuned path).
static inline int source_group(int N, int F, int T, int K, double domain, int source_model) {
  if (N <= 4) return 0;
  for (int G = 4; G >= 2; --G)
    if (N % G == 0 && fast_path(G, F, T, K, domain, source_model)) return G;
  return 0;
}

// shapes that may take the wide-basis path of wide_basis.hip (its buffers are sized for them; the
// source model, which the workspace query does not know, decides at the call)
// The register-tiled passes win up to 32 bases (16: 1.0 ms, 32: 1.6 ms per iteration at 32 mixtures of
// the configs[1] shape); from 33 on their four-k-tile form (4.1-4.7 ms) loses to the dense products
// (3.1-3.2 ms; 80: 4.1, 128: 4.6, 256: 7.1, 1024: 22.8 -- benchmarks/wide_basis.py, round 4).
static inline bool wide_basis_shape(int N, int K) {
  return K >= 33 || N > SSSPY_MAX_SOURCES;
}
// The general form (any source count above 4, e.g. 5 or 7): the B N sources of the batch, in memory
// order, are cut into at most three runs of `count` groups of G sources each -- groups of 4 and one
// or two closing groups of 3 / 2 sources -- and every run is one launch of the tuned kernels on its
// slice of y, T and V.  Returns the number of runs (0: not on the grouped path).
struct SourceRun {
  long long first;  // first source of the run in the flat (B N) order
  int count, G;     // `count` groups of G sources
};
static inline int source_runs(int B, int N, int F, int T, int K, double domain, int source_model,
                              SourceRun (&run)[3]) {
  if (N <= 4) return 0;
  if (K > 32 && wide_basis_shape(N, K)) return 0;  // the dense products win from 33 bases on
  if (const int G = source_group(N, F, T, K, domain, source_model)) {
    run[0] = SourceRun{0, B * (N / G), G};
    return 1;
  }
  for (int G = 2; G <= 4; ++G)
    if (!fast_path(G, F, T, K, domain, source_model)) return 0;
  const long long S = (long long)B * N;  // >= 5
  const int r = (int)(S % 4);
  const int tail = r == 0 ? 0 : (r == 1 ? 5 : r);  // 5 = 3 + 2
  int n = 0;
  if (S - tail > 0) run[n++] = SourceRun{0, (int)((S - tail) / 4), 4};
  if (tail == 5) {
    run[n++] = SourceRun{S - 5, 1, 3};
    run[n++] = SourceRun{S - 2, 1, 2};
  } else if (tail) {
    run[n++] = SourceRun{S - tail, 1, tail};
  }
  return n;
}
static inline bool grouped_path(int B, int N, int F, int T, int K, double domain, int source_model) {
  SourceRun run[3];
  return source_runs(B, N, F, T, K, domain, source_model, run) > 0;
}

static inline int check_model(int source_model, double param, double domain = 2.0) {
  const int model = source_model & 0xff;
  const bool me = (source_model & SSSPY_SOURCE_ME) != 0;
  if ((source_model & ~(0xff | SSSPY_SOURCE_ME)) != 0)
    return fail(SSSPY_ERR_BADARG, "bad source model flags");
  if (me && (domain != 2.0 || model == SSSPY_SOURCE_GGD))
    return fail(SSSPY_ERR_BADARG, "ME source updates need domain == 2 and a Gauss or t model");
  if (model == SSSPY_SOURCE_GAUSS) return SSSPY_OK;
  if (model == SSSPY_SOURCE_T && param > 0.0) return SSSPY_OK;
  if (model == SSSPY_SOURCE_GGD && param > 0.0 && param < 2.0) return SSSPY_OK;
  return fail(SSSPY_ERR_BADARG, "bad source model / model_param (t: dof > 0, GGD: 0 < beta < 2)");
}

#define ILRMA_DISPATCH(N_, fn, ...)                                                  \
  switch (N_) {                                                                      \
    case 2: return fn##_n2(__VA_ARGS__);                                             \
    case 3: return fn##_n3(__VA_ARGS__);                                             \
    case 4: return fn##_n4(__VA_ARGS__);                                             \
    case 5: return fn##_n5(__VA_ARGS__);                                             \
    case 6: return fn##_n6(__VA_ARGS__);                                             \
    case 7: return fn##_n7(__VA_ARGS__);                                             \
    case 8: return fn##_n8(__VA_ARGS__);                                             \
    default: return fail(SSSPY_ERR_UNSUPPORTED, "ILRMA: n_sources must be in [2, 8]"); \
  }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int ngroups_of(int N) { return N <= 4 ? 1 : (N + 1) / 2; }

// number of bin chunks the activation pass splits into (partials are summed by the finalize
// kernel): enough blocks to occupy the chip for small batches, one chunk for large ones.
static inline int act_chunks(int B, int N, int F, int T, int K) {
  const long long blocks0 = (long long)B * ngroups_of(N) * ((T + 63) / 64) * ((K + 15) / 16);
  const int ntiles = (F + 15) / 16;
  // (round 4: chunk count by the cost search of tail_plan.hpp instead of "just fill one round";
  //  24 mixtures of the configs[1] shape: 5 chunks in two short rounds instead of 3 in two long ones)
  const int want = best_split(blocks0, ntiles, 512, 16, 2048);
  // a single chunk finishes in place (no partial sums): keep it whenever the batch fills the chip
  return blocks0 >= 2048 ? 1 : want;
}

// (the latency kernel's partials when the shape can take it: the workspace is sized without knowing
// the source model, so Gauss at domain 2 stands for "any model of the tuned path")
static inline size_t act_part_bytes(int B, int N, int F, int T, int K) {
  const size_t ap = (N >= 2 && N <= 4 && small_path(B, N, F, T, K, 2.0)) ? small_scratch(B, N, F, T, K) : 0;
  const size_t base = (size_t)B * act_chunks(B, N, F, T, K) * N * 2 * K * T * sizeof(double);
  return align256(base > ap ? base : ap);
}
static inline size_t basis_tmp_bytes(int B, int N, int F, int K) {
  return K > 16 ? align256((size_t)B * N * F * K * sizeof(double)) : 0;
}
static inline size_t qbuf_bytes(int B, int N, int F) {
  return align256((size_t)B * F * N * sizeof(double));
}

int ip1_with_power(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N,
                   int floor_kind, double floor_eps, int *info, hipStream_t st);
int separate_power(const void *X, const void *W, double *P, int B, int N, int F, int T,
                   hipStream_t st);
// wide_cov.hip: weighted covariance of 5..8 channels on the matrix cores
bool wide_weighted_cov_ok(int N, int S, int F, int T, int kind);
int wide_weighted_cov(const void *A, const double *weight, int kind, void *U, int B, int N, int S,
                      int F, int T, hipStream_t st);

// scratch of the bin-major fast kernels (basis, covariance): partial sums of the at most 1024
// split blocks of the closing scheduling rounds (TailPlan in ilrma_fast.hip)
// (wide mixtures run them in groups of at most 4 sources, see source_group())
// (the wide variants, 16 < n_basis <= 64: at most 256 split blocks, each leaving one 16-k record per
//  k tile it accumulates -- up to 4)
static inline size_t basis_part_bytes(int N) {
  const int G = N < 4 ? N : 4;
  return align256((size_t)1024 * G * 64 * 16 * 2 * sizeof(double));
}
// scratch of the deterministic loss sums: the larger of what the tuned kernels (by-product of the
// basis pass, loss pass) and the generic loss kernel need
size_t wb_loss_ws_bytes(int B, int N, int F, int T);  // wide_basis.hip
// K, with_filter: the wide-basis loss (n_basis above 16, or more than 8 sources) parks one slot per
// 64 x 64 tile of every source and, when a filter is given, |W x|^2 (B N F T doubles) -- only
// then (round 4 added both terms for every shape: 2.1 GB idle at the headline batch, twice)
static inline size_t loss_slots_bytes(int B, int N, int F, int T, int K, bool with_filter) {
  auto generic = [&]() -> size_t {
    switch (N) {
      case 2: return ilrma_loss_ws_bytes_n2(B, F);
      case 3: return ilrma_loss_ws_bytes_n3(B, F);
      case 4: return ilrma_loss_ws_bytes_n4(B, F);
      case 5: return ilrma_loss_ws_bytes_n5(B, F);
      case 6: return ilrma_loss_ws_bytes_n6(B, F);
      case 7: return ilrma_loss_ws_bytes_n7(B, F);
      case 8: return ilrma_loss_ws_bytes_n8(B, F);
      default: return 0;
    }
  };
  auto tuned = [&]() -> size_t {
    switch (N) {
      case 2: return ilrma_fast_loss_ws_bytes_n2(B, F);
      case 3: return ilrma_fast_loss_ws_bytes_n3(B, F);
      case 4: return ilrma_fast_loss_ws_bytes_n4(B, F);
      default: return 0;
    }
  };
  const size_t a = generic(), b = tuned();
  size_t c = rt_sources_ok(N) ? rt_ilrma_loss_ws_bytes(B, N, F) : 0;
  // (the wide-basis loss: one slot per 64 x 64 tile of every source; a y = W x buffer when a filter
  //  is given)
  if (K > 16 || rt_sources_ok(N)) {
    const size_t g = align256(wb_loss_ws_bytes(B, N, F, T)) +
                     (with_filter ? align256((size_t)B * N * F * T * sizeof(double)) : 0);
    c = c > g ? c : g;
  }
  return align256(a > b ? (a > c ? a : c) : (b > c ? b : c));
}
static inline size_t u_part_bytes(int N) {
  return N <= 4 ? align256((size_t)1024 * 64 * N * N * N * 2 * sizeof(double)) : 0;
}
int row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st);

// V <- floor(V * (sum_chunks num / sum_chunks den)^(p/(p+2)))
__global__ __launch_bounds__(256) void k_ilrma_activation_finalize(double *act,
                                                                   const double *__restrict__ part,
                                                                   int N, int K, int T,
                                                                   int nchunks, IlrmaDims d) {
  const int b = blockIdx.z, n = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over K*T
  if (e >= (long long)K * T) return;
  double sn = 0.0, sd = 0.0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const long long base = ((((long long)b * nchunks + ch) * N + n) * 2) * K * T;
    sn += part[base + e];
    sd += part[base + (long long)K * T + e];
  }
  double *dst = act + ((long long)b * N + n) * K * T + e;
  *dst = apply_floor(mm_ratio_pow(sn, sd, d) * (*dst), d.floor_kind, d.floor_eps);
}


// ------------------------------------------------------------------ power normalisation (filter)
// psi_n^2 = (1/F) sum_i q[i][n], q[i][n] = Re(w_in C_i w_in^H) = mean_j |y_nij|^2.
// q comes from the IP1 kernel (fused iteration) or from k_row_power (stand-alone call).
// grid: (ceil(F/64), B).  Every block folds q over all bins (fixed order: deterministic), then
// scales the demixing rows and basis rows of its own 64 bins.
// basis == NULL (partitioning): only W is scaled and psi is published to psi_out (B, N)
__global__ __launch_bounds__(256) void k_norm_scale(c128 *W, double *basis,
                                                    const double *__restrict__ qbuf, int N, int F,
                                                    int K, double p, int floor_kind, double eps,
                                                    double *psi_out) {
  __shared__ double wsum[4][SSSPY_RT_MAX_SOURCES];
  __shared__ double psi[SSSPY_RT_MAX_SOURCES];
  const int b = blockIdx.y;
  const double *qb = qbuf + (long long)b * F * N;
  // thread t walks the flat (bin, n) array with a stride that keeps its source index fixed
  {
    const int n = threadIdx.x % N;
    const int stride = (blockDim.x / N) * N;
    double local = 0.0;
    if (threadIdx.x < stride)
      for (int e = threadIdx.x; e < F * N; e += stride) local += qb[e];
    // lanes with equal (lane % N) hold the same source; fold them inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n2 = 0; n2 < N; ++n2) {
      const double mine = (n == n2) ? local : 0.0;
      const double tot = wave_sum(mine);
      if (lane == 0) wsum[wave][n2] = tot;
    }
    __syncthreads();
    if (threadIdx.x < N) {
      double v = 0.0;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += wsum[wv][threadIdx.x];
      v = v / (double)F;
      v = v < 0.0 ? 0.0 : v;
      psi[threadIdx.x] = apply_floor(sqrt(v), floor_kind, eps);
    }
  }
  __syncthreads();
  const int i0 = blockIdx.x * 64;
  const int nb = min(64, F - i0);
  c128 *Wb = W + ((long long)b * F + i0) * N * N;
  for (int e = threadIdx.x; e < nb * N * N; e += blockDim.x) {
    const int n = (e / N) % N;
    const c128 v = Wb[e];
    Wb[e] = cmake(v.x / psi[n], v.y / psi[n]);
  }
  if (psi_out && blockIdx.x == 0 && threadIdx.x < N) psi_out[b * N + threadIdx.x] = psi[threadIdx.x];
  if (!basis) return;
  for (int n = 0; n < N; ++n) {
    const double pp = (p == 2.0) ? psi[n] * psi[n] : pow(psi[n], p);
    double *Tb = basis + (((long long)b * N + n) * F + i0) * K;
    for (int e = threadIdx.x; e < nb * K; e += blockDim.x) Tb[e] = Tb[e] / pp;
  }
}

// ------------------------------------------------------------------ power normalisation (output)
// acc[b][n][blk] = sum of |y|^2 over 16 bin rows: one slot per block, no fp64 atomics (the
// normalised state is the same on every run); the consumer adds the slots in order
__global__ __launch_bounds__(256) void k_output_power(const c128 *__restrict__ Y, double *acc,
                                                      int N, int F, int T) {
  // grid: (ceil(F / 16), N, B): 16 consecutive bin rows (one contiguous run) per block
  __shared__ double scratch[4];
  const int n = blockIdx.y, b = blockIdx.z;
  const int i0 = blockIdx.x * 16;
  const long long len = (long long)min(16, F - i0) * T;
  const c128 *run = Y + (((long long)b * N + n) * F + i0) * T;
  double local = 0.0;
  for (long long e = threadIdx.x; e < len; e += blockDim.x) local += cabs2(run[e]);
  const double total = block_sum(local, scratch);
  if (threadIdx.x == 0) acc[((long long)b * N + n) * gridDim.x + blockIdx.x] = total;
}

// psi of source n from the `slots` partial powers of (b, n), every thread the same sum
__device__ __forceinline__ double psi_from_slots(const double *__restrict__ acc, int slots, int F,
                                                 int T, int floor_kind, double eps) {
  double v = 0.0;
  for (int k = 0; k < slots; ++k) v += acc[k];
  v = v / ((double)F * (double)T);
  return apply_floor(sqrt(v), floor_kind, eps);
}

__global__ __launch_bounds__(256) void k_ilrma_normalize_output(c128 *Y, double *basis,
                                                                const double *__restrict__ acc,
                                                                int slots, int N, int F, int T,
                                                                int K, double p, int floor_kind,
                                                                double eps, double *psi_out,
                                                                double *logdet) {
  const int i = blockIdx.x, n = blockIdx.y, b = blockIdx.z;
  const double psi = psi_from_slots(acc + ((long long)b * N + n) * slots, slots, F, T, floor_kind, eps);
  if (psi_out && i == 0 && threadIdx.x == 0) psi_out[b * N + n] = psi;
  // (row n of every implied demixing matrix is divided by psi: sum_i log|det W_i| moves by
  // -F sum_n log psi_n; ONE thread per mixture adds all N terms, in source order)
  if (logdet && i == 0 && n == 0 && threadIdx.x == 0) {
    double delta = 0.0;
    for (int m = 0; m < N; ++m)
      delta += log(psi_from_slots(acc + ((long long)b * N + m) * slots, slots, F, T, floor_kind, eps));
    logdet[b] -= (double)F * delta;
  }
  c128 *row = Y + (((long long)b * N + n) * F + i) * T;
  for (int j = threadIdx.x; j < T; j += blockDim.x) {
    c128 y = row[j];
    row[j] = cmake(y.x / psi, y.y / psi);
  }
  if (!basis) return;
  const double pp = (p == 2.0) ? psi * psi : pow(psi, p);
  double *tr = basis + (((long long)b * N + n) * F + i) * K;
  for (int k = threadIdx.x; k < K; k += blockDim.x) tr[k] = tr[k] / pp;
}

// ------------------------------------------------------- partitioning (latent variables Z)
// Shared basis t (B,F,K) / activation v (B,K,T) are assigned to the sources by z (B,N,K):
// R_nij = sum_k z_nk t_ik v_kj.  Every kernel of the non-partitioned path runs unchanged on the
// expanded pair Teff[b,n,i,k] = z_nk t_ik, Vrep[b,n,k,j] = v_kj; the three parameter updates
// then recombine the per-source sums those kernels produce.
// ref: ssspy/bss/ilrma.py:297-327 (reconstruct_nmf), :1007-1049, :1094-1128, :1170-1204.
__global__ __launch_bounds__(256) void k_partition_expand(const double *__restrict__ basis,
                                                          const double *__restrict__ act,
                                                          const double *__restrict__ latent,
                                                          double *__restrict__ Teff,
                                                          double *__restrict__ Vrep, int N, int F,
                                                          int T, int K) {
  const int n = blockIdx.y, b = blockIdx.z;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const double *z = latent + ((long long)b * N + n) * K;
  if (e < (long long)F * K) Teff[((long long)b * N + n) * F * K + e] = z[e % K] * basis[(long long)b * F * K + e];
  if (e < (long long)K * T) Vrep[((long long)b * N + n) * K * T + e] = act[(long long)b * K * T + e];
}

// z_nk <- z_nk (sum_i t_ik S_nik / sum_i t_ik D_nik)^e, then every column of z sums to one.
// raw: (B,N,F,K,2) basis-type sums over frames with the shared v.  grid: (B), 256 threads.
__global__ __launch_bounds__(256) void k_partition_latent(const double *__restrict__ raw,
                                                          const double *__restrict__ basis,
                                                          double *latent, int N, int F, int K,
                                                          IlrmaDims d) {
  __shared__ double znew[SSSPY_MAX_SOURCES * SSSPY_MAX_PARTITION_BASIS];
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int n = e / K, k = e % K;
    double sn = 0.0, sd = 0.0;
    for (int i = 0; i < F; ++i) {
      const double t = basis[((long long)b * F + i) * K + k];
      const double *r = raw + ((((long long)b * N + n) * F + i) * K + k) * 2;
      sn = fma(t, r[0], sn);
      sd = fma(t, r[1], sd);
    }
    znew[e] = mm_ratio_pow(sn, sd, d) * latent[((long long)b * N + n) * K + k];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int k = e % K;
    double col = 0.0;
    for (int n = 0; n < N; ++n) col += znew[n * K + k];
    latent[(long long)b * N * K + e] = znew[e] / col;
  }
}

// t_ik <- floor(t_ik (sum_n z_nk S_nik / sum_n z_nk D_nik)^e).  one thread per (b, i, k)
__global__ __launch_bounds__(256) void k_partition_basis(const double *__restrict__ raw,
                                                         const double *__restrict__ latent,
                                                         double *basis, int N, int F, int K,
                                                         IlrmaDims d) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (i, k)
  if (e >= (long long)F * K) return;
  const int k = (int)(e % K);
  double sn = 0.0, sd = 0.0;
  for (int n = 0; n < N; ++n) {
    const double z = latent[((long long)b * N + n) * K + k];
    const double *r = raw + (((long long)b * N + n) * F * K + e) * 2;
    sn = fma(z, r[0], sn);
    sd = fma(z, r[1], sd);
  }
  double *dst = basis + (long long)b * F * K + e;
  *dst = apply_floor(mm_ratio_pow(sn, sd, d) * (*dst), d.floor_kind, d.floor_eps);
}

// v_kj <- floor(v_kj (sum_n num_nkj / sum_n den_nkj)^e); the per-source sums over bins were taken
// with Teff, so they already carry z_nk.  part: [b][chunk][n][2][K][T].  one thread per (b, k, j)
__global__ __launch_bounds__(256) void k_partition_activation(const double *__restrict__ part,
                                                              double *act, int N, int K, int T,
                                                              int nchunks, IlrmaDims d) {
  const int b = blockIdx.y;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (k, j)
  if (e >= (long long)K * T) return;
  double sn = 0.0, sd = 0.0;
  for (int ch = 0; ch < nchunks; ++ch)
    for (int n = 0; n < N; ++n) {
      const long long base = ((((long long)b * nchunks + ch) * N + n) * 2) * K * T;
      sn += part[base + e];
      sd += part[base + (long long)K * T + e];
    }
  double *dst = act + (long long)b * K * T + e;
  *dst = apply_floor(mm_ratio_pow(sn, sd, d) * (*dst), d.floor_kind, d.floor_eps);
}

// z <- (z / psi^p) / scale, t <- t * scale, scale_k = sum_n z_nk / psi_n^p.  grid: (B)
// ref: ssspy/bss/ilrma.py:418-427 (normalize_by_power, partitioning branch).
__global__ __launch_bounds__(256) void k_partition_normalize(double *basis, double *latent,
                                                             const double *__restrict__ psi, int N,
                                                             int F, int K, double p) {
  __shared__ double scale[SSSPY_MAX_PARTITION_BASIS];
  const int b = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
      const double ps = psi[b * N + n];
      s += latent[((long long)b * N + n) * K + k] / ((p == 2.0) ? ps * ps : pow(ps, p));
    }
    scale[k] = s;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N * K; e += blockDim.x) {
    const int n = e / K, k = e % K;
    const double ps = psi[b * N + n];
    double *z = latent + (long long)b * N * K + e;
    *z = (*z / ((p == 2.0) ? ps * ps : pow(ps, p))) / scale[k];
  }
  for (long long e = threadIdx.x; e < (long long)F * K; e += blockDim.x)
    basis[(long long)b * F * K + e] *= scale[e % K];
}

// ------------------------------------------------------------------------------ ISS weight
// varphi[b, n, i, j] = spatial_weight(|y|^2, (T V)_nij) (the 1 / R~ of the ISS sweep and of the wide
// covariance pass).  A wave owns 16 bins of one source and walks the frames 16 at a time: the tile of
// T V is ceil(K / 4) f64 MFMAs (basis rows in registers for K <= 16, the activation slab one 8-byte
// load per lane and k-step), and the D layout puts 16 consecutive frames of one bin in 16 lanes, so
// the writes are 128-byte rows.  grid: (ceil(F / 64) * nchunks) x N x B, wave w of block (g, chunk)
// owns bins [64 g + 16 w, +16) and the chunk's frame tiles (small batches split the frames so that
// the launch still has a few thousand waves: iss_weight_chunks()).
// (The previous thread-per-frame version fetched the basis entries one scalar load at a time:
// 0.31 ms for 0.54 GB of output.)
// (Ypow: |y|^2 handed in instead of y)
// GAUSS2: the Gauss model at domain 2 without the (a, b) mode -- varphi = 1 / R and nothing else
// compiled in (the general body carries the inlined pow() of every model and domain: 17 000 lines of
// ISA around a 300-line hot path)
template <bool GAUSS2 = false>
__global__ __launch_bounds__(256) void k_ilrma_iss_weight(const c128 *__restrict__ Y,
                                                          const double *__restrict__ Ypow,
                                                          const double *__restrict__ basis,
                                                          const double *__restrict__ act,
                                                          double *__restrict__ varphi, int N,
                                                          IlrmaDims d, int nchunks,
                                                          double *__restrict__ bout = nullptr) {
  // bout: the (a, b) mode of the wide-basis path (wide_basis.hip): varphi <- the numerator factor a
  // of the MM update, bout <- b = 1 / R (mm_weights) instead of the spatial weight
  const int n = blockIdx.y, b = blockIdx.z;
  const int F = d.F, T = d.T, K = d.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int group = blockIdx.x / nchunks, chunk = blockIdx.x - group * nchunks;
  const int i0 = group * 64 + wave * 16;
  if (i0 >= F) return;
  const int ntiles = (T + 15) >> 4, tpc = (ntiles + nchunks - 1) / nchunks;
  const int j_begin = chunk * tpc * 16, j_end = min(T, (chunk + 1) * tpc * 16);
  const double *Tn = basis + (((long long)b * N + n) * F) * K;
  const double *Vn = act + ((long long)b * N + n) * K * T;
  const long long row0 = ((long long)b * N + n) * F;
  const int ksteps = (K + 3) >> 2;
  // A operand: basis[bin i0 + c][4 ks + q]
  const int abin = min(i0 + c, F - 1);
  double ta[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int kk = 4 * ks + q;
    ta[ks] = kk < K ? Tn[(long long)abin * K + kk] : 0.0;
  }
  const bool need_y = !GAUSS2 && (d.model != SSSPY_SOURCE_GAUSS || bout != nullptr);
  // two frame tiles per pass: their loads and MFMA chains are independent (four: slower, 431 us
  // against 278 at 16 mixtures of 8 sources)
  constexpr int U = 2;
  for (int j0 = j_begin; j0 < j_end; j0 += 16 * U) {
    int jc[U];
    double4_t R[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      jc[u] = min(j0 + 16 * u + c, T - 1);
      R[u] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (ks < ksteps) {
        const int kk = 4 * ks + q;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double vb = kk < K ? Vn[(long long)kk * T + jc[u]] : 0.0;
          R[u] = mfma_f64(ta[ks], vb, R[u]);
        }
      }
    for (int ks = 4; ks < ksteps; ++ks) {  // n_basis above 16: both operands from memory
      const int kk = 4 * ks + q;
      const double av = kk < K ? Tn[(long long)abin * K + kk] : 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double vb = kk < K ? Vn[(long long)kk * T + jc[u]] : 0.0;
        R[u] = mfma_f64(av, vb, R[u]);
      }
    }
    // D: bin i0 + q + 4 r, frame j0 + 16 u + c
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bin = i0 + q + 4 * r, jf = j0 + 16 * u + c;
        if (bin < F && jf < j_end) {
          const long long e = (row0 + bin) * T + jf;
          if constexpr (GAUSS2) {
            varphi[e] = recip_weight(R[u][r]);
          } else {
            const double P = need_y ? (Ypow ? Ypow[e] : cabs2(Y[e])) : 0.0;
            if (bout) {
              double wa, wb;
              mm_weights(P, R[u][r], d, true, wa, wb);
              varphi[e] = wa;
              bout[e] = wb;
            } else {
              varphi[e] = spatial_weight(P, R[u][r], d);
            }
          }
        }
      }
  }
}

static inline int iss_weight_chunks(int B, int N, int F, int T) {
  const long long waves = (long long)B * N * ((F + 15) / 16);
  long long want = (4096 + waves - 1) / waves;
  const int ntiles = (T + 15) / 16;
  if (want > ntiles) want = ntiles;
  return want < 1 ? 1 : (int)want;
}

// acc[b, n] = sum_j frame_power[b, n, j]; grid: (N, B)
__global__ __launch_bounds__(256) void k_power_from_frames(const double *__restrict__ fp,
                                                           double *acc, int N, int T) {
  __shared__ double scratch[4];
  const int n = blockIdx.x, b = blockIdx.y;
  const double *row = fp + ((long long)b * N + n) * T;
  double local = 0.0;
  for (int j = threadIdx.x; j < T; j += blockDim.x) local += row[j];
  const double total = block_sum(local, scratch);
  if (threadIdx.x == 0) acc[b * N + n] = total;
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

// One scratch layout for every ILRMA entry point: callers pass the same buffer everywhere.
struct IlrmaWs {
  size_t act_part, btmp, qbuf, psi, lslots, bpart, upart, praw, ybuf, wbuf, gb, gnd, total;
};
static inline IlrmaWs ilrma_ws(int B, int N, int F, int T, int K) {
  IlrmaWs w;
  size_t off = 0;
  w.act_part = off;
  off += act_part_bytes(B, N, F, T, K);
  w.btmp = off;
  off += basis_tmp_bytes(B, N, F, K);
  w.qbuf = off;
  off += qbuf_bytes(B, N, F);
  w.psi = off;
  off += align256((size_t)B * N * sizeof(double));
  w.lslots = off;  // per-wave shares of a loss, folded in a fixed order (no fp64 atomics)
  off += loss_slots_bytes(B, N, F, T, K, true);
  w.bpart = off;
  off += basis_part_bytes(N);
  w.upart = off;
  off += u_part_bytes(N);
  w.praw = off;  // (num, den) basis sums of the partitioned updates
  off += align256((size_t)B * N * F * K * 2 * sizeof(double));
  const bool wb = wide_basis_shape(N, K);
  w.ybuf = off;  // y = W x of a wide mixture (more than 4 sources), see source_group()
  off += (N > 4 || wb) ? align256((size_t)B * N * F * T * 2 * sizeof(double)) : 0;
  w.wbuf = off;  // varphi (B, N, F, T) of a wide mixture's covariance pass (wide_cov.hip); a of wide_basis
  off += (N > 4 || wb) ? align256((size_t)B * N * F * T * sizeof(double)) : 0;
  w.gb = off;    // wide-basis path: b = 1 / R (B, N, F, T)
  off += wb ? align256((size_t)B * N * F * T * sizeof(double)) : 0;
  w.gnd = off;   // wide-basis path: (num, den) of the products
  off += wb ? align256((size_t)2 * B * N * (F > T ? F : T) * K * sizeof(double)) : 0;
  w.total = off;
  return w;
}

size_t ssspy_ilrma_workspace_bytes(int B, int N, int F, int T, int K) {
  if (B <= 0 || N <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  return ilrma_ws(B, N, F, T, K).total;
}

extern "C++" {
namespace ssspy {  // wide_basis.hip
int wb_update_basis(const double *a, const double *b, double *basis, const double *activation,
                    double *nd, int BN, int F, int T, int K, const IlrmaDims &d, hipStream_t st);
int wb_update_activation(const double *a, const double *b, const double *basis, double *activation,
                         double *nd, int BN, int F, int T, int K, const IlrmaDims &d,
                         hipStream_t st);
int wb_tv_weights(int mode, const double *basis, const double *activation, const double *ypow,
                  const void *y, double *out0, double *out1, int BN, int F, int T, int K,
                  const IlrmaDims &d, hipStream_t st);
int wb_loss_data(const double *basis, const double *activation, const double *ypow, const void *y,
                 double *out, void *ws, int B, int N, int F, int T, int K, const IlrmaDims &d,
                 hipStream_t st);
}  // namespace ssspy
}  // extern "C++"

// a = numerator factor, b = 1 / R of the MM updates for the wide-basis path: |y|^2 from the filter
// (ybuf), from a power input, or from y itself; the weight kernel's (a, b) mode writes wbuf / gb
static int wb_weights(const void *X, const void *W, bool x_is_power, const double *basis,
                      const double *activation, int N, const IlrmaDims &d, char *ws, size_t ybuf,
                      size_t wbuf, size_t gb, hipStream_t st);

static IlrmaDims make_dims(int B, int F, int T, int K, double domain, int model, double mparam,
                           int floor_kind, double floor_eps) {
  return IlrmaDims{B, F, T, K, domain, model & 0xff, (model & SSSPY_SOURCE_ME) ? 1 : 0, mparam,
                   floor_kind, floor_eps, 0};
}

static int wb_weights(const void *X, const void *W, bool x_is_power, const double *basis,
                      const double *activation, int N, const IlrmaDims &d, char *ws, size_t ybuf,
                      size_t wbuf, size_t gb, hipStream_t st) {
  const c128 *Y = nullptr;
  const double *Ypow = nullptr;
  if (W) {
    int r = separate_power(X, W, (double *)(ws + ybuf), d.B, N, d.F, d.T, st);
    if (r) return r;
    Ypow = (const double *)(ws + ybuf);
  } else if (x_is_power) {
    Ypow = (const double *)X;
  } else {
    Y = (const c128 *)X;
  }
  // (T V as a tiled GEMM with the (a, b) map in its epilogue: the weight kernel's own walk fetches
  //  both operands from memory at every k-step -- 1.75 ms against 0.45 at n_basis 128, 32 mixtures)
  return wb_tv_weights(1, basis, activation, Ypow, Y, (double *)(ws + wbuf), (double *)(ws + gb),
                       d.B * N, d.F, d.T, d.K, d, st);
}

// Basis update; with loss_out (B zeroed doubles) the tuned kernel also leaves the data term of the loss
// of the state at entry there.  Returns 1 (not an error code of the ABI) when loss_out was requested
// but the shape takes the generic kernels, which have no such by-product: the caller then makes the
// dedicated loss pass.
static int update_basis_impl(const void *X, const void *W, double *basis, const double *activation,
                             int B, int N, int F, int T, int K, double domain, int source_model,
                             double model_param, int floor_kind, double floor_eps, void *workspace,
                             size_t workspace_bytes, double *loss_out, bool *loss_done,
                             void *stream, bool x_is_power = false, long long loss_stride = 0) {
  // loss_stride > 0: loss_out is a raw slot array (see ilrma_fast_basis), tuned path only
  // x_is_power (grouped path of a wide mixture only, W == NULL): X holds |y|^2 (B, N, F, T) f64
  SSSPY_REQUIRE(X && basis && activation && B > 0 && F > 0 && T > 0, "update_basis: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "update_basis: n_basis must be in [1, 65536]");
  SSSPY_REQUIRE(domain > 0.0 && domain <= 2.0, "update_basis: domain must be in (0, 2]");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  const IlrmaWs w = ilrma_ws(B, N, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "update_basis: workspace too small");
  char *ws = (char *)workspace;
  hipStream_t st = as_stream(stream);
  if (loss_done) *loss_done = false;
  SSSPY_REQUIRE(!x_is_power || (!W && (grouped_path(B, N, F, T, K, domain, source_model) ||
                                       wide_basis_shape(N, K))),
                "update_basis: power input off the grouped / wide-basis path");
  // above 16 bases the update cannot be in place (several items per bin group read the old basis)
  double *out = K > 16 ? (double *)(ws + w.btmp) : basis;
  bool in_place = false;
  auto run = [&]() -> int {
    SourceRun runs[3];
    if (const int nruns = source_runs(B, N, F, T, K, domain, source_model, runs)) {
      // what the tuned kernels read: the separated spectrogram handed in (ISS / IPA state), or the
      // power |W x|^2 formed here (half the bytes of y)
      const char *Y = (const char *)X;
      bool power = x_is_power;
      if (W) {
        int r = separate_power(X, W, (double *)(ws + w.ybuf), B, N, F, T, st);
        if (r) return r;
        Y = ws + w.ybuf;
        power = true;
      }
      const size_t elem = power ? sizeof(double) : sizeof(c128);
      for (int i = 0; i < nruns; ++i) {
        const SourceRun &sr = runs[i];
        auto one = [&]() -> int {
          ILRMA_FAST_DISPATCH(sr.G, ilrma_fast_basis, Y + (size_t)sr.first * F * T * elem, nullptr,
                              basis + sr.first * F * K, out + sr.first * F * K,
                              activation + sr.first * K * T, sr.count, F, T, K, floor_kind,
                              floor_eps, (double *)(ws + w.bpart),
                              fast_model_id(domain, source_model), fast_model_param(domain, source_model, model_param),
                              is_me(source_model), nullptr, nullptr, power ? 1 : 0, st, 0ll);
        };
        const int r = one();
        if (r) return r;
      }
      return SSSPY_OK;
    }
    if (fast_path(N, F, T, K, domain, source_model) && !(K > 32 && wide_basis_shape(N, K))) {
      if (loss_done) *loss_done = loss_out != nullptr && K <= 16;
      ILRMA_FAST_DISPATCH(N, ilrma_fast_basis, X, W, basis, out, activation, B, F, T, K, floor_kind,
                          floor_eps, (double *)(ws + w.bpart), fast_model_id(domain, source_model),
                          fast_model_param(domain, source_model, model_param), is_me(source_model),
                          K <= 16 ? loss_out : nullptr, ws + w.lslots, 0, st, loss_stride);
    }
    const IlrmaDims d =
        make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
    if (wide_basis_shape(N, K)) {
      // n_basis above 64 (or more than 8 sources off the grouped path): dense products, in place
      int r = wb_weights(X, W, x_is_power, basis, activation, N, d, ws, w.ybuf, w.wbuf, w.gb, st);
      if (r) return r;
      in_place = true;
      return wb_update_basis((const double *)(ws + w.wbuf), (const double *)(ws + w.gb), basis,
                             activation, (double *)(ws + w.gnd), B * N, F, T, K, d, st);
    }
    ILRMA_DISPATCH(N, ilrma_basis, X, W, basis, out, activation, d, st);
  };
  rc = run();
  if (rc) return rc;
  if (out != basis && !in_place) {
    hipError_t e = hipMemcpyAsync(basis, out, (size_t)B * N * F * K * sizeof(double),
                                  hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
  }
  return SSSPY_OK;
}

int ssspy_ilrma_update_basis(const void *X, const void *W, double *basis, const double *activation,
                             int B, int N, int F, int T, int K, double domain, int source_model,
                             double model_param, int floor_kind, double floor_eps, void *workspace,
                             size_t workspace_bytes, void *stream) {
  return update_basis_impl(X, W, basis, activation, B, N, F, T, K, domain, source_model, model_param,
                           floor_kind, floor_eps, workspace, workspace_bytes, nullptr, nullptr,
                           stream);
}

static int update_activation_impl(const void *X, const void *W, const double *basis,
                                  double *activation, int B, int N, int F, int T, int K,
                                  double domain, int source_model, double model_param,
                                  int floor_kind, double floor_eps, void *workspace,
                                  size_t workspace_bytes, void *stream, bool x_is_power) {
  SSSPY_REQUIRE(X && basis && activation && B > 0 && F > 0 && T > 0,
                "update_activation: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "update_activation: n_basis must be in [1, 65536]");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  const IlrmaWs w = ilrma_ws(B, N, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "update_activation: workspace too small");
  double *part = (double *)((char *)workspace + w.act_part);
  const int chunks = act_chunks(B, N, F, T, K);
  hipStream_t st = as_stream(stream);
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
  SourceRun runs[3];
  const int nruns = source_runs(B, N, F, T, K, domain, source_model, runs);
  SSSPY_REQUIRE(!x_is_power || (!W && (nruns || wide_basis_shape(N, K))),
                "update_activation: power input off the grouped / wide-basis path");
  if (!nruns && small_path(B, N, F, T, K, domain, source_model)) {
    // a handful of mixtures: the latency kernel and its own fold (in place)
    ILRMA_FAST_DISPATCH(N, ilrma_small_activation, X, W, basis, activation, B, F, T, K, floor_kind,
                        floor_eps, part, fast_model_id(domain, source_model), fast_model_param(domain, source_model, model_param),
                        is_me(source_model), st);
  }
  if (!nruns && wide_basis_shape(N, K) &&
      !(fast_path(N, F, T, K, domain, source_model) && K <= 32)) {
    // n_basis above 64 (or more than 8 sources off the grouped path): dense products (wide_basis.hip)
    char *ws = (char *)workspace;
    rc = wb_weights(X, W, x_is_power, basis, activation, N, d, ws, w.ybuf, w.wbuf, w.gb, st);
    if (rc) return rc;
    return wb_update_activation((const double *)(ws + w.wbuf), (const double *)(ws + w.gb), basis,
                                activation, (double *)(ws + w.gnd), B * N, F, T, K, d, st);
  }
  // (the partial sums of a run keep the (group, chunk, source) layout at the run's offset: every
  // source owns `chunks` slabs of 2 K T doubles wherever its group starts)
  const size_t part_per_source = (size_t)chunks * 2 * K * T;
  auto run = [&]() -> int {
    if (nruns) {
      const char *Y = (const char *)X;
      bool power = x_is_power;
      if (W) {
        int r = separate_power(X, W, (double *)((char *)workspace + w.ybuf), B, N, F, T, st);
        if (r) return r;
        Y = (const char *)workspace + w.ybuf;
        power = true;
      }
      const size_t elem = power ? sizeof(double) : sizeof(c128);
      for (int i = 0; i < nruns; ++i) {
        const SourceRun &sr = runs[i];
        auto one = [&]() -> int {
          ILRMA_FAST_DISPATCH(sr.G, ilrma_fast_activation, Y + (size_t)sr.first * F * T * elem,
                              nullptr, basis + sr.first * F * K, activation + sr.first * K * T,
                              part + sr.first * part_per_source, chunks, sr.count, F, T, K,
                              fast_model_id(domain, source_model), fast_model_param(domain, source_model, model_param), power ? 1 : 0, st);
        };
        const int r = one();
        if (r) return r;
      }
      return SSSPY_OK;
    }
    if (fast_path(N, F, T, K, domain, source_model)) {
      ILRMA_FAST_DISPATCH(N, ilrma_fast_activation, X, W, basis, activation, part, chunks, B, F, T,
                          K, fast_model_id(domain, source_model), fast_model_param(domain, source_model, model_param), 0, st);
    }
    ILRMA_DISPATCH(N, ilrma_activation, X, W, basis, activation, part, chunks, d, st);
  };
  rc = run();
  if (rc) return rc;
  // fold the chunks in the layout the kernel wrote: (mixture, chunk, source) of the regrouped batch
  // when the wide-mixture path ran
  if (!nruns) {
    dim3 g2((unsigned)(((long long)K * T + 255) / 256), N, B);
    hipLaunchKernelGGL(k_ilrma_activation_finalize, g2, dim3(256), 0, st, activation,
                       (const double *)part, N, K, T, chunks, d);
    return check_launch("k_ilrma_activation_finalize");
  }
  for (int i = 0; i < nruns; ++i) {
    const SourceRun &sr = runs[i];
    dim3 g2((unsigned)(((long long)K * T + 255) / 256), sr.G, sr.count);
    hipLaunchKernelGGL(k_ilrma_activation_finalize, g2, dim3(256), 0, st,
                       activation + sr.first * K * T,
                       (const double *)(part + sr.first * part_per_source), sr.G, K, T, chunks, d);
  }
  return check_launch("k_ilrma_activation_finalize");
}

int ssspy_ilrma_update_activation(const void *X, const void *W, const double *basis,
                                  double *activation, int B, int N, int F, int T, int K,
                                  double domain, int source_model, double model_param,
                                  int floor_kind, double floor_eps, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  return update_activation_impl(X, W, basis, activation, B, N, F, T, K, domain, source_model,
                                model_param, floor_kind, floor_eps, workspace, workspace_bytes,
                                stream, false);
}

// U[b,i,n] for every model; `upart` is the fast path's scratch for split blocks.  Wide mixtures
// (5..8 sources) go through the weights varphi = 1 / R~ (`wbuf`, (B, N, F, T)) and the matrix-core
// covariance of wide_cov.hip; the heavy-tailed models need the separated spectrogram for that
// (`Ysep`, or X itself when W is NULL) and keep the generic kernel without it.
static int wcov_into(const void *X, const void *W, const double *basis, const double *activation,
                     void *U, int N, const IlrmaDims &d, void *upart, double *wbuf,
                     const void *Ysep, bool ysep_is_power, hipStream_t st) {
  if (rt_sources_ok(N)) {
    // more than 8 sources: varphi = 1 / R^(2/p) by the ISS weight kernel (run-time N), then the
    // run-time covariance of wide_n.hip.  Gauss model, or any model with y at hand
    const void *Y = Ysep ? Ysep : (W ? nullptr : X);
    const bool ypow = Ysep && ysep_is_power;
    if (!wbuf || !(d.model == SSSPY_SOURCE_GAUSS || Y))
      return fail(SSSPY_ERR_UNSUPPORTED,
                  "ILRMA above 8 sources: covariance weights need the Gauss model or the separated "
                  "spectrogram");
    const int chunks = iss_weight_chunks(d.B, N, d.F, d.T);
    dim3 grid(((d.F + 63) / 64) * chunks, N, d.B), block(256);
    if (d.model == SSSPY_SOURCE_GAUSS && d.p == 2.0)
      hipLaunchKernelGGL(k_ilrma_iss_weight<true>, grid, block, 0, st, (const c128 *)nullptr,
                         (const double *)nullptr, basis, activation, wbuf, N, d, chunks,
                         (double *)nullptr);
    else
      hipLaunchKernelGGL(k_ilrma_iss_weight<false>, grid, block, 0, st,
                         ypow ? nullptr : (const c128 *)Y, ypow ? (const double *)Y : nullptr, basis,
                         activation, wbuf, N, d, chunks, (double *)nullptr);
    int rc = check_launch("k_ilrma_iss_weight");
    if (rc) return rc;
    return rt_covariance(X, X, wbuf, SSSPY_WEIGHT_BIN_FRAME, U, d.B, N, N, d.F, d.T, st);
  }
  if (wbuf && d.K > 32 && wide_basis_shape(N, d.K) &&
      (d.model == SSSPY_SOURCE_GAUSS || Ysep || !W)) {
    // n_basis above 64: the weight kernel walks any n_basis on the matrix cores; the covariance is
    // then the shared weighted-covariance operator (no T V inside it)
    const void *Y = Ysep ? Ysep : (W ? nullptr : X);
    const bool ypow = Ysep && ysep_is_power;
    int rc = wb_tv_weights(2, basis, activation, ypow ? (const double *)Y : nullptr,
                           ypow ? nullptr : Y, wbuf, nullptr, d.B * N, d.F, d.T, d.K, d, st);
    if (rc) return rc;
    return ssspy_weighted_covariance(X, wbuf, SSSPY_WEIGHT_BIN_FRAME, U, d.B, N, N, d.F, d.T, st);
  }
  if (N > 4 && wbuf && wide_weighted_cov_ok(N, N, d.F, d.T, SSSPY_WEIGHT_BIN_FRAME)) {
    const void *Y = Ysep ? Ysep : (W ? nullptr : X);
    const bool ypow = Ysep && ysep_is_power;
    if (d.model == SSSPY_SOURCE_GAUSS || Y) {
      const int chunks = iss_weight_chunks(d.B, N, d.F, d.T);
      dim3 grid(((d.F + 63) / 64) * chunks, N, d.B), block(256);
      if (d.model == SSSPY_SOURCE_GAUSS && d.p == 2.0)
        hipLaunchKernelGGL(k_ilrma_iss_weight<true>, grid, block, 0, st, (const c128 *)nullptr,
                           (const double *)nullptr, basis, activation, wbuf, N, d, chunks,
                           (double *)nullptr);
      else
        hipLaunchKernelGGL(k_ilrma_iss_weight<false>, grid, block, 0, st,
                           ypow ? nullptr : (const c128 *)Y, ypow ? (const double *)Y : nullptr,
                           basis, activation, wbuf, N, d, chunks, (double *)nullptr);
      return wide_weighted_cov(X, wbuf, SSSPY_WEIGHT_BIN_FRAME, U, d.B, N, N, d.F, d.T, st);
    }
  }
  if (fast_path(N, d.F, d.T, d.K, d.p, d.model) && (d.model == SSSPY_SOURCE_GAUSS || W)) {
    ILRMA_FAST_DISPATCH(N, ilrma_fast_wcov, X, W, basis, activation, U, d.B, d.F, d.T, d.K, upart,
                        fast_model_id(d.p, d.model), fast_model_param(d.p, d.model, d.mparam), d.floor_kind, d.floor_eps, st,
                        nullptr, nullptr);
  }
  ILRMA_DISPATCH(N, ilrma_wcov, X, W, basis, activation, U, d, st);
}

int ssspy_ilrma_weighted_covariance(const void *X, const void *W, const double *basis,
                                    const double *activation, void *U, int B, int N, int F, int T,
                                    int K, double domain, int source_model, double model_param,
                                    int floor_kind, double floor_eps, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && basis && activation && U && B > 0 && F > 0 && T > 0,
                "ilrma_weighted_covariance: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "ilrma_weighted_covariance: bad n_basis");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  const IlrmaWs w = ilrma_ws(B, N, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total,
                "ilrma_weighted_covariance: workspace too small");
  hipStream_t st = as_stream(stream);
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
  // more than 8 sources with a heavy-tailed model: the run-time-N weights need |W x|^2 (the fused
  // IP1 iteration has it from its NMF passes; a caller of this entry alone -- IP2 -- has not)
  const void *Ysep = nullptr;
  if (rt_sources_ok(N) && source_model != SSSPY_SOURCE_GAUSS && W) {
    rc = separate_power(X, W, (double *)((char *)workspace + w.ybuf), B, N, F, T, st);
    if (rc) return rc;
    Ysep = (const char *)workspace + w.ybuf;
  }
  return wcov_into(X, W, basis, activation, U, N, d, (char *)workspace + w.upart,
                   (N > 4 || wide_basis_shape(N, K)) ? (double *)((char *)workspace + w.wbuf) : nullptr,
                   Ysep, Ysep != nullptr, st);
}

static int launch_norm_scale(void *W, double *basis, const double *qbuf, int B, int N, int F, int K,
                             double domain, int floor_kind, double floor_eps, hipStream_t st) {
  hipLaunchKernelGGL(k_norm_scale, dim3((F + 63) / 64, B), dim3(256), 0, st, (c128 *)W, basis,
                     qbuf, N, F, K, domain, floor_kind, floor_eps, (double *)nullptr);
  return check_launch("k_norm_scale");
}

int ssspy_ilrma_normalize_filter(void *W, const void *C, double *basis, int B, int N, int F, int K,
                                 double domain, int floor_kind, double floor_eps, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(W && C && basis && B > 0 && N >= 1 && N <= SSSPY_RT_MAX_SOURCES,
                "normalize_filter: bad argument");
  hipStream_t st = as_stream(stream);
  SSSPY_REQUIRE(workspace && workspace_bytes >= qbuf_bytes(B, N, F),
                "normalize_filter: workspace too small");
  double *qbuf = (double *)workspace;  // any B*F*N doubles of scratch
  int rc = row_power(W, C, qbuf, B, F, N, st);
  if (rc) return rc;
  return launch_norm_scale(W, basis, qbuf, B, N, F, K, domain, floor_kind, floor_eps, st);
}

static int normalize_output_impl(void *Y, double *basis, const double *frame_power, int B, int N,
                                 int F, int T, int K, double domain, int floor_kind,
                                 double floor_eps, void *workspace, size_t workspace_bytes,
                                 double *logdet, void *stream) {
  SSSPY_REQUIRE(Y && basis && B > 0 && N >= 1, "normalize_output: bad argument");
  // frame powers given: one slot per (mixture, source); else one per block of 16 bin rows
  const int slots = frame_power ? 1 : (F + 15) / 16;
  SSSPY_REQUIRE(workspace && workspace_bytes >= (size_t)B * N * slots * sizeof(double),
                "normalize_output: workspace too small");
  hipStream_t st = as_stream(stream);
  double *acc = (double *)workspace;
  dim3 grid(F, N, B), block(256);
  if (frame_power) {
    hipLaunchKernelGGL(k_power_from_frames, dim3(N, B), block, 0, st, frame_power, acc, N, T);
  } else {
    hipLaunchKernelGGL(k_output_power, dim3(slots, N, B), block, 0, st, (const c128 *)Y, acc, N, F,
                       T);
  }
  hipLaunchKernelGGL(k_ilrma_normalize_output, grid, block, 0, st, (c128 *)Y, basis, acc, slots, N,
                     F, T, K, domain, floor_kind, floor_eps, (double *)nullptr, logdet);
  return check_launch("k_ilrma_normalize_output");
}

int ssspy_ilrma_normalize_output(void *Y, double *basis, const double *frame_power, int B, int N,
                                 int F, int T, int K, double domain, int floor_kind,
                                 double floor_eps, void *workspace, size_t workspace_bytes,
                                 void *stream) {
  return normalize_output_impl(Y, basis, frame_power, B, N, F, T, K, domain, floor_kind, floor_eps,
                               workspace, workspace_bytes, nullptr, stream);
}

int ssspy_ilrma_normalize_output_tracked(void *Y, double *basis, const double *frame_power, int B,
                                         int N, int F, int T, int K, double domain, int floor_kind,
                                         double floor_eps, void *workspace, size_t workspace_bytes,
                                         double *logdet, void *stream) {
  SSSPY_REQUIRE(logdet, "normalize_output_tracked: bad argument");
  return normalize_output_impl(Y, basis, frame_power, B, N, F, T, K, domain, floor_kind, floor_eps,
                               workspace, workspace_bytes, logdet, stream);
}

int ssspy_ilrma_iss_weight_power(const double *Ypow, const double *basis, const double *activation,
                                 double *varphi, int B, int N, int F, int T, int K, double domain,
                                 int source_model, double model_param, int floor_kind,
                                 double floor_eps, void *stream) {
  SSSPY_REQUIRE(Ypow && basis && activation && varphi && B > 0, "iss_weight_power: bad argument");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
  const int chunks = iss_weight_chunks(B, N, F, T);
  dim3 grid(((F + 63) / 64) * chunks, N, B), block(256);
  hipLaunchKernelGGL(k_ilrma_iss_weight<false>, grid, block, 0, as_stream(stream),
                     (const c128 *)nullptr, Ypow, basis, activation, varphi, N, d, chunks,
                     (double *)nullptr);
  return check_launch("k_ilrma_iss_weight");
}

int ssspy_ilrma_iss_weight(const void *Y, const double *basis, const double *activation,
                           double *varphi, int B, int N, int F, int T, int K, double domain,
                           int source_model, double model_param, int floor_kind, double floor_eps,
                           void *stream) {
  SSSPY_REQUIRE(basis && activation && varphi && B > 0, "iss_weight: bad argument");
  SSSPY_REQUIRE(source_model == SSSPY_SOURCE_GAUSS || Y, "iss_weight: this model needs Y");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
  const int chunks = iss_weight_chunks(B, N, F, T);
  dim3 grid(((F + 63) / 64) * chunks, N, B), block(256);
  if (source_model == SSSPY_SOURCE_GAUSS && domain == 2.0)
    hipLaunchKernelGGL(k_ilrma_iss_weight<true>, grid, block, 0, as_stream(stream),
                       (const c128 *)nullptr, (const double *)nullptr, basis, activation, varphi, N,
                       d, chunks, (double *)nullptr);
  else
    hipLaunchKernelGGL(k_ilrma_iss_weight<false>, grid, block, 0, as_stream(stream), (const c128 *)Y,
                       (const double *)nullptr, basis, activation, varphi, N, d, chunks,
                       (double *)nullptr);
  return check_launch("k_ilrma_iss_weight");
}

size_t ssspy_ilrma_loss_workspace_bytes(int B, int N, int F, int T, int K, int with_filter) {
  if (B <= 0 || N <= 0 || F <= 0 || T <= 0 || K <= 0) return 0;
  return loss_slots_bytes(B, N, F, T, K, with_filter != 0);
}

int ssspy_ilrma_loss_data(const void *X, const void *W, const double *basis,
                          const double *activation, double *out, int B, int N, int F, int T, int K,
                          double domain, int source_model, double model_param, void *workspace,
                          size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && basis && activation && out && B > 0, "ilrma_loss_data: bad argument");
  SSSPY_REQUIRE(K >= 1 && K <= SSSPY_MAX_BASIS, "ilrma_loss_data: bad n_basis");
  SSSPY_REQUIRE(N >= 2 && N <= SSSPY_RT_MAX_SOURCES, "ilrma_loss_data: n_sources must be in [2, 16]");
  SSSPY_REQUIRE(workspace && workspace_bytes >= loss_slots_bytes(B, N, F, T, K, W != nullptr),
                "ilrma_loss_data: workspace too small (ssspy_ilrma_loss_workspace_bytes)");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  if (K <= 16 && fast_path(N, F, T, K, domain, source_model)) {
    ILRMA_FAST_DISPATCH(N, ilrma_fast_loss, X, W, basis, activation, out, workspace, B, F, T, K,
                        fast_model_id(domain, source_model), fast_model_param(domain, source_model, model_param), st);
  }
  if (K > 16 || rt_sources_ok(N)) {
    // n_basis above 16 (or more than 8 sources): T V on the matrix cores, the terms summed in the
    // product's epilogue (wide_basis.hip) -- the per-N loss kernels took 1.7 / 2.8 / 5.0 ms at
    // n_basis 32 / 64 / 128 (32 mixtures), as much as the iteration they follow
    const IlrmaDims dl = make_dims(B, F, T, K, domain, source_model, model_param, SSSPY_FLOOR_NONE, 0.0);
    char *wsb = (char *)workspace;
    const size_t slots = align256(wb_loss_ws_bytes(B, N, F, T));
    const double *ypow = nullptr;
    const void *y = X;
    if (W) {
      rc = separate_power(X, W, (double *)(wsb + slots), B, N, F, T, st);
      if (rc) return rc;
      ypow = (const double *)(wsb + slots);
      y = nullptr;
    }
    return wb_loss_data(basis, activation, ypow, y, out, workspace, B, N, F, T, K, dl, st);
  }
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, SSSPY_FLOOR_NONE, 0.0);
  ILRMA_DISPATCH(N, ilrma_loss, X, W, basis, activation, out, workspace, d, st);
}

static int ip1_update_impl(const void *X, const void *C, void *W, double *basis, double *activation,
                           void *U, int B, int N, int F, int T, int K, double domain,
                           int source_model, double model_param, int normalize, int floor_kind,
                           double floor_eps, void *workspace, size_t workspace_bytes, int *info,
                           double *loss_data, double *logdet, void *stream,
                           long long loss_stride = 0) {
  SSSPY_REQUIRE(X && W && basis && activation && U, "ilrma_ip1_update: bad argument");
  SSSPY_REQUIRE(!normalize || C, "ilrma_ip1_update: normalisation needs C");
  SSSPY_REQUIRE((loss_data == nullptr) == (logdet == nullptr),
                "ilrma_ip1_update: loss_data and logdet go together");
  const IlrmaWs w = ilrma_ws(B, N, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "ilrma_ip1_update: workspace too small");
  char *ws = (char *)workspace;
  hipStream_t st = as_stream(stream);
  int rc;
  if (loss_data) {
    // (the Student-t data term is not linear in the pass's accumulators; no by-product there)
    if (!ssspy_ilrma_deferred_loss_supported(N, F, T, K, domain, source_model))
      return fail(SSSPY_ERR_UNSUPPORTED,
                  "ilrma_ip1_update_deferred_loss: this shape takes the generic kernels, which have "
                  "no loss by-product (use ssspy_ilrma_loss_data + ssspy_ilrma_ip1_update)");
    // loss of the state at entry: log-determinants now (IP1 rewrites W below) -- or, with slots on
    // the latency path, as one share per 16-bin tile from the IP1 kernel itself, which reads the
    // same filters before it rewrites them -- data term as a by-product of the basis pass
    if (!(loss_stride > 0 && small_path(B, N, F, T, K, domain, source_model))) {
      rc = ssspy_sum_logdet(W, logdet, B, F, N, stream);
      if (rc) return rc;
    }
    // (loss_data is stored, not accumulated: the basis pass folds its per-wave shares into it)
  }
  bool loss_done = false;
  // wide mixture on the grouped path: y = W x once for both NMF passes (they then see the ISS-style
  // state: the spectrogram itself, no filter)
  const void *Xs = X, *Ws = W;
  bool xs_is_power = false;
  if (grouped_path(B, N, F, T, K, domain, source_model)) {
    // (the passes that follow need |y|^2 only: the NMF passes, and the weights of a heavy-tailed
    // covariance pass)
    rc = separate_power(X, W, (double *)(ws + w.ybuf), B, N, F, T, st);
    if (rc) return rc;
    Xs = ws + w.ybuf;
    Ws = nullptr;
    xs_is_power = true;
  } else if (wide_basis_shape(N, K) && !(K <= 32 && fast_path(N, F, T, K, domain, source_model))) {
    // dense-product path (wide_basis.hip): |W x|^2 once for both source updates and the weights
    rc = separate_power(X, W, (double *)(ws + w.ybuf), B, N, F, T, st);
    if (rc) return rc;
    Xs = ws + w.ybuf;
    Ws = nullptr;
    xs_is_power = true;
  }
  rc = update_basis_impl(Xs, Ws, basis, activation, B, N, F, T, K, domain, source_model,
                         model_param, floor_kind, floor_eps, workspace, workspace_bytes, loss_data,
                         &loss_done, stream, xs_is_power, loss_stride);
  if (rc) return rc;
  // (unreachable: ssspy_ilrma_deferred_loss_supported above admits exactly the shapes whose basis
  // pass leaves the data term; kept as an internal error because the basis is already rewritten)
  if (loss_data && !loss_done)
    return fail(SSSPY_ERR_INTERNAL, "ilrma_ip1_update: basis pass left no loss by-product");
  rc = update_activation_impl(Xs, Ws, basis, activation, B, N, F, T, K, domain, source_model,
                              model_param, floor_kind, floor_eps, workspace, workspace_bytes, stream,
                              xs_is_power);
  if (rc) return rc;
  double *qbuf = (double *)(ws + w.qbuf);
  if (small_path(B, N, F, T, K, domain, source_model)) {
    // a handful of mixtures: the covariance pass leaves its split items' records, and one kernel
    // folds them, runs IP1 and forms the output power; U is materialised only if no item was split
    int split = 0, rbins = 0;
    auto cov = [&]() -> int {
      ILRMA_FAST_DISPATCH(N, ilrma_fast_wcov, X, W, basis, activation, U, B, F, T, K, ws + w.upart,
                          fast_model_id(domain, source_model), fast_model_param(domain, source_model, model_param), floor_kind, floor_eps,
                          st, &split, &rbins);
    };
    rc = cov();
    if (rc) return rc;
    auto ip1 = [&]() -> int {
      if (loss_data && loss_stride > 0) {
        ILRMA_FAST_DISPATCH(N, ilrma_small_ip1_logdet,
                            split ? (const void *)(ws + w.upart) : (const void *)U, split, rbins, 0ll,
                            normalize ? C : nullptr, W, B, F, floor_kind, floor_eps, qbuf, info,
                            logdet, loss_stride, st);
      }
      ILRMA_FAST_DISPATCH(N, ilrma_small_ip1, split ? (const void *)(ws + w.upart) : (const void *)U,
                          split, rbins, 0ll, normalize ? C : nullptr, W, B, F, floor_kind, floor_eps,
                          qbuf, info, st);
    };
    rc = ip1();
    if (rc || !normalize) return rc;
    ILRMA_FAST_DISPATCH(N, ilrma_small_norm, W, basis, qbuf, B, F, K, domain, floor_kind, floor_eps,
                        st);
  }
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
  rc = wcov_into(X, W, basis, activation, U, N, d, ws + w.upart,
                 (N > 4 || wide_basis_shape(N, K)) ? (double *)(ws + w.wbuf) : nullptr,
                 Ws ? nullptr : Xs, xs_is_power, st);
  if (rc) return rc;
  rc = ip1_with_power(W, U, normalize ? C : nullptr, normalize ? qbuf : nullptr, B, F, N,
                      floor_kind, floor_eps, info, st);
  if (rc || !normalize) return rc;
  return launch_norm_scale(W, basis, qbuf, B, N, F, K, domain, floor_kind, floor_eps, st);
}

int ssspy_ilrma_ip1_update(const void *X, const void *C, void *W, double *basis, double *activation,
                           void *U, int B, int N, int F, int T, int K, double domain,
                           int source_model, double model_param, int normalize, int floor_kind,
                           double floor_eps, void *workspace, size_t workspace_bytes, int *info,
                           void *stream) {
  return ip1_update_impl(X, C, W, basis, activation, U, B, N, F, T, K, domain, source_model,
                         model_param, normalize, floor_kind, floor_eps, workspace, workspace_bytes,
                         info, nullptr, nullptr, stream);
}

int ssspy_ilrma_deferred_loss_supported(int N, int F, int T, int K, double domain,
                                        int source_model) {
  return K <= 16 && fast_path(N, F, T, K, domain, source_model) &&
         fast_model_id(domain, source_model) != 1;
}

int ssspy_ilrma_ip1_update_deferred_loss(const void *X, const void *C, void *W, double *basis,
                                         double *activation, void *U, int B, int N, int F, int T,
                                         int K, double domain, int source_model, double model_param,
                                         int normalize, int floor_kind, double floor_eps,
                                         void *workspace, size_t workspace_bytes, int *info,
                                         double *loss_data, double *logdet, void *stream) {
  SSSPY_REQUIRE(loss_data && logdet, "ilrma_ip1_update_deferred_loss: bad argument");
  return ip1_update_impl(X, C, W, basis, activation, U, B, N, F, T, K, domain, source_model,
                         model_param, normalize, floor_kind, floor_eps, workspace, workspace_bytes,
                         info, loss_data, logdet, stream);
}

// ---- the same with the loss by-product left as raw slots (round 5): a run of n_iter iterations
// zeroes one array and folds it once instead of a memset, a counter memset and a fold launch per
// iteration (3 of the 12 launches of a one-mixture iteration, 17 of its 124 us)
int ssspy_ilrma_deferred_loss_slots(int B, int N, int F, int T, int K, double domain,
                                    int source_model) {
  if (!ssspy_ilrma_deferred_loss_supported(N, F, T, K, domain, source_model)) return 0;
  switch (N) {
    case 2: return ilrma_fast_basis_loss_slots_n2(B, F, T);
    case 3: return ilrma_fast_basis_loss_slots_n3(B, F, T);
    case 4: return ilrma_fast_basis_loss_slots_n4(B, F, T);
    default: return 0;
  }
}

int ssspy_ilrma_deferred_logdet_slots(int B, int N, int F, int T, int K, double domain,
                                      int source_model) {
  if (!ssspy_ilrma_deferred_loss_supported(N, F, T, K, domain, source_model)) return 0;
  return small_path(B, N, F, T, K, domain, source_model) ? (F + 15) / 16 : 1;
}

int ssspy_ilrma_ip1_update_loss_slots(const void *X, const void *C, void *W, double *basis,
                                      double *activation, void *U, int B, int N, int F, int T,
                                      int K, double domain, int source_model, double model_param,
                                      int normalize, int floor_kind, double floor_eps,
                                      void *workspace, size_t workspace_bytes, int *info,
                                      double *slots, long long slot_stride, double *logdet,
                                      void *stream) {
  SSSPY_REQUIRE(slots && logdet && slot_stride >= B && slot_stride < (1ll << 31),
                "ilrma_ip1_update_loss_slots: bad argument");
  return ip1_update_impl(X, C, W, basis, activation, U, B, N, F, T, K, domain, source_model,
                         model_param, normalize, floor_kind, floor_eps, workspace, workspace_bytes,
                         info, slots, logdet, stream, slot_stride);
}

size_t ssspy_fold_scalar_slots_workspace_bytes(long long total, int nslots) {
  if (total <= 0 || nslots <= 0) return 0;
  return align256(fold_scratch_bytes(total, nslots)) + 256;
}

int ssspy_fold_scalar_slots(const double *slots, long long total, int nslots, double *out,
                            void *workspace, size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(slots && out && total > 0 && nslots > 0, "fold_scalar_slots: bad argument");
  SSSPY_REQUIRE(workspace && workspace_bytes >= ssspy_fold_scalar_slots_workspace_bytes(total, nslots),
                "fold_scalar_slots: workspace too small");
  return launch_fold_slabs(slots, workspace, out, total, nslots, as_stream(stream), 0);
}

int ssspy_ilrma_partition_expand(const double *basis, const double *activation,
                                 const double *latent, double *Teff, double *Vrep, int B, int N,
                                 int F, int T, int K, void *stream) {
  SSSPY_REQUIRE(basis && activation && latent && Teff && Vrep && B > 0 && N >= 1 && F > 0 &&
                    T > 0 && K >= 1,
                "partition_expand: bad argument");
  const long long per = (long long)K * (F > T ? F : T);
  hipLaunchKernelGGL(k_partition_expand, dim3((unsigned)((per + 255) / 256), N, B), dim3(256), 0,
                     as_stream(stream), basis, activation, latent, Teff, Vrep, N, F, T, K);
  return check_launch("k_partition_expand");
}

int ssspy_ilrma_partition_update(const void *X, const void *W, double *basis, double *activation,
                                 double *latent, double *Teff, double *Vrep, int B, int N, int F,
                                 int T, int K, double domain, int source_model, double model_param,
                                 int steps, int floor_kind, double floor_eps, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(X && basis && activation && latent && Teff && Vrep && B > 0 && F > 0 && T > 0,
                "partition_update: bad argument");
  SSSPY_REQUIRE(N >= 1, "partition_update: bad n_sources");
  if (N > SSSPY_MAX_SOURCES)
    return fail(SSSPY_ERR_UNSUPPORTED, "ILRMA: partitioning takes up to 8 sources");
  SSSPY_REQUIRE(K >= 1, "partition_update: bad n_basis");
  if (K > SSSPY_MAX_PARTITION_BASIS)
    return fail(SSSPY_ERR_UNSUPPORTED, "ILRMA: partitioning takes n_basis up to 1024");
  SSSPY_REQUIRE(domain > 0.0 && domain <= 2.0, "partition_update: domain must be in (0, 2]");
  int rc = check_model(source_model, model_param, domain);
  if (rc) return rc;
  const IlrmaWs w = ilrma_ws(B, N, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "partition_update: workspace too small");
  char *ws = (char *)workspace;
  double *raw = (double *)(ws + w.praw);
  double *part = (double *)(ws + w.act_part);
  hipStream_t st = as_stream(stream);
  const IlrmaDims d = make_dims(B, F, T, K, domain, source_model, model_param, floor_kind, floor_eps);
  IlrmaDims draw = d;
  draw.raw = 1;
  auto basis_sums = [&]() -> int {
    int r = ssspy_ilrma_partition_expand(basis, activation, latent, Teff, Vrep, B, N, F, T, K, stream);
    if (r) return r;
    ILRMA_DISPATCH(N, ilrma_basis, X, W, Teff, raw, Vrep, draw, st);
  };
  if (steps & SSSPY_PARTITION_LATENT) {
    rc = basis_sums();
    if (rc) return rc;
    hipLaunchKernelGGL(k_partition_latent, dim3(B), dim3(256), 0, st, (const double *)raw,
                       (const double *)basis, latent, N, F, K, d);
    rc = check_launch("k_partition_latent");
    if (rc) return rc;
  }
  if (steps & SSSPY_PARTITION_BASIS) {
    rc = basis_sums();
    if (rc) return rc;
    hipLaunchKernelGGL(k_partition_basis, dim3((unsigned)(((long long)F * K + 255) / 256), B),
                       dim3(256), 0, st, (const double *)raw, (const double *)latent, basis, N, F,
                       K, d);
    rc = check_launch("k_partition_basis");
    if (rc) return rc;
  }
  if (steps & SSSPY_PARTITION_ACTIVATION) {
    rc = ssspy_ilrma_partition_expand(basis, activation, latent, Teff, Vrep, B, N, F, T, K, stream);
    if (rc) return rc;
    const int chunks = act_chunks(B, N, F, T, K);
    auto run = [&]() -> int {
      ILRMA_DISPATCH(N, ilrma_activation, X, W, Teff, Vrep, part, chunks, d, st);
    };
    rc = run();
    if (rc) return rc;
    hipLaunchKernelGGL(k_partition_activation, dim3((unsigned)(((long long)K * T + 255) / 256), B),
                       dim3(256), 0, st, (const double *)part, activation, N, K, T, chunks, d);
    rc = check_launch("k_partition_activation");
    if (rc) return rc;
  }
  return ssspy_ilrma_partition_expand(basis, activation, latent, Teff, Vrep, B, N, F, T, K, stream);
}

int ssspy_ilrma_partition_normalize(void *W, const void *C, void *Y, double *basis, double *latent,
                                    int B, int N, int F, int T, int K, double domain,
                                    int floor_kind, double floor_eps, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  SSSPY_REQUIRE(basis && latent && B > 0 && N >= 1, "partition_normalize: bad argument");
  if (N > SSSPY_MAX_SOURCES)
    return fail(SSSPY_ERR_UNSUPPORTED, "ILRMA: partitioning takes up to 8 sources");
  SSSPY_REQUIRE((W && C && !Y) || (Y && !W), "partition_normalize: pass (W, C) or Y");
  if (K > SSSPY_MAX_PARTITION_BASIS)
    return fail(SSSPY_ERR_UNSUPPORTED, "ILRMA: partitioning takes n_basis up to 1024");
  const IlrmaWs w = ilrma_ws(B, N, F, T, K);
  SSSPY_REQUIRE(workspace && workspace_bytes >= w.total, "partition_normalize: workspace too small");
  char *ws = (char *)workspace;
  double *qbuf = (double *)(ws + w.qbuf), *psi = (double *)(ws + w.psi);
  hipStream_t st = as_stream(stream);
  if (W) {
    int rc = row_power(W, C, qbuf, B, F, N, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_norm_scale, dim3((F + 63) / 64, B), dim3(256), 0, st, (c128 *)W,
                       (double *)nullptr, (const double *)qbuf, N, F, K, domain, floor_kind,
                       floor_eps, psi);
    rc = check_launch("k_norm_scale");
    if (rc) return rc;
  } else {
    // (qbuf holds B F N doubles: room for the B N ceil(F / 16) power slots)
    const int slots = (F + 15) / 16;
    dim3 grid(F, N, B), block(256);
    hipLaunchKernelGGL(k_output_power, dim3(slots, N, B), block, 0, st, (const c128 *)Y, qbuf, N, F,
                       T);
    hipLaunchKernelGGL(k_ilrma_normalize_output, grid, block, 0, st, (c128 *)Y, (double *)nullptr,
                       (const double *)qbuf, slots, N, F, T, K, domain, floor_kind, floor_eps, psi,
                       (double *)nullptr);
    int rc = check_launch("k_ilrma_normalize_output");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_partition_normalize, dim3(B), dim3(256), 0, st, basis, latent,
                     (const double *)psi, N, F, K, domain);
  return check_launch("k_partition_normalize");
}

}  // extern "C"
