/*
 * ssspy_amd.h -- C ABI of the MI355X (gfx950) hot-path library libssspy_amd.so
 *
 * Drop-in boundary for the iterative frequency-domain demixing path of
 * tky823/ssspy v0.2.0.  The reference has no FFI layer:
 * the "operators" below are the NumPy expression groups inside the reference's
 * separator methods, and each entry point cites the reference lines it replaces.
 *
 * Conventions
 *  - every array argument is a DEVICE pointer to C-contiguous fp64 / complex128
 *    data (complex128 = two doubles, re then im, as numpy.complex128);
 *  - shapes carry a leading batch axis B of independent mixtures; the reference
 *    shapes follow it unchanged, e.g. X is (B, N, F, T), W is (B, F, N, N),
 *    basis (B, N, F, K), activation (B, N, K, T);
 *    N = n_sources = n_channels, F = n_bins, T = n_frames, K = n_basis;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *    call only enqueues work and returns; nothing here synchronises;
 *  - the callee never allocates or frees device memory: scratch is passed in
 *    (`*_workspace_bytes` tells how much) and outputs are caller-allocated;
 *  - return value: 0 on success, an SSSPY_ERR_* code otherwise
 *    (ssspy_last_error() gives the message); no C++ exception crosses the ABI;
 *  - singular per-bin systems (reference: numpy.linalg.LinAlgError from
 *    np.linalg.solve / inv) are counted into the caller's `int *info` device
 *    word; the host reads it when it next synchronises.
 *  - flooring (reference: ssspy/special/flooring.py:6-18) is named by
 *    (floor_kind, floor_eps): NONE = identity, MAX = max(x, eps), ADD = x + eps.
 */
#ifndef SSSPY_AMD_H
#define SSSPY_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SSSPY_OK = 0,
  SSSPY_ERR_BADARG = 1,      /* shape / enum outside what the kernels support */
  SSSPY_ERR_HIP = 2,         /* a HIP runtime call failed */
  SSSPY_ERR_UNSUPPORTED = 3, /* valid reference configuration not built here */
  SSSPY_ERR_INTERNAL = 4,    /* an invariant of the library itself was violated (a bug) */
};

enum { SSSPY_FLOOR_NONE = 0, SSSPY_FLOOR_MAX = 1, SSSPY_FLOOR_ADD = 2 };

/* weight layouts accepted by ssspy_weighted_covariance */
enum {
  SSSPY_WEIGHT_UNIT = 0,       /* weight == 1, one weight set                    */
  SSSPY_WEIGHT_FRAME = 1,      /* weight (B, S, T): broadcast over bins (AuxIVA) */
  SSSPY_WEIGHT_BIN_FRAME = 2,  /* weight (B, S, F, T)                            */
};

/* contrast functions of AuxIVA (reference: iva.py:3093-3115, :3256-3289) */
enum {
  SSSPY_CONTRAST_LAPLACE = 0,
  SSSPY_CONTRAST_GAUSS = 1,       /* refreshes variance = r2 / F, then G' = 2r / variance  */
  SSSPY_CONTRAST_GAUSS_FIXED = 2, /* uses the given variance unchanged (AuxGaussIVA IP2)   */
};

/* source models of ILRMA (reference classes GaussILRMA / TILRMA / GGDILRMA); `model_param` is
 * unused, the degree of freedom nu, or the shape beta respectively */
enum { SSSPY_SOURCE_GAUSS = 0, SSSPY_SOURCE_T = 1, SSSPY_SOURCE_GGD = 2 };
/* OR-ed into `source_model`: source_algorithm="ME" -- the same numerator / denominator sums with
 * exponent 1 (domain must be 2; Gauss and t models; ssspy/bss/ilrma.py:1249-1401, :2659-2830) */
enum { SSSPY_SOURCE_ME = 0x100 };

#define SSSPY_MAX_SOURCES 8 /* kernels compiled per source count (everything in registers) */
/* Above that, up to SSSPY_RT_MAX_SOURCES, the shared operators, the AuxIVA entry points and the ILRMA
 * iteration on the Gauss model's tuned passes run with the source count at run time (wide_n.hip:
 * correct, not tuned; the reference has no limit: ssspy/bss/ilrma.py:180, iva.py:152), IPA included
 * (ssspy_ipa_sweep, ipa_rt.hip, round 6), and so do the Hermitian operators / ssspy_solve (to
 * 16 x 16) and the standalone ssspy_lqpqm2 (to dimension 15; hermitian_rt.hip).  The MNMF entry
 * points and the ssspy_ilrma_partition_* entry points (partitioning=True) stay at SSSPY_MAX_SOURCES
 * (SSSPY_ERR_UNSUPPORTED above). */
#define SSSPY_RT_MAX_SOURCES 16
/* n_basis: the kernels walk any number of bases (dense products above 32; checked against the
 * oracle at 1500 and 3000); the bound only keeps 32-bit index arithmetic safe.  Up to round 5: 1024. */
#define SSSPY_MAX_BASIS 65536
#define SSSPY_MAX_PARTITION_BASIS 1024 /* partitioning=True: the latent update keeps N x n_basis in LDS */
#define SSSPY_MAX_PAIRS 128 /* every pair of 16 sources: 120 */

const char *ssspy_amd_version(void);

/* Bumped whenever an entry point changes its argument list (a caller built against another header
 * would pass a stream where a workspace is expected): the binding compares it with the value of
 * the header it was written against and refuses to load a library that disagrees.
 * 2: round 6 (ssspy_covariance_congruence_tracked added; round-5 signatures of
 *    ssspy_ilrma_loss_workspace_bytes / ssspy_fastmnmf_diagonalizer_covariance).
 * 3: round 6 (ssspy_ilrma_ip1_update_loss_slots: `logdet` became slots,
 *    ssspy_ilrma_deferred_logdet_slots added). */
#define SSSPY_ABI_VERSION 3
int ssspy_abi_version(void);
const char *ssspy_last_error(void);

/* ------------------------------------------------------------------ shared operators */

/* Y[b,n,i,j] = sum_m W[b,i,n,m] X[b,m,i,j].   X (B,N,F,T), W (B,F,N,N), Y (B,N,F,T).
 * In-place (Y == X) is allowed: every (bin, frame) column is read before it is written.
 * replaces: ssspy/bss/ilrma.py:272-295, ssspy/bss/iva.py:171-194 (separate). */
int ssspy_separate(const void *X, const void *W, void *Y, int B, int N, int F, int T,
                   void *stream);

/* Cout (B,F,N,N) = G C G^H per bin (C != Cout): the covariance of y' = G y from the covariance of y.
 * ILRMA's ISS2 / IPA iterations keep C_i = (1/T) sum_j y y^H of the separated spectrogram alongside it
 * (round 5): the power normalisation psi_n^2 = mean_i g_n^H C_i g_n of the UPDATED spectrogram is
 * then known before y <- G y runs, its scale goes into the rows of G, and the two extra passes of
 * ssspy/bss/ilrma.py:412-444 (mean |y|^2, y / psi) disappear.  n_sources <= 16. */
int ssspy_covariance_congruence(const void *C, const void *G, void *Cout, int B, int F, int N,
                                void *stream);

/* The same with S matrices per bin sharing its G: C, Cout (B,F,S,N,N), G (B,F,N,N).  The statistics
 * of the separated spectrogram from those of the mixture, mean phi y y^H = W (mean phi x x^H) W^H:
 * the ISS / ISS2 / IPA iterations of ILRMA then read the mixture through the filters they imply
 * (ssspy_compose_filters) like the IP iterations do, and y <- G y (ssspy/bss/ilrma.py:1635-1908)
 * is carried out once, when the output is read. */
int ssspy_covariance_congruence_sets(const void *C, const void *G, void *Cout, int B, int F, int S,
                                     int N, void *stream);

/* ssspy_covariance_congruence_sets for 2..4 sources that also reports how far the product can round
 * from the sum over the samples it replaces.  The direct mean phi |y_r|^2 of
 * ssspy/bss/_update_spatial_model.py:176-194 / :283-395 adds positive terms; the product cancels,
 * eps * S_rr (S_rr = sum_kl |g_rk| |c_kl| |g_rl|) bounds its error and kappa_r = S_rr / (G C G^H)_rr
 * is the loss of relative accuracy of that entry (infinite for a non-positive diagonal).  S == N:
 * set n of a bin holds the statistics under source n's weights, which steer output row n only
 * (y_n <- y_n - (V_n[n,r] / V_n[r,r]) y_r), so kappa_n = max_r kappa_r of set n weighs with that
 * row's power p_n = g_n P g_n^H.  power (B,F,N,N): the unweighted covariance of the data G applies
 * to.  amplification: 2 x (B,2) device doubles, zero before the first launch; the launch with
 * `phase` adds { sum kappa_n^2 p_n, sum p_n } over bins and sets to amplification[phase & 1][b] and
 * clears the other half for the next launch (alternate phase).
 * eps * sqrt([b][0] / [b][1]) estimates the relative Frobenius error the route adds to mixture b's
 * spectrogram per iteration; the separators poll the slots without waiting and return to the
 * reference's on-Y iteration past their bound. */
int ssspy_covariance_congruence_tracked(const void *C, const void *G, void *Cout, int B, int F,
                                        int S, int N, const void *power, void *amplification,
                                        int phase, void *stream);

/* out (B,F,N,N) = G W per bin (out aliases neither): the filters implied by y <- G y with y = W x. */
int ssspy_compose_filters(const void *G, const void *W, void *out, int B, int F, int N,
                          void *stream);

/* U[b,i,s,a,c] = (1/T) sum_j weight[...] A[b,a,i,j] conj(A[b,c,i,j]), s < S.
 * A (B,N,F,T) complex, U (B,F,S,N,N) complex, weight per `weight_kind`.
 * replaces: the (F,N,N,N,T) broadcast + mean of ssspy/bss/ilrma.py:1500-1505,
 * ssspy/bss/iva.py:1785-1791, ssspy/bss/mnmf.py:1504-1512. */
int ssspy_weighted_covariance(const void *A, const double *weight, int weight_kind, void *U,
                              int B, int N, int S, int F, int T, void *stream);

/* C[b,i,a,c] = (1/T) sum_j A[b,a,i,j] conj(Bm[b,c,i,j]).  A, Bm (B,N,F,T), C (B,F,N,N).  replaces the X Y^H / Y Y^H / X X^H products of
 * ssspy/algorithm/projection_back.py:104-110, ssspy/bss/ilrma.py:1941-1944,
 * ssspy/bss/iva.py:2182-2185 (up to the 1/T factor, which cancels there). */
int ssspy_cross_covariance(const void *A, const void *Bm, void *C, int B, int N, int F, int T,
                           void *stream);

/* One iterative-projection sweep, in place on W.  W (B,F,N,N), U (B,F,N,N,N).
 * `info` (device int, may be NULL) is incremented once per singular bin.
 * replaces: ssspy/bss/_update_spatial_model.py:17-78 (update_by_ip1, overwrite=True). */
int ssspy_update_by_ip1(void *W, const void *U, int B, int F, int N, int floor_kind,
                        double floor_eps, int *info, void *stream);

/* The same for a record_loss loop: the call also leaves sum_i log|det W_i| of the filters AS THEY
 * COME IN (the log-determinant term of the loss of the state the update starts from,
 * ssspy/bss/iva.py:200-222) as ssspy_update_by_ip1_logdet_slots() shares per mixture at
 * logdet[s * logdet_stride + b], to be added in slot order (ssspy_fold_scalar_slots): 1 where the
 * finished sums are stored, ceil(F / 16) for a handful of mixtures of up to 4 sources, where the
 * latency form of the kernel reads the filters anyway (shares the call does not write stay as the
 * caller zeroed them). */
int ssspy_update_by_ip1_logdet_slots(int B, int F, int N);
int ssspy_update_by_ip1_logdet(void *W, const void *U, int B, int F, int N, int floor_kind,
                               double floor_eps, int *info, double *logdet,
                               long long logdet_stride, void *stream);

/* The same sweep one source at a time, for a flooring_fn that is an arbitrary Python callable (the
 * reference accepts any, ssspy/bss/ilrma.py:70-89) and so cannot run in a kernel: the solve of source
 * `source_idx` leaves the unnormalised row conj(w) in W and denom (B,F) = sqrt(max(Re(w^H U_n w), 0));
 * the caller applies its callable to denom and divides the row by the result.
 * replaces: ssspy/bss/_update_spatial_model.py:63-76 (one iteration of the source loop). */
int ssspy_ip1_source_solve(void *W, const void *U, double *denom, int source_idx, int B, int F,
                           int N, int *info, void *stream);
int ssspy_scale_filter_row(void *W, const double *denom, int source_idx, int B, int F, int N,
                           void *stream);

/* One iterative-source-steering sweep expressed on per-bin statistics:
 * given Vc[b,i,s] = (1/T) sum_j varphi_s y y^H (B,F,N,N,N) it runs the N rank-1
 * steps of the reference on the N x N matrices and returns the accumulated
 * transform G (B,F,N,N) such that Y_new[:,i,:] = G_i Y[:,i,:].
 * replaces: ssspy/bss/_update_spatial_model.py:146-194 (update_by_iss1). */
int ssspy_iss1_transform(const void *Vc, void *G, int B, int F, int N, int floor_kind,
                         double floor_eps, void *stream);

/* Pairwise iterative projection, in place on W (B,F,N,N).  `pairs` is a HOST array of 2*n_pairs
 * ints (m0,n0,m1,n1,...), walked in order inside the kernel.  pair_only == 0: U (B,F,N,N,N) holds
 * one covariance per source; pair_only != 0: U (B,F,2,N,N) holds the pair's own two covariances and
 * n_pairs must be 1 (the AuxIVA form, where the weights are recomputed for every pair).  Rows come out with the arbitrary
 * phase of the 2x2 eigenvectors (fixed by scale restoration).
 * replaces: ssspy/bss/_update_spatial_model.py:81-143 (update_by_ip2), :317-395 (one pair). */
int ssspy_update_by_ip2(void *W, const void *U, int pair_only, const int *pairs, int n_pairs, int B,
                        int F, int N, int floor_kind, double floor_eps, int *info, void *stream);

/* Pairwise iterative source steering on per-bin statistics Vc (B,F,N,N,N) (as ssspy_iss1_transform):
 * returns G (B,F,N,N) with Y_new[:,i,:] = G_i Y[:,i,:].
 * replaces: ssspy/bss/_update_spatial_model.py:197-314 (update_by_iss2). */
int ssspy_iss2_transform(const void *Vc, void *G, const int *pairs, int n_pairs, int B, int F,
                         int N, int floor_kind, double floor_eps, int *info, void *stream);

/* The pairwise updates for a flooring callable that cannot run in a kernel (the reference takes any
 * callable: ssspy/bss/_update_spatial_model.py:81-143, :197-314).  ONE pair per call, rows left
 * UNNORMALISED, denom (B, F, 2) f64 <- sqrt(max(h^H G h, 0)) of the pair's two members; the caller
 * applies its callable to each (n_bins,) slice on the host and divides the rows with
 * ssspy_scale_filter_row (IP2: rows pair[0], pair[1] of W; ISS2: the same rows of the transform G,
 * which `accumulate` continues from its current value instead of the identity). */
int ssspy_update_by_ip2_deferred(void *W, const void *U, int pair_only, const int *pair, int B,
                                 int F, int N, double *denom, int *info, void *stream);
int ssspy_iss2_transform_deferred(const void *Vc, void *G, const int *pair, int accumulate, int B,
                                  int F, int N, double *denom, int *info, void *stream);

/* Largest n_frames the fused ISS kernel holds in registers for N sources (0 if N unsupported). */
int ssspy_iss1_fused_max_frames(int N);

/* One whole update_by_iss1 sweep set, in place on Y (B,N,F,T): a bin's N x T slab stays in the
 * registers of one workgroup through the N rank-1 steps (one read + one write of Y).
 * weight: (B,N,T) for SSSPY_WEIGHT_FRAME, (B,N,F,T) for SSSPY_WEIGHT_BIN_FRAME.
 * r2_next (B,N,T), optional: receives sum_i |y_new|^2 -- the frame powers the next AuxIVA iteration
 * needs, saving its separate pass.  They are summed without atomics (every block leaves its bins' sums
 * in `workspace`, a second kernel adds them in block order), so the result -- and with it the
 * trajectory -- is the same on every run; `workspace` (ssspy_iss1_fused_workspace_bytes) is needed
 * with r2_next (and by the tracked form below).
 * replaces: ssspy/bss/_update_spatial_model.py:146-194 (update_by_iss1). */
size_t ssspy_iss1_fused_workspace_bytes(int B, int N, int F, int T);
int ssspy_iss1_fused(void *Y, const double *weight, int weight_kind, double *r2_next, int B, int N,
                     int F, int T, int floor_kind, double floor_eps, void *workspace,
                     size_t workspace_bytes, void *stream);

/* The same sweep set, also tracking the log-determinant of the demixing filter the ISS state never
 * forms: the sweep of source n multiplies W_i by (I - v e_n^T), det = d_in^(-1/2), so
 * logdet[b] += -1/2 sum_i sum_n log d_in (logdet: B doubles holding sum_i log|det W_i| of the Y
 * passed in; NOT zeroed by the call; the blocks' shares go through `workspace` and are added in
 * block order: no atomics).  Lets compute_loss() (ssspy/bss/iva.py:2177-2192) skip the
 * reconstruction of W from Y X^H -- two more passes per recorded loss. */
int ssspy_iss1_fused_tracked(void *Y, const double *weight, int weight_kind, double *r2_next, int B,
                             int N, int F, int T, int floor_kind, double floor_eps, double *logdet,
                             void *workspace, size_t workspace_bytes, void *stream);

/* W <- W * (W^-1)[ref,:]^T.  W (B,F,N,N) in place.  G (B,F,N,N), optional: receives
 * diag((W^-1)[ref, :]), the scales (projection-back normalisation needs them for the basis).
 * replaces: ssspy/algorithm/projection_back.py:87-99. */
int ssspy_projection_back_filter(void *W, void *G, int B, int F, int N, int reference_id, int *info,
                                 void *stream);

/* minimal distortion principle: G[b,i] = diag(conj(z_n)), z_n = sum_j y_n conj(x_ref) / sum_j |y_n|^2,
 * from YX = ssspy_cross_covariance(Y, X) and YY = ssspy_cross_covariance(Y, Y); apply with
 * ssspy_separate(Y, G).   replaces: ssspy/algorithm/minimal_distortion_principle.py:6-43. */
int ssspy_mdp_scale(const void *YX, const void *YY, void *G, int B, int F, int N, int reference_id,
                    void *stream);

/* basis[b,n,i,:] *= |G[b,i,n,n]|^domain, the basis side of normalization="projection_back".
 * replaces: ssspy/bss/ilrma.py:518-522. */
int ssspy_ilrma_scale_basis(double *basis, const void *G, int B, int N, int F, int K, double domain,
                            void *stream);

/* scale[b,i,n] = ((X Y^H)(Y Y^H)^-1)[ref, n] from XY (B,F,N,N) and YY (B,F,N,N);
 * G[b,i] = diag(scale) so that ssspy_separate(Y, G) applies it.
 * replaces: ssspy/algorithm/projection_back.py:100-121. */
int ssspy_projection_back_scale(const void *XY, const void *YY, void *G, int B, int F, int N,
                                int reference_id, int *info, void *stream);

/* W[b,i] = YX[b,i] (XX[b,i])^-1.   replaces: ssspy/bss/ilrma.py:1941-1944, iva.py:2182-2185 */
int ssspy_demix_from_covariance(const void *YX, const void *XX, void *W, int B, int F, int N,
                                int *info, void *stream);

/* out[b] = sum_i log|det W[b,i]|.   W (B,F,N,N), out (B) doubles.
 * replaces: np.linalg.slogdet at ssspy/bss/ilrma.py:534, iva.py:234, mnmf.py:1274. */
int ssspy_sum_logdet(const void *W, double *out, int B, int F, int N, void *stream);

/* ------------------------------------------------------------------ batched small linear algebra */
/* Device counterparts of ssspy.linalg / ssspy.special.psd: `n` independent matrices, one per lane.
 * Sizes up to 8 x 8 are instantiated per size; 9 x 9 .. SSSPY_RT_MAX_SOURCES (16) take the size at
 * run time (csrc/hermitian_rt.hip: correct, not tuned); larger ones are SSSPY_ERR_UNSUPPORTED. */

/* X = A^-1 B.  A (n,N,N), B (n,N,nrhs), X (n,N,nrhs) complex128; N <= 16; LU with partial pivoting.
 * replaces: ssspy/linalg/_solve.py:9-21 (np.linalg.solve). */
int ssspy_solve(const void *A, const void *Bm, void *X, long long n, int N, int nrhs, int *info,
                void *stream);

/* closed-form 2x2 inverse.  replaces: ssspy/linalg/inv.py:4-54 (inv2). */
int ssspy_inv2(const void *A, void *out, long long n, void *stream);

/* Hermitian eigen-decomposition (cyclic complex Jacobi): lamb (n,M) ascending, V (n,M,M) unit
 * eigenvectors in columns (phase convention differs from LAPACK; A V = V diag(lamb) holds).
 * M = 2 (and ssspy_eigh2 below): the eigenvectors carry the phases reference LAPACK's zheevd gives
 * them (zhetd2 + dlaev2, restated in csrc/eigh2.hpp) -- what np.linalg.eigh returns in the
 * reference's eigh2 and pairwise updates (ssspy/linalg/eigh.py:155-157, :198).
 * replaces: np.linalg.eigh at ssspy/linalg/eigh.py:77,157,198 and ssspy/special/psd.py:54. */
int ssspy_eigh(const void *A, double *lamb, void *V, long long n, int M, void *stream);

/* Hermitise, floor the eigenvalues, rebuild, Hermitise.  replaces: ssspy/special/psd.py:11-71. */
int ssspy_to_psd(const void *A, void *out, long long n, int M, int floor_kind, double floor_eps,
                 void *stream);
/* out = P diag(w) P^H, Hermitised when `hermitise`: the rebuild half of to_psd / invsqrtmh for an
 * eigenvalue map the kernels cannot run (any flooring callable: ssspy_eigh -> the callable on the
 * (n, M) eigenvalues on the host -> this).  P (n, M, M) c128, w (n, M) f64, any M.
 * replaces: ssspy/special/psd.py:54-69, ssspy/linalg/sqrtm.py:58-64. */
int ssspy_herm_rebuild(const void *P, const double *w, void *out, long long n, int M, int hermitise,
                       void *stream);

/* generalised 2x2 Hermitian eigenproblem via Cholesky of B; type 1: A z = l B z, 2: A B z = l z,
 * 3: B A z = l z.  lamb (n,2) ascending, Z (n,2,2).  `info` counts non-positive-definite B.
 * replaces: ssspy/linalg/eigh.py:84-207 (eigh2 / _eigh). */
int ssspy_eigh2(const void *A, const void *Bm, double *lamb, void *Z, long long n, int type,
                int *info, void *stream);

/* ------------------------------------------------------------------ GaussILRMA (IP1/ISS1, MM) */

/* Scratch (bytes) for the ILRMA entry points below: one buffer of this size serves all of them
 * (each uses its own region: bin-chunk partials of the activation pass, frame-chunk partials of
 * the basis / covariance passes for small batches, per-bin powers for the normalisation). */
size_t ssspy_ilrma_workspace_bytes(int B, int N, int F, int T, int K);

/* The ILRMA entry points take the source model as (source_model, model_param):
 *   GAUSS: numerator P/R^((p+2)/p), exponent p/(p+2), varphi = 1/R^(2/p)       (ilrma.py:582-1989)
 *   T    : numerator P/(R~ R), R~ = nu/(nu+2) R^(2/p) + 2/(nu+2) P, varphi = 1/R~  (:1992-3334)
 *   GGD  : numerator (beta/2) P^(beta/2)/R^((beta+p)/p), exponent p/(beta+p),
 *          varphi = 1/((2/beta) floor(P^((2-beta)/2)) R^(beta/p))                (:3337-4410)
 * with R = (T V)_nij, P = |y_nij|^2, p = domain. */

/* basis update: T <- floor(T * (sum_j V num / sum_j V/R)^expo).  y = W x, or y = X when W == NULL
 * (the ISS state passes the separated spectrogram).
 * replaces: ssspy/bss/ilrma.py:1051-1128, :2470-2521, :3745-3824 (update_basis_mm, no partitioning). */
int ssspy_ilrma_update_basis(const void *X, const void *W, double *basis, const double *activation,
                             int B, int N, int F, int T, int K, double domain, int source_model,
                             double model_param, int floor_kind, double floor_eps, void *workspace,
                             size_t workspace_bytes, void *stream);

/* activation update (sum over bins with the NEW basis).
 * replaces: ssspy/bss/ilrma.py:1130-1204, :2523-2600, :3826-3905 (update_activation_mm). */
int ssspy_ilrma_update_activation(const void *X, const void *W, const double *basis,
                                  double *activation, int B, int N, int F, int T, int K,
                                  double domain, int source_model, double model_param,
                                  int floor_kind, double floor_eps, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* U[b,i,n] = (1/T) sum_j varphi_nij x x^H  ->  U (B,F,N,N,N).  W is read only by the heavy-tailed
 * models (varphi depends on |w x|^2); the floor only by GGD.
 * replaces: ssspy/bss/ilrma.py:1494-1505, :2915-2942, :3990-4018 (weights + covariance broadcast). */
int ssspy_ilrma_weighted_covariance(const void *X, const void *W, const double *basis,
                                    const double *activation, void *U, int B, int N, int F, int T,
                                    int K, double domain, int source_model, double model_param,
                                    int floor_kind, double floor_eps, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* power normalisation from the static covariance C (B,F,N,N) = (1/T) sum_j x x^H:
 * psi_n = floor(sqrt(mean_i w_in^H C_i w_in)); W[:,n,:] /= psi_n; basis[n] /= psi_n^p.
 * replaces: ssspy/bss/ilrma.py:365-444 (normalize_by_power, demix-filter branch). */
int ssspy_ilrma_normalize_filter(void *W, const void *C, double *basis, int B, int N, int F, int K,
                                 double domain, int floor_kind, double floor_eps, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* same for the ISS state: psi from |Y|^2 directly, Y /= psi, basis /= psi^p.
 * frame_power (B,N,T), optional: sum_i |y_nij|^2 of the CURRENT Y, as ssspy_iss1_fused leaves it in
 * r2_next -- saves the pass over Y that computes the power; NULL: computed here (one partial sum
 * per 16 bin rows, added in order: no atomics).  workspace: B * N * ceil(F / 16) doubles (B * N with
 * frame_power); the buffer of ssspy_ilrma_workspace_bytes is large enough.
 * replaces: ssspy/bss/ilrma.py:365-444 (normalize_by_power, demix_filter is None branch). */
int ssspy_ilrma_normalize_output(void *Y, double *basis, const double *frame_power, int B, int N,
                                 int F, int T, int K, double domain, int floor_kind,
                                 double floor_eps, void *workspace, size_t workspace_bytes,
                                 void *stream);

/* The same, also moving the tracked sum_i log|det W_i| of the ISS state (ssspy_iss1_fused_tracked):
 * dividing source n by psi_n divides row n of every implied filter, logdet[b] -= F sum_n log psi_n.
 * replaces: ssspy/bss/ilrma.py:365-444 (normalize_by_power) + the filter rebuild of :1946-1965. */
int ssspy_ilrma_normalize_output_tracked(void *Y, double *basis, const double *frame_power, int B,
                                         int N, int F, int T, int K, double domain, int floor_kind,
                                         double floor_eps, void *workspace, size_t workspace_bytes,
                                         double *logdet, void *stream);

/* varphi[b,n,i,j] (B,N,F,T) doubles for the ISS paths; Y (the separated spectrogram) is read only
 * by the heavy-tailed models.
 * replaces: ssspy/bss/ilrma.py:1690-1694, :3125-3143, :4202-4220. */
int ssspy_ilrma_iss_weight(const void *Y, const double *basis, const double *activation,
                           double *varphi, int B, int N, int F, int T, int K, double domain,
                           int source_model, double model_param, int floor_kind, double floor_eps,
                           void *stream);

/* The same from the POWER of the separated spectrogram, Ypow (B,N,F,T) f64 = |y|^2, instead of y.
 * Also the seam for a flooring callable the kernels do not know on GGD's weights (round 6): the
 * reference floors |y|^(2 - beta) per element (ssspy/bss/ilrma.py:3993-3995, :4123-4125); the
 * binding evaluates q = |y|^(2 - beta) and the callable on the host and hands in
 * Ypow = callable(q)^(2 / (2 - beta)) with SSSPY_FLOOR_NONE. */
int ssspy_ilrma_iss_weight_power(const double *Ypow, const double *basis, const double *activation,
                                 double *varphi, int B, int N, int F, int T, int K, double domain,
                                 int source_model, double model_param, int floor_kind,
                                 double floor_eps, void *stream);

/* out[b] = sum_{n,i} mean_j ( data term of the model + (2/p) log R ), y = W x (or x when
 * W == NULL); `out` (B doubles) is overwritten.  The caller adds -2 * ssspy_sum_logdet.
 * The per-workgroup shares are parked in `workspace` (ssspy_ilrma_loss_workspace_bytes for this
 * n_basis, with_filter = (W != NULL); the workspace of ssspy_ilrma_workspace_bytes is large enough
 * too) and added up in a fixed order:
 * no fp64 atomics, the same bits on every run.
 * replaces: ssspy/bss/ilrma.py:1946-1965, :3291-3310, :4367-4386. */
size_t ssspy_ilrma_loss_workspace_bytes(int B, int N, int F, int T, int K, int with_filter);
int ssspy_ilrma_loss_data(const void *X, const void *W, const double *basis,
                          const double *activation, double *out, int B, int N, int F, int T, int K,
                          double domain, int source_model, double model_param, void *workspace,
                          size_t workspace_bytes, void *stream);

/* One whole update_once() of an ILRMA with spatial_algorithm="IP1", source_algorithm="MM": basis,
 * activation, weighted covariance, IP1, power normalisation -- the launches the host would otherwise
 * issue one by one.  U (B,F,N,N,N) is caller-provided scratch.
 * replaces: ssspy/bss/ilrma.py:900-922 (and the TILRMA / GGDILRMA equivalents). */
int ssspy_ilrma_ip1_update(const void *X, const void *C, void *W, double *basis, double *activation,
                           void *U, int B, int N, int F, int T, int K, double domain,
                           int source_model, double model_param, int normalize, int floor_kind,
                           double floor_eps, void *workspace, size_t workspace_bytes, int *info,
                           void *stream);

/* The same update_once(), which also leaves the negative log-likelihood of the state AT ENTRY:
 * loss_data[b] (B doubles, overwritten) = the data term ssspy_ilrma_loss_data would give,
 * logdet[b] = ssspy_sum_logdet(W) before W is rewritten; loss = loss_data - 2 logdet.  The data term
 * is a by-product of the basis pass (which forms |y|^2 and R under the same parameters), so a loop
 * with record_loss=True costs three passes over X per iteration, not four: the loss after iteration t
 * comes out of iteration t + 1, and only the last one needs ssspy_ilrma_loss_data.  Returns
 * SSSPY_ERR_UNSUPPORTED (before touching any state) where the tuned kernels have no such
 * by-product: n_basis > 16, n_sources > 4, domains other than 1 and 2 (and domain 1 with a model
 * other than Gauss / MM), the Student-t model (its data term is not linear in the pass's
 * accumulators), and mixtures with n_bins * n_frames * 16 bytes >= 4 GiB.
 * replaces: ssspy/bss/base.py:68-77 around ssspy/bss/ilrma.py:900-922 and :1946-1965. */
/* 1 when ssspy_ilrma_ip1_update_deferred_loss has the by-product for this shape and model, else 0. */
int ssspy_ilrma_deferred_loss_supported(int N, int F, int T, int K, double domain,
                                        int source_model);
int ssspy_ilrma_ip1_update_deferred_loss(const void *X, const void *C, void *W, double *basis,
                                         double *activation, void *U, int B, int N, int F, int T,
                                         int K, double domain, int source_model, double model_param,
                                         int normalize, int floor_kind, double floor_eps,
                                         void *workspace, size_t workspace_bytes, int *info,
                                         double *loss_data, double *logdet, void *stream);

/* The same with the by-product left as RAW slots (round 5): slot s of mixture b at
 * slots[s * slot_stride + b], s < ssspy_ilrma_deferred_loss_slots() (0: no by-product for this shape).
 * A run of n_iter iterations allocates one zeroed array of slots x (n_iter + 1) B doubles, passes
 * slots + t B with slot_stride = (n_iter + 1) B in iteration t, and folds all iterations' slots in
 * slot order with one ssspy_fold_scalar_slots(slots, (n_iter + 1) B, n_slots, out, ...) at the end --
 * instead of a memset, a counter memset and a fold launch per iteration.
 * logdet (round 6) is laid out the same way: ssspy_ilrma_deferred_logdet_slots() shares per mixture
 * at logdet[s * slot_stride + b], to be folded in slot order like the data slots -- 1 where the
 * call leaves the finished sums (logdet[b]), ceil(F / 16) for a handful of mixtures, where the IP1
 * kernel of the latency path leaves the log-determinants of the filters it reads, per tile of 16
 * bins, instead of a dedicated one-block launch per iteration (0: no by-product for this shape). */
int ssspy_ilrma_deferred_loss_slots(int B, int N, int F, int T, int K, double domain,
                                    int source_model);
int ssspy_ilrma_deferred_logdet_slots(int B, int N, int F, int T, int K, double domain,
                                      int source_model);
int ssspy_ilrma_ip1_update_loss_slots(const void *X, const void *C, void *W, double *basis,
                                      double *activation, void *U, int B, int N, int F, int T,
                                      int K, double domain, int source_model, double model_param,
                                      int normalize, int floor_kind, double floor_eps,
                                      void *workspace, size_t workspace_bytes, int *info,
                                      double *slots, long long slot_stride, double *logdet,
                                      void *stream);
/* out[e] = sum over s < nslots of slots[s * total + e], e < total, in slot order (deterministic) */
size_t ssspy_fold_scalar_slots_workspace_bytes(long long total, int nslots);
int ssspy_fold_scalar_slots(const double *slots, long long total, int nslots, double *out,
                            void *workspace, size_t workspace_bytes, void *stream);

/* IPA (iterative projection with adjustment): a whole sweep on per-bin statistics (round 5; the
 * per-source entry point ssspy_ipa_transform of rounds 3-5 went in round 6 with its 81 kernel
 * instantiations).  n_sources in [2, 16]: a lane per bin up to 6 sources, a bin on 8 lanes at 7 / 8,
 * and above 8 a lane per bin with the source count at run time and its working set in scratch
 * memory (ipa_rt.hip, round 6: correct, not tuned).  Vc (B,F,N,N,N) = ssspy_weighted_covariance(Y,
 * weight) of the spectrogram BEFORE the sweep (overwritten: after source step s it holds
 * G_s Vc G_s^H, the covariances of the spectrogram the reference would have formed by then); G
 * (B,F,N,N) <- G_{N-1} ... G_0.  The caller applies ssspy_separate(Y, G) once: three passes over Y per
 * sweep (weights, covariance, separate) instead of 3 N.  The weights are those of the sweep's start
 * for every source, as in the reference (_update_spatial_model.py:436-445: varphi is an argument).
 * newton_ws: ssspy_ipa_sweep_newton_words(B, N) 64-bit words of scratch.  The reference's Newton loop
 * runs over all bins of a mixture at once and stops at the first step at which every bin has
 * converged (linalg/lqpqm.py:196-213); with the scratch the bins of a mixture vote and make exactly
 * as many steps (max_iter <= 62); with NULL every bin makes max_iter steps.  not_converged
 * (optional): incremented once per mixture whose bins had not all converged after max_iter steps
 * (the reference's UserWarning).  info: incremented per singular per-bin system.
 * replaces: ssspy/bss/_update_spatial_model.py:398-513 (update_by_ipa), linalg/lqpqm.py:13-352
 * (lqpqm2, solve_equation). */
int ssspy_ipa_sweep(void *Vc, void *G, int B, int F, int N, int normalization, int max_iter,
                    int floor_kind, double floor_eps, int *info, void *newton_ws,
                    int *not_converged, void *stream);
/* 64-bit words ssspy_ipa_sweep's newton_ws must hold: a vote word and an arrival counter per
 * (mixture, source step).  Up to 4 sources the sweep is ONE launch (round 6): a workgroup walks
 * the N source steps of its 64 bins and meets the mixture's other workgroups at every Newton vote
 * instead of ending the kernel there (4 launches per source step before). */
size_t ssspy_ipa_sweep_newton_words(int B, int N);
/* Development aid: how many of those in-kernel meetings gave up waiting since the library was
 * loaded (0 unless a launch is broken; synchronises).  -1: the query itself failed. */
int ssspy_debug_barrier_timeouts(void);

/* ---- partitioning (latent variables): basis (B,F,K), activation (B,K,T), latent (B,N,K) with
 * R_nij = sum_k z_nk t_ik v_kj (ssspy/bss/ilrma.py:297-327).  Every entry point above that takes
 * (basis, activation) runs unchanged on the expanded pair
 *   Teff[b,n,i,k] = z_nk t_ik (B,N,F,K),  Vrep[b,n,k,j] = v_kj (B,N,K,T). */
int ssspy_ilrma_partition_expand(const double *basis, const double *activation,
                                 const double *latent, double *Teff, double *Vrep, int B, int N,
                                 int F, int T, int K, void *stream);

enum { SSSPY_PARTITION_LATENT = 1, SSSPY_PARTITION_BASIS = 2, SSSPY_PARTITION_ACTIVATION = 4 };

/* The selected source-model updates in the reference's order (latent, basis, activation), each
 * from fresh sums; Teff / Vrep are scratch on entry and hold the expansion of the final
 * parameters on return.  y = W x, or X itself when W == NULL.
 * replaces: ssspy/bss/ilrma.py:1007-1049 (update_latent_mm), :1051-1128, :1130-1204 with
 * partitioning=True, their ME forms (:1206-1401) and the TILRMA / GGDILRMA equivalents. */
int ssspy_ilrma_partition_update(const void *X, const void *W, double *basis, double *activation,
                                 double *latent, double *Teff, double *Vrep, int B, int N, int F,
                                 int T, int K, double domain, int source_model, double model_param,
                                 int steps, int floor_kind, double floor_eps, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* power normalisation with partitioning: psi_n as above from (W, C) -- pass Y == NULL -- or from
 * the separated spectrogram Y (W == C == NULL); W rows (or Y) /= psi_n;
 * z_nk <- (z_nk / psi_n^p) / s_k, t_ik <- t_ik s_k with s_k = sum_n z_nk / psi_n^p.
 * replaces: ssspy/bss/ilrma.py:365-444 (normalize_by_power, partitioning branch). */
int ssspy_ilrma_partition_normalize(void *W, const void *C, void *Y, double *basis, double *latent,
                                    int B, int N, int F, int T, int K, double domain,
                                    int floor_kind, double floor_eps, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ AuxIVA (IP1/ISS1) */

/* r2[b,n,j] = sum_i |y_nij|^2 with y = W x (or y = X when W == NULL).   r2 (B,N,T).
 * Summed without atomics (bin chunks leave partial sums in `workspace`, folded in chunk order): the
 * same bits on every run.  workspace: ssspy_iva_frame_power_workspace_bytes (0 for large batches).
 * replaces: np.linalg.norm(Y, axis=1) at ssspy/bss/iva.py:1787,1962 (squared). */
size_t ssspy_iva_frame_power_workspace_bytes(int B, int N, int F, int T);
int ssspy_iva_frame_power(const void *X, const void *W, double *r2, int B, int N, int F, int T,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Y = W X (Y may be X: in place) AND r2[b,n,j] = sum_i |y_nij|^2 of the result in one walk (round 5):
 * the separate() that ends an AuxIVA ISS2 / IPA step and the frame-power pass of the next iteration's
 * weights (ssspy/bss/iva.py:1968-2066 followed by :1917-1935).  Workspace as ssspy_iva_frame_power.
 * n_sources <= 8. */
int ssspy_separate_frame_power(const void *X, const void *W, void *Y, double *r2, int B, int N,
                               int F, int T, void *workspace, size_t workspace_bytes,
                               void *stream);

/* weight[b,n,j] = G'(r)/floor(2 r), r = sqrt(r2); Gauss also refreshes variance = r2 / F.
 * replaces: ssspy/bss/iva.py:1788-1789, :1963-1964, :3105-3115, :3273-3289, :3465-3473. */
int ssspy_iva_weight(const double *r2, double *weight, double *variance, int B, int N, int F, int T,
                     int contrast, int floor_kind, double floor_eps, void *stream);

/* out[b] = sum_n mean_j G(y_nj) from r2 (and variance for Gauss).
 * replaces: ssspy/bss/iva.py:216-219, :2181-2187 (contrast part of the loss). */
int ssspy_iva_loss_data(const double *r2, const double *variance, double *out, int B, int N, int F,
                        int T, int contrast, void *stream);

/* ------------------------------------------------------------------ FastGaussMNMF (IP1) */

/* steps of FastGaussMNMF.update_once, OR-ed into `steps` of ssspy_fastmnmf_update */
enum {
  SSSPY_MNMF_BASIS = 1,
  SSSPY_MNMF_ACTIVATION = 2,
  SSSPY_MNMF_DIAGONALIZER = 4,
  SSSPY_MNMF_SPATIAL = 8,
  SSSPY_MNMF_NORMALIZE = 16,
  SSSPY_MNMF_ALL = 31,
};

size_t ssspy_fastmnmf_workspace_bytes(int B, int N, int M, int F, int T, int K);

/* The selected steps of update_once(), in the reference's order: basis, activation,
 * diagonaliser (IP1), spatial, power normalisation.
 * X (B,M,F,T), Q (B,F,M,M), D (B,F,N,M), basis (B,N,F,K), activation (B,N,K,T),
 * C (B,F,M,M) static covariance of X (needed by the normalisation step only).
 * replaces: ssspy/bss/mnmf.py:1278-1303 and :1305-1360, :1362-1417, :1449-1514, :1635-1675,
 * :632-678. */
int ssspy_fastmnmf_update(const void *X, const void *C, void *Q, double *D, double *basis,
                          double *activation, int B, int N, int M, int F, int T, int K, int steps,
                          int floor_kind, double floor_eps, void *workspace, size_t workspace_bytes,
                          int *info, void *stream);

/* The same steps with the |(Q x)_m|^2 hand-over: the reference recomputes Q x in every one of
 * update_basis / update_activation / update_spatial (mnmf.py:1329-1331, :1386-1388, :1659-1661);
 * here the spatial pass stores |Q x|^2 and the next iteration's basis and activation passes read
 * it (half the bytes of x, no M x M products).  `handover`: ssspy_fastmnmf_handover_doubles()
 * doubles owned by the caller -- |Q x|^2 (B,M,F,T) followed by a per-(mixture, channel) scale
 * (B,M) that absorbs the power normalisation.  *handover_valid (host): on entry, whether the
 * buffer matches the Q and X passed in (0 on the first call or after the caller changed Q or X:
 * it is then rebuilt when a step needs it); on return, whether it matches Q on exit.
 * ssspy_fastmnmf_handover_doubles() is 0 for shapes without the hand-over (use
 * ssspy_fastmnmf_update). */
size_t ssspy_fastmnmf_handover_doubles(int B, int N, int M, int F, int T, int K);
int ssspy_fastmnmf_update_handover(const void *X, const void *C, void *Q, double *D, double *basis,
                                   double *activation, int B, int N, int M, int F, int T, int K,
                                   int steps, int floor_kind, double floor_eps, void *workspace,
                                   size_t workspace_bytes, int *info, double *handover,
                                   int *handover_valid, void *stream);

/* The same (handover / handover_valid both NULL: as ssspy_fastmnmf_update) for a record_loss loop:
 * `steps` must hold SSSPY_MNMF_DIAGONALIZER, and the call also leaves sum_i log|det Q_i| of the
 * diagonalisers AS THEY COME IN -- the log-determinant term of the loss of the state the call
 * starts from (ssspy/bss/mnmf.py:1219-1261) -- as ssspy_fastmnmf_deferred_logdet_slots() shares per
 * mixture at logdet[s * logdet_stride + b], to be added in slot order (ssspy_fold_scalar_slots; a
 * run keeps one zeroed array of slots x (n_iter + 1) B doubles and passes logdet + t B): 1 where
 * the finished sums are stored, ceil(F / 16) for a handful of mixtures of up to 4 channels, where
 * the latency form of IP1 reads the diagonalisers anyway and leaves one share per tile of 16 bins
 * instead of a one-block ssspy_sum_logdet launch per iteration (shares the call does not write
 * stay as the caller zeroed them). */
int ssspy_fastmnmf_deferred_logdet_slots(int B, int N, int M, int F, int T, int K);
int ssspy_fastmnmf_update_handover_logdet(const void *X, const void *C, void *Q, double *D,
                                          double *basis, double *activation, int B, int N, int M,
                                          int F, int T, int K, int steps, int floor_kind,
                                          double floor_eps, void *workspace, size_t workspace_bytes,
                                          int *info, double *handover, int *handover_valid,
                                          double *logdet, long long logdet_stride, void *stream);

/* The data term of the loss (as ssspy_fastmnmf_loss_data) from a VALID hand-over buffer instead of
 * X and Q: half the bytes, no M x M products.  The caller guarantees that the buffer matches the
 * current Q and X (ssspy_fastmnmf_update_handover returned *handover_valid = 1 and neither moved
 * since).  out: B doubles, overwritten; workspace: ssspy_fastmnmf_loss_workspace_bytes (the
 * per-wave shares, added up in a fixed order: no fp64 atomics).
 * replaces: ssspy/bss/mnmf.py:1240-1258. */
int ssspy_fastmnmf_loss_data_handover(const double *D, const double *basis,
                                      const double *activation, const double *handover,
                                      double *out, int B, int N, int M, int F, int T, int K,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* The same with the per-wave shares left RAW: share s of mixture b at slots[s * slot_stride + b],
 * s < ssspy_fastmnmf_loss_handover_slots() (0: no hand-over for the shape).  A record_loss run
 * zeroes one array of slots x (n_iter + 1) B doubles, passes slots + t B with slot_stride =
 * (n_iter + 1) B for the state after iteration t, and folds all of them in slot order with one
 * ssspy_fold_scalar_slots at the end -- instead of two memsets and a fold launch per recorded loss
 * (shares a launch does not write stay as the caller zeroed them). */
int ssspy_fastmnmf_loss_handover_slots(int B, int N, int M, int F, int T, int K);
int ssspy_fastmnmf_loss_data_handover_slots(const double *D, const double *basis,
                                            const double *activation, const double *handover,
                                            double *slots, long long slot_stride, int B, int N,
                                            int M, int F, int T, int K, void *stream);

/* U[b,i,m] = (1/T) sum_j x x^H / R~_ijm  -> (B,F,M,M,M): the covariances the diagonaliser update
 * (IP1 inside ssspy_fastmnmf_update, or ssspy_update_by_ip2 for diagonalizer_algorithm="IP2") needs.
 * workspace: NULL, or ssspy_fastmnmf_workspace_bytes() of scratch -- with it the tuned pass of the
 * fused update runs (its split work items park partial sums there).
 * replaces: ssspy/bss/mnmf.py:1504-1512, :1621-1629. */
int ssspy_fastmnmf_diagonalizer_covariance(const void *X, const double *D, const double *basis,
                                           const double *activation, void *U, int B, int N, int M,
                                           int F, int T, int K, void *workspace,
                                           size_t workspace_bytes, void *stream);

/* weights[b,m,i,j] = 1 / R~_ijm, R~ = sum_n lambda_nij d_inm (B,M,F,T): the per-channel weights of the
 * diagonaliser covariance, U = ssspy_weighted_covariance(X, weights, SSSPY_WEIGHT_BIN_FRAME, S = M).
 * Any n_sources <= 8, n_channels in [2, 8].  replaces: ssspy/bss/mnmf.py:1489-1512 (the weight part). */
int ssspy_fastmnmf_weights(const void *X, const void *Q, const double *D, const double *basis,
                           const double *activation, double *weights, int B, int N, int M, int F,
                           int T, int K, void *stream);

/* out[b] = sum_i mean_j sum_m ( |q x|^2 / R~ + log R~ ) (overwritten); caller adds
 * -2 sum logdet Q.  The per-block shares go through `workspace`
 * (ssspy_fastmnmf_loss_workspace_bytes) and are added up in a fixed order: no fp64 atomics, the
 * same bits on every run.   replaces: ssspy/bss/mnmf.py:1240-1258. */
size_t ssspy_fastmnmf_loss_workspace_bytes(int B, int N, int M, int F, int T);
int ssspy_fastmnmf_loss_data(const void *X, const void *Q, const double *D, const double *basis,
                             const double *activation, double *out, int B, int N, int M, int F,
                             int T, int K, void *workspace, size_t workspace_bytes, void *stream);

/* Multichannel Wiener filter output Y (B,N,F,T) for reference channel `reference_id`:
 * per (bin, frame) an M x M Hermitian eigen-decomposition (Jacobi) with floored eigenvalues.
 * replaces: ssspy/bss/mnmf.py:1174-1217 (separate) incl. to_psd (special/psd.py:11-71). */
int ssspy_fastmnmf_separate(const void *X, const void *Q, const double *D, const double *basis,
                            const double *activation, void *Y, int B, int N, int M, int F, int T,
                            int K, int reference_id, int floor_kind, double floor_eps,
                            void *workspace, size_t workspace_bytes, int *info, void *stream);

/* The same filter split at the eigenvalue floor of to_psd, for a flooring callable the kernels do
 * not know (round 5): stage 1 leaves the eigenvalues of R_ij in ascending order, lam (B,F,T,M) -- what
 * numpy.linalg.eigh hands the reference's flooring_fn (special/psd.py:54-62) -- and the eigenvectors
 * P (B,F,T,M,M); the caller applies the callable to lam on the host; stage 2 (same workspace,
 * untouched in between: it holds Q^-1) rebuilds R^-1 from them and writes Y.  Any N, M <= 8.
 * replaces: ssspy/bss/mnmf.py:1174-1217 with an arbitrary flooring_fn. */
int ssspy_fastmnmf_separate_eig(const void *X, const void *Q, const double *D, const double *basis,
                                const double *activation, void *Y, int B, int N, int M, int F,
                                int T, int K, int reference_id, int stage, double *lam, void *P,
                                void *workspace, size_t workspace_bytes, int *info, void *stream);

/* ------------------------------------------------------------------ GaussMNMF (full-rank SCM)
 * State: basis (B,N,F,K) f64, activation (B,N,K,T) f64, spatial (B,N,F,M,M) c128 Hermitian PSD
 * (the reference's `spatial` (N,F,M,M) with a batch axis).  n_channels M in [2, 8]: one lane per
 * (bin, frame) point or per spatial matrix; from 4 channels on the point lives in one packed
 * Hermitian matrix inverted in place (herm_packed.hpp) and the full-storage kernels only redo the
 * blocks whose points leave the fast route of to_psd (flags in the workspace).
 * R_ij = to_psd(sum_n lambda_nij H_ni); the instantaneous covariance to_psd(x x^H) of
 * MNMFBase._init_instant_covariance (ssspy/bss/mnmf.py:167-188) is applied in closed form. */
enum {
  SSSPY_GMNMF_BASIS = 1,
  SSSPY_GMNMF_ACTIVATION = 2,
  SSSPY_GMNMF_SPATIAL = 4,
  SSSPY_GMNMF_NORMALIZE = 8,
  SSSPY_GMNMF_ALL = 15,
  SSSPY_GMNMF_LATENT = 16 /* partitioning only; runs last, as in update_once() */
};

size_t ssspy_gmnmf_workspace_bytes(int B, int N, int M, int F, int T, int K);

/* The steps of update_once() selected by `steps`, in the reference's order: basis, activation,
 * spatial (H <- to_psd(P^-1 # H Q H), the matrix geometric mean of linalg/mean.py:6-83 type 2),
 * unit-trace normalisation of H with the scale moved into the basis, latent variables.
 * latent == NULL: basis (B,N,F,K), activation (B,N,K,T).  latent (B,N,K) given (partitioning):
 * basis (B,F,K) and activation (B,K,T) are shared, lambda_nij = sum_k z_nk t_ik v_kj, and the
 * normalisation leaves the basis alone.  ssspy_gmnmf_loss / _separate take the per-source pair
 * that ssspy_ilrma_partition_expand writes.
 * replaces: ssspy/bss/mnmf.py:806-834, :836-901, :903-968, :970-1016, :391-414, :1018-1073. */
int ssspy_gmnmf_update(const void *X, double *basis, double *activation, double *latent,
                       void *spatial, int B, int N, int M, int F, int T, int K, int steps,
                       int floor_kind, double floor_eps, void *workspace, size_t workspace_bytes,
                       void *stream);

/* out[b] = sum_i mean_j ( tr(R^-1 XX) + log det R ); `out` (B doubles) is overwritten; the
 * per-block shares go through `workspace` (ssspy_gmnmf_loss_workspace_bytes) and are added up in
 * a fixed order (no fp64 atomics).
 * replaces: ssspy/bss/mnmf.py:765-804. */
size_t ssspy_gmnmf_loss_workspace_bytes(int B, int F, int T);
int ssspy_gmnmf_loss(const void *X, const double *basis, const double *activation,
                     const void *spatial, double *out, int B, int N, int M, int F, int T, int K,
                     int floor_kind, double floor_eps, void *workspace, size_t workspace_bytes,
                     void *stream);

/* multichannel Wiener filter: Y[b,n,i,j] = (lambda_nij H_ni R_ij^-1 x_ij)[reference_id]
 * -> Y (B,N,F,T) c128.  replaces: ssspy/bss/mnmf.py:729-763. */
int ssspy_gmnmf_separate(const void *X, const double *basis, const double *activation,
                         const void *spatial, void *Y, int B, int N, int M, int F, int T, int K,
                         int reference_id, int floor_kind, double floor_eps, void *stream);

/* ------------------------------------------------------------------ Hermitian matrix functions
 * One matrix per lane, M in [2, 16] (9 .. 16: the size at run time, csrc/hermitian_rt.hip); arrays
 * flattened to (n, M, M) c128.
 * generalised eigh through the Cholesky factor of B, eigenvalues ascending -> lamb (n, M), Z (n, M, M):
 *   type 1: A z = lamb B z; 2: A B z = lamb z; 3: B A z = lamb z.   replaces: ssspy/linalg/eigh.py:8-81, :164-207 */
int ssspy_eigh_general(const void *A, const void *Bm, double *lamb, void *Z, long long n, int M,
                       int type, int *info, void *stream);
/* inverse == 0: X^(1/2); inverse == 1: P diag(1 / floor(sqrt(lam))) P^H.  replaces: linalg/sqrtm.py:8-64 */
int ssspy_sqrtmh(const void *X, void *out, long long n, int M, int inverse, int floor_kind,
                 double floor_eps, void *stream);
/* geometric mean: type 1: A # B; 2: A^-1 # B; 3: A # B^-1.   replaces: ssspy/linalg/mean.py:6-83 */
int ssspy_gmeanmh(const void *A, const void *Bm, void *G, long long n, int M, int type,
                  void *stream);
/* y = argmin of the log-quadratically penalised quadratic (type 2): H (n, L, L) Hermitian, v (n, L),
 * z (n) -> y (n, L), L in [1, 15] (8 .. 15: the dimension at run time, csrc/ipa_rt.hip).
 * newton_ws (one 64-bit word of scratch) / not_converged: as for
 * ssspy_ipa_sweep, the n problems forming one group (the reference's loop stops when all of them
 * have converged); NULL: max_iter Newton steps per problem.
 * replaces: ssspy/linalg/lqpqm.py:13-352 (lqpqm2 with singular_fn = "x < flooring_fn(0)"). */
int ssspy_lqpqm2(const void *H, const void *v, const double *z, void *y, long long n, int L,
                 int max_iter, int floor_kind, double floor_eps, void *newton_ws,
                 int *not_converged, void *stream);
/* The same with the reference's `singular_fn` evaluated by the caller: singular (n ints, device) is
 * singular_fn(||v||) per problem -- None: ||v|| == 0, or any callable (the norms are n numbers the
 * host forms from its own v); problems flagged singular take the "v = 0" branch and stay out of the
 * joint convergence test.  replaces: ssspy/linalg/lqpqm.py:61-78 (singular_fn), :80-110. */
int ssspy_lqpqm2_masked(const void *H, const void *v, const double *z, void *y, long long n, int L,
                        int max_iter, int floor_kind, double floor_eps, void *newton_ws,
                        int *not_converged, const int *singular, void *stream);

/* ------------------------------------------------------------------ STFT / ISTFT
 * The transforms the reference's workflow takes from SciPy either side of a separator
 * (tests/package/bss/test_ilrma.py, test_iva.py, test_mnmf.py: scipy.signal.stft(x, window="hann",
 * nperseg=n_fft, noverlap=n_fft - hop) and scipy.signal.istft), with SciPy's defaults:
 * boundary="zeros", padded=True, one-sided, scaling="spectrum".  n_fft: a power of two <= 65536
 * or ANY length in [2, 32768], even or odd, as SciPy's nperseg (Bluestein's chirp-z on two radix-2
 * transforms).  Transforms of up to 8192 points run in LDS; longer ones (a power of two above 8192,
 * any other length above 4096) on workgroup-private slices of the workspace.  `workspace` =
 * ssspy_stft_workspace_bytes(n_fft) bytes on the device: the chirp's spectrum and those slices; 0 /
 * NULL for a power of two <= 8192.
 * x (B, C, n_samples) f64 -> Z (B, C, n_fft/2+1, ssspy_stft_frames(...)) c128; `window` (n_fft) f64
 * on the device, `window_sum` its sum. */
int ssspy_stft_frames(long long n_samples, int n_fft, int hop);
size_t ssspy_stft_workspace_bytes(int n_fft);
int ssspy_stft(const double *x, void *Z, const double *window, double window_sum, int B, int C,
               long long n_samples, int n_fft, int hop, void *workspace, void *stream);

/* Z (B, C, n_fft/2+1, n_frames) -> x (B, C, ssspy_istft_samples(...)); `segments` is scratch of
 * B*C*n_frames*n_fft doubles, `workspace` as for ssspy_stft. */
long long ssspy_istft_samples(int n_frames, int n_fft, int hop);
int ssspy_istft(const void *Z, double *x, const double *window, double window_sum,
                double *segments, int B, int C, int n_frames, int n_fft, int hop, void *workspace,
                void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSSPY_AMD_H */
