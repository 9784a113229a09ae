// Run-time source counts above SSSPY_MAX_SOURCES (wide_n.hip): launchers the entry points fall back to.
#pragma once

#include "common.hpp"

namespace ssspy {

bool rt_sources_ok(int N);  // SSSPY_MAX_SOURCES < N <= SSSPY_RT_MAX_SOURCES
int rt_separate(const void *X, const void *W, void *Y, int B, int N, int F, int T, bool power,
                hipStream_t st);
// C[b, i, s] = (1/T) sum_j w_sj A conj(Bm)^T; Bm == A for the weighted auto-covariance
int rt_covariance(const void *A, const void *Bm, const double *weight, int kind, void *C, int B,
                  int N, int S, int F, int T, hipStream_t st);
int rt_ip1(void *W, const void *U, const void *C, double *qbuf, int B, int F, int N, int floor_kind,
           double floor_eps, int *info, hipStream_t st);
int rt_row_power(const void *W, const void *C, double *qbuf, int B, int F, int N, hipStream_t st);
int rt_iss1_transform(const void *Vc, void *G, int B, int F, int N, int floor_kind, double floor_eps,
                      hipStream_t st);
int rt_pb_filter(void *W, void *G, int B, int F, int N, int ref, int *info, hipStream_t st);
int rt_pb_scale(const void *XY, const void *YY, void *G, int B, int F, int N, int ref, int *info,
                hipStream_t st);
int rt_demix_from_cov(const void *YX, const void *XX, void *W, int B, int F, int N, int *info,
                      hipStream_t st);
int rt_sum_logdet(const void *W, double *out, int B, int F, int N, hipStream_t st);
size_t rt_ilrma_loss_ws_bytes(int B, int N, int F);
// Gauss model only: out[b] = sum_{n,i} mean_j (|y|^2 / R^(2/p) + (2/p) log R); ws: rt_ilrma_loss_ws_bytes
int rt_ilrma_loss(const void *X, const void *W, const double *basis, const double *act, double *out,
                  void *ws, int B, int N, int F, int T, int K, double domain, hipStream_t st);
int rt_frame_power(const void *X, const void *W, double *dst, int B, int N, int F, int T,
                   int bins_per_chunk, int chunks, hipStream_t st);

}  // namespace ssspy
