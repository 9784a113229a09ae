// Batched small dense linear algebra (one matrix per lane): solve, inv2, Hermitian eigh,
// generalised 2x2 eigh, PSD projection.  Device counterparts of ssspy.linalg / ssspy.special.psd.
#include <cstdlib>

#include "common.hpp"
#include "eigh2.hpp"
#include "hermitian.hpp"
#include "herm_packed.hpp"
#include "smallmat.hpp"

namespace ssspy {

// hermitian_rows.hip
bool hermitian_rows_wanted(int M, int always_from);
int eigh_rows(const void *A, double *lamb, void *V, long long n, int M, int mode, int floor_kind,
              double eps, hipStream_t st);
// hermitian_rt.hip: 9 x 9 .. 16 x 16, the size at run time
bool hermitian_rt_wanted(int M);
int solve_rt(const void *A, const void *Bm, void *X, long long n, int M, int nrhs, int *info,
             hipStream_t st);
int eigh_rt(const void *A, double *lamb, void *V, long long n, int M, int mode, int floor_kind,
            double eps, hipStream_t st);

// X = A^-1 B, B (N x nrhs)
template <int N>
__global__ __launch_bounds__(64) void k_solve(const c128 *__restrict__ A, const c128 *__restrict__ Bm,
                                              c128 *X, long long n, int nrhs, int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  Mat<N> Am, Inv;
  load_mat<N>(Am, A + idx * (N * N));
  const bool ok = invert<N>(Am, Inv);
  for (int c = 0; c < nrhs; ++c) {
    c128 col[N];
#pragma unroll
    for (int r = 0; r < N; ++r) col[r] = Bm[(idx * N + r) * nrhs + c];
#pragma unroll
    for (int r = 0; r < N; ++r) {
      c128 acc = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < N; ++k) cfma(acc, Inv.a[r][k], col[k]);
      X[(idx * N + r) * nrhs + c] = acc;
    }
  }
  if (!ok && info) atomicAdd(info, 1);
}

// closed-form 2x2 inverse (no pivoting), ref: ssspy/linalg/inv.py:4-54
__global__ __launch_bounds__(256) void k_inv2(const c128 *__restrict__ A, c128 *out, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const c128 a = A[idx * 4], b = A[idx * 4 + 1], c = A[idx * 4 + 2], d = A[idx * 4 + 3];
  const c128 det = csub(cmul(a, d), cmul(b, c));
  out[idx * 4] = cdiv(d, det);
  out[idx * 4 + 1] = cdiv(cmake(-b.x, -b.y), det);
  out[idx * 4 + 2] = cdiv(cmake(-c.x, -c.y), det);
  out[idx * 4 + 3] = cdiv(a, det);
}

// Hermitian eigen-decomposition, eigenvalues ascending (as numpy.linalg.eigh), unit eigenvectors.
// mode 0: write lamb (n, M) and V (n, M, M);  mode 1 (to_psd): floor eigenvalues, rebuild into V.
template <int M>
__global__ __launch_bounds__(64) void k_eigh(const c128 *__restrict__ A, double *lamb, c128 *V,
                                             long long n, int mode, int floor_kind, double eps) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  c128 Am[M][M], P[M][M];
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) Am[r][c] = A[(idx * M + r) * M + c];
  hermitize<M>(Am);
  jacobi_eigh<M>(Am, P);
  if (mode == 0) {
    // rank of every eigenvalue (ties broken by index) -> ascending order without dynamic indexing
#pragma unroll
    for (int k = 0; k < M; ++k) {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < M; ++j)
        rank += (Am[j][j].x < Am[k][k].x || (Am[j][j].x == Am[k][k].x && j < k)) ? 1 : 0;
      lamb[idx * M + rank] = Am[k][k].x;
#pragma unroll
      for (int r = 0; r < M; ++r) V[(idx * M + r) * M + rank] = P[r][k];
    }
  } else {
    c128 R[M][M];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int c = 0; c < M; ++c) {
        c128 acc = cmake(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < M; ++k) {
          const double ev = apply_floor(Am[k][k].x, floor_kind, eps);
          const c128 z = cmulc(P[r][k], P[c][k]);
          acc.x = fma(ev, z.x, acc.x);
          acc.y = fma(ev, z.y, acc.y);
        }
        R[r][c] = acc;
      }
    hermitize<M>(R);
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
      for (int c = 0; c < M; ++c) V[(idx * M + r) * M + c] = R[r][c];
  }
}

// The same for 6-8 x 6-8 on packed storage (herm_packed.hpp): the matrix being diagonalised takes M^2
// registers and the eigenvector rows ride the rotations in turns of 4 (from 7 on; each turn repeats
// the rotation sequence on a fresh copy), so nothing spills where k_eigh<8> parks 3 266 registers
// in scratch.  Same rotations in the same order as jacobi_eigh: the same eigen-pairs.
// mode 1 (to_psd) collects the rows in LDS ([entry][lane]) for the rebuild.
constexpr int EIGH_LD = 65;

template <int M>
__device__ __forceinline__ void eigh_load_packed(HermP<M> &Ap, const c128 *__restrict__ src) {
#pragma unroll
  for (int a = 0; a < M; ++a) {
    Ap.d[a] = src[a * M + a].x;
#pragma unroll
    for (int c = a + 1; c < M; ++c) {
      const c128 u = src[a * M + c], l = src[c * M + a];
      Ap.o[tri<M>(a, c)] = cmake(0.5 * (u.x + l.x), 0.5 * (u.y - l.y));
    }
  }
}

template <int M, int TURN>
__device__ __forceinline__ void eigh_turn(const c128 *__restrict__ src, double *lamb, c128 *V,
                                          long long idx, bool live, int mode, double (&ev)[M],
                                          c128 *park) {
  constexpr int NRH = M >= 7 ? 4 : M;
  HermP<M> Aw;
  eigh_load_packed<M>(Aw, src);
  c128 W[NRH][M];
#pragma unroll
  for (int r = 0; r < NRH; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) W[r][c] = cmake(TURN * NRH + r == c ? 1.0 : 0.0, 0.0);
  hp_jacobi_rows<M, NRH>(Aw, W);
#pragma unroll
  for (int k = 0; k < M; ++k) ev[k] = Aw.d[k];
  if (mode == 0) {
    if (!live) return;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < M; ++j)
        rank += (ev[j] < ev[k] || (ev[j] == ev[k] && j < k)) ? 1 : 0;
      if (TURN == 0) lamb[idx * M + rank] = ev[k];
#pragma unroll
      for (int r = 0; r < NRH; ++r) {
        const int row = TURN * NRH + r;
        if (row < M) V[(idx * M + row) * M + rank] = W[r][k];
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < NRH; ++r) {
      const int row = TURN * NRH + r;
      if (row < M) {
#pragma unroll
        for (int c = 0; c < M; ++c) park[(row * M + c) * EIGH_LD] = W[r][c];
      }
    }
  }
}

template <int M>
__global__ __launch_bounds__(64) void k_eigh_p(const c128 *__restrict__ A, double *lamb, c128 *V,
                                               long long n, int mode, int floor_kind, double eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  c128 *park = reinterpret_cast<c128 *>(smem) + threadIdx.x;  // mode 1 only
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < n;
  const c128 *src = A + (live ? idx : n - 1) * (M * M);
  double ev[M];
  eigh_turn<M, 0>(src, lamb, V, idx, live, mode, ev, park);
  if constexpr (M >= 7) eigh_turn<M, 1>(src, lamb, V, idx, live, mode, ev, park);
  if (mode == 0 || !live) return;
  // (every lane reads only the rows it parked itself: no barrier)
#pragma unroll
  for (int k = 0; k < M; ++k) ev[k] = apply_floor(ev[k], floor_kind, eps);
#pragma unroll
  for (int a = 0; a < M; ++a) {
    c128 ra[M];
#pragma unroll
    for (int k = 0; k < M; ++k) ra[k] = cscale(park[(a * M + k) * EIGH_LD], ev[k]);
#pragma unroll
    for (int c = a; c < M; ++c) {
      c128 s2 = cmake(0.0, 0.0);
#pragma unroll
      for (int k = 0; k < M; ++k) {  // ra[k] conj(P_ck)
        const c128 pc = park[(c * M + k) * EIGH_LD];
        s2.x = fma(ra[k].x, pc.x, s2.x);
        s2.x = fma(ra[k].y, pc.y, s2.x);
        s2.y = fma(ra[k].y, pc.x, s2.y);
        s2.y = fma(-ra[k].x, pc.y, s2.y);
      }
      if (c == a) {
        V[(idx * M + a) * M + a] = cmake(s2.x, 0.0);
      } else {
        V[(idx * M + a) * M + c] = s2;
        V[(idx * M + c) * M + a] = cconj(s2);
      }
    }
  }
}

// generalised 2x2 Hermitian eigenproblem via Cholesky of B (ref: ssspy/linalg/eigh.py:164-207)
//   type 1: A z = lamb B z   (C = L^-1 A L^-H, z = L^-H y)
//   type 2: A B z = lamb z   (C = L^H A L,    z = L^-H y)
//   type 3: B A z = lamb z   (C = L^H A L,    z = L y)
__global__ __launch_bounds__(256) void k_eigh2(const c128 *__restrict__ A, const c128 *__restrict__ Bm,
                                               double *lamb, c128 *Z, long long n, int type,
                                               int *info) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const c128 a00 = A[idx * 4], a01 = A[idx * 4 + 1], a10 = A[idx * 4 + 2], a11 = A[idx * 4 + 3];
  const double b00 = Bm[idx * 4].x, b11 = Bm[idx * 4 + 3].x;
  const c128 b10 = Bm[idx * 4 + 2];
  // Cholesky B = L L^H (numpy uses the lower triangle)
  const double l00 = sqrt(b00);
  const c128 l10 = cmake(b10.x / l00, b10.y / l00);
  const double d = b11 - cabs2(l10);
  const double l11 = sqrt(d);
  if (!(b00 > 0.0) || !(d > 0.0)) {
    if (info) atomicAdd(info, 1);
  }
  c128 Cm[2][2], P[2][2];
  if (type == 1) {
    // Li = L^-1 = [[1/l00, 0], [-l10/(l00 l11), 1/l11]]
    const double i00 = 1.0 / l00, i11 = 1.0 / l11;
    const c128 i10 = cmake(-l10.x * i00 * i11, -l10.y * i00 * i11);
    // M1 = Li A
    const c128 m00 = cscale(a00, i00), m01 = cscale(a01, i00);
    const c128 m10 = cadd(cmul(i10, a00), cscale(a10, i11));
    const c128 m11 = cadd(cmul(i10, a01), cscale(a11, i11));
    // C = M1 Li^H, Li^H = [[i00, conj(i10)], [0, i11]]
    Cm[0][0] = cscale(m00, i00);
    Cm[0][1] = cadd(cmulc(m00, i10), cscale(m01, i11));
    Cm[1][0] = cscale(m10, i00);
    Cm[1][1] = cadd(cmulc(m10, i10), cscale(m11, i11));
  } else {
    // C = L^H A L, L = [[l00, 0], [l10, l11]], L^H = [[l00, conj(l10)], [0, l11]]
    const c128 m00 = cadd(cscale(a00, l00), cmulc(a10, l10));   // (L^H A) row 0: l00 a0* + conj(l10) a1*
    const c128 m01 = cadd(cscale(a01, l00), cmulc(a11, l10));
    const c128 m10 = cscale(a10, l11), m11 = cscale(a11, l11);
    Cm[0][0] = cadd(cscale(m00, l00), cmul(m01, l10));
    Cm[0][1] = cscale(m01, l11);
    Cm[1][0] = cadd(cscale(m10, l00), cmul(m11, l10));
    Cm[1][1] = cscale(m11, l11);
  }
  // eigenvectors with np.linalg.eigh's phases (lower triangle; eigh2.hpp)
  double ev[2];
  eigh2_lapack(Cm[0][0].x, Cm[1][1].x, Cm[1][0], ev, P);
  lamb[idx * 2] = ev[0];
  lamb[idx * 2 + 1] = ev[1];
  const c128 y0[2] = {P[0][0], P[1][0]}, y1[2] = {P[0][1], P[1][1]};
  c128 z0[2], z1[2];
  if (type == 3) {
    // z = L y
    z0[0] = cscale(y0[0], l00);
    z0[1] = cadd(cmul(l10, y0[0]), cscale(y0[1], l11));
    z1[0] = cscale(y1[0], l00);
    z1[1] = cadd(cmul(l10, y1[0]), cscale(y1[1], l11));
  } else {
    // z = L^-H y, L^-H = [[1/l00, -conj(l10)/(l00 l11)], [0, 1/l11]]
    const double i00 = 1.0 / l00, i11 = 1.0 / l11;
    const c128 i01 = cmake(-l10.x * i00 * i11, l10.y * i00 * i11);
    z0[0] = cadd(cscale(y0[0], i00), cmul(i01, y0[1]));
    z0[1] = cscale(y0[1], i11);
    z1[0] = cadd(cscale(y1[0], i00), cmul(i01, y1[1]));
    z1[1] = cscale(y1[1], i11);
  }
  Z[idx * 4] = z0[0];
  Z[idx * 4 + 1] = z1[0];
  Z[idx * 4 + 2] = z0[1];
  Z[idx * 4 + 3] = z1[1];
}

// plain 2 x 2 Hermitian eigh with np.linalg.eigh's eigenvector phases (the reference's eigh2 without B
// IS np.linalg.eigh, ssspy/linalg/eigh.py:155-157): lower triangle, eigh2_lapack
__global__ __launch_bounds__(256) void k_eigh2_plain(const c128 *__restrict__ A, double *lamb,
                                                     c128 *V, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  double ev[2];
  c128 P[2][2];
  eigh2_lapack(A[idx * 4].x, A[idx * 4 + 3].x, A[idx * 4 + 2], ev, P);
  lamb[idx * 2] = ev[0];
  lamb[idx * 2 + 1] = ev[1];
  V[idx * 4] = P[0][0];
  V[idx * 4 + 1] = P[0][1];
  V[idx * 4 + 2] = P[1][0];
  V[idx * 4 + 3] = P[1][1];
}

// out = P diag(w) P^H (optionally Hermitised), dimension at run time: one thread per entry.
// The second half of to_psd / invsqrtmh when the eigenvalue map is a host callable
// (ref: ssspy/special/psd.py:54-69, ssspy/linalg/sqrtm.py:58-64).
__global__ __launch_bounds__(256) void k_herm_rebuild(const c128 *__restrict__ P,
                                                      const double *__restrict__ w, c128 *out,
                                                      long long n, int M, int hermitise) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * M * M) return;
  const long long idx = e / (M * M);
  const int a = (int)(e % (M * M)) / M, c = (int)(e % M);
  const c128 *Pm = P + idx * M * M;
  const double *wm = w + idx * M;
  c128 ac = cmake(0.0, 0.0), ca = cmake(0.0, 0.0);
  for (int k = 0; k < M; ++k) {
    const c128 pa = cscale(Pm[a * M + k], wm[k]), pc = cscale(Pm[c * M + k], wm[k]);
    ac = cadd(ac, cmulc(pa, Pm[c * M + k]));  // P[a][k] w[k] conj(P[c][k])
    ca = cadd(ca, cmulc(pc, Pm[a * M + k]));
  }
  out[e] = hermitise ? cmake(0.5 * (ac.x + ca.x), 0.5 * (ac.y - ca.y)) : ac;
}

}  // namespace ssspy

using namespace ssspy;

extern "C" {

int ssspy_herm_rebuild(const void *P, const double *w, void *out, long long n, int M, int hermitise,
                       void *stream) {
  SSSPY_REQUIRE(P && w && out && n > 0 && M >= 1, "herm_rebuild: bad argument");
  const long long total = n * M * M;
  hipLaunchKernelGGL(k_herm_rebuild, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     as_stream(stream), (const c128 *)P, w, (c128 *)out, n, M, hermitise);
  return check_launch("k_herm_rebuild");
}

int ssspy_solve(const void *A, const void *Bm, void *X, long long n, int N, int nrhs, int *info,
                void *stream) {
  SSSPY_REQUIRE(A && Bm && X && n > 0 && nrhs > 0, "solve: bad argument");
  if (hermitian_rt_wanted(N)) return solve_rt(A, Bm, X, n, N, nrhs, info, as_stream(stream));
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  DISPATCH_N(N, hipLaunchKernelGGL((k_solve<NN>), grid, block, 0, as_stream(stream),
                                   (const c128 *)A, (const c128 *)Bm, (c128 *)X, n, nrhs, info));
  return check_launch("k_solve");
}

int ssspy_inv2(const void *A, void *out, long long n, void *stream) {
  SSSPY_REQUIRE(A && out && n > 0, "inv2: bad argument");
  hipLaunchKernelGGL(k_inv2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const c128 *)A, (c128 *)out, n);
  return check_launch("k_inv2");
}

// up to 5 x 5: a lane per matrix; 6 x 6: the same on packed storage (k_eigh_p); 7 x 7 and 8 x 8: a
// matrix on 8 lanes (hermitian_rows.hip; the lane-per-matrix instantiations there -- 300-3 300
// spilled VGPRs -- are gone)
#define SSSPY_DISPATCH_EIGH5(M_, CALL)                                                            \
  switch (M_) {                                                                                   \
    case 1: { constexpr int NN = 1; CALL; } break;                                                \
    case 2: { constexpr int NN = 2; CALL; } break;                                                \
    case 3: { constexpr int NN = 3; CALL; } break;                                                \
    case 4: { constexpr int NN = 4; CALL; } break;                                                \
    case 5: { constexpr int NN = 5; CALL; } break;                                                \
    default: return ::ssspy::fail(SSSPY_ERR_UNSUPPORTED, "eigh: size must be in [1, 16]");         \
  }

int ssspy_eigh(const void *A, double *lamb, void *V, long long n, int M, void *stream) {
  SSSPY_REQUIRE(A && lamb && V && n > 0, "eigh: bad argument");
  if (hermitian_rt_wanted(M)) return eigh_rt(A, lamb, V, n, M, 0, 0, 0.0, as_stream(stream));
  if (M == 2) {
    hipLaunchKernelGGL(k_eigh2_plain, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       as_stream(stream), (const c128 *)A, lamb, (c128 *)V, n);
    return check_launch("k_eigh2_plain");
  }
  if (hermitian_rows_wanted(M, 7))
    return eigh_rows(A, lamb, V, n, M, 0, 0, 0.0, as_stream(stream));
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  if (M == 6) {  // packed storage
    hipLaunchKernelGGL((k_eigh_p<6>), grid, block, 0, as_stream(stream), (const c128 *)A, lamb,
                       (c128 *)V, n, 0, 0, 0.0);
    return check_launch("k_eigh_p");
  }
  SSSPY_DISPATCH_EIGH5(M, hipLaunchKernelGGL((k_eigh<NN>), grid, block, 0, as_stream(stream),
                                             (const c128 *)A, lamb, (c128 *)V, n, 0, 0, 0.0));
  return check_launch("k_eigh");
}

int ssspy_to_psd(const void *A, void *out, long long n, int M, int floor_kind, double floor_eps,
                 void *stream) {
  SSSPY_REQUIRE(A && out && n > 0, "to_psd: bad argument");
  if (hermitian_rt_wanted(M))
    return eigh_rt(A, nullptr, out, n, M, 1, floor_kind, floor_eps, as_stream(stream));
  if (hermitian_rows_wanted(M, 7))
    return eigh_rows(A, nullptr, out, n, M, 1, floor_kind, floor_eps, as_stream(stream));
  dim3 grid((unsigned)((n + 63) / 64)), block(64);
  if (M == 6) {
    const size_t smem = (size_t)M * M * EIGH_LD * sizeof(c128);
    if (smem > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)k_eigh_p<6>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return fail(SSSPY_ERR_HIP, hipGetErrorString(e));
    }
    hipLaunchKernelGGL((k_eigh_p<6>), grid, block, smem, as_stream(stream), (const c128 *)A,
                       (double *)nullptr, (c128 *)out, n, 1, floor_kind, floor_eps);
    return check_launch("k_to_psd_p");
  }
  SSSPY_DISPATCH_EIGH5(M, hipLaunchKernelGGL((k_eigh<NN>), grid, block, 0, as_stream(stream),
                                             (const c128 *)A, (double *)nullptr, (c128 *)out, n, 1,
                                             floor_kind, floor_eps));
  return check_launch("k_to_psd");
}

int ssspy_eigh2(const void *A, const void *Bm, double *lamb, void *Z, long long n, int type,
                int *info, void *stream) {
  SSSPY_REQUIRE(A && Bm && lamb && Z && n > 0, "eigh2: bad argument");
  SSSPY_REQUIRE(type >= 1 && type <= 3, "eigh2: type must be 1, 2 or 3");
  hipLaunchKernelGGL(k_eigh2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                     (const c128 *)A, (const c128 *)Bm, lamb, (c128 *)Z, n, type, info);
  return check_launch("k_eigh2");
}

}  // extern "C"
