// Precision of v_rcp_f64 and of one / two Newton steps on it (max relative error over 2^24 inputs).
// Build: hipcc --offload-arch=gfx950 -O3 rcp_precision.hip -o rcp_precision
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double *x, double *e0, double *e1, double *e2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i], exact = 1.0 / v;
  double r = __builtin_amdgcn_rcp(v);
  e0[i] = fabs(r - exact) / exact;
  double e = fma(-v, r, 1.0);
  r = fma(r, e, r);
  e1[i] = fabs(r - exact) / exact;
  e = fma(-v, r, 1.0);
  r = fma(r, e, r);
  e2[i] = fabs(r - exact) / exact;
}

int main() {
  const int n = 1 << 24;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    h[i] = ldexp(1.0 + u, (int)(s % 120) - 60);
  }
  double *x, *e0, *e1, *e2;
  hipMalloc(&x, n * 8); hipMalloc(&e0, n * 8); hipMalloc(&e1, n * 8); hipMalloc(&e2, n * 8);
  hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
  std::vector<double> a(n), b(n), c(n);
  hipMemcpy(a.data(), e0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), e1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), e2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) { m0 = fmax(m0, a[i]); m1 = fmax(m1, b[i]); m2 = fmax(m2, c[i]); }
  printf("max rel err: v_rcp_f64 %.3e, +1 Newton %.3e, +2 Newton %.3e (eps = 2.2e-16)\n", m0, m1, m2);
  return 0;
}
