// Lane mapping of gfx950's v_permlane32_swap / v_permlane16_swap as the fused ISS kernel uses them
// (iss_fused.hip: fold_halves / fold_row_pairs).  Prints, for x[l] = l and y[l] = 100 + l, what
// every 16-lane row holds after each swap, and checks the reduce-scatter of 8 values against a
// plain sum.      hipcc --offload-arch=gfx950 -O2 permlane_swap.hip -o bin/permlane_swap
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k(int *o32, int *o16, double *rs) {
  const int l = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane32_swap((unsigned)l, 100u + l, false, false);
  o32[l] = (int)a[0];
  o32[64 + l] = (int)a[1];
  const auto b = __builtin_amdgcn_permlane16_swap((unsigned)l, 100u + l, false, false);
  o16[l] = (int)b[0];
  o16[64 + l] = (int)b[1];
  // reduce-scatter of v[k] = (k + 1) * (l + 1), k < 8: wave totals are (k + 1) * 2080
  double v[8];
  for (int k2 = 0; k2 < 8; ++k2) v[k2] = (double)((k2 + 1) * (l + 1));
  for (int p = 0; p < 4; ++p) {
    const double x = v[2 * p], y = v[2 * p + 1];
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    v[p] = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  }
  for (int u = 0; u < 2; ++u) {
    const double x = v[2 * u], y = v[2 * u + 1];
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    v[u] = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  }
  for (int u = 0; u < 2; ++u) {
    double s = v[u];
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
    rs[u * 64 + l] = s;
  }
}

int main() {
  int *o32, *o16;
  double *rs;
  (void)hipMallocManaged(&o32, 128 * sizeof(int));
  (void)hipMallocManaged(&o16, 128 * sizeof(int));
  (void)hipMallocManaged(&rs, 128 * sizeof(double));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o32, o16, rs);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < 4; ++r)
    printf("permlane32_swap row %d: first -> %d.., second -> %d..\n", r, o32[16 * r], o32[64 + 16 * r]);
  for (int r = 0; r < 4; ++r)
    printf("permlane16_swap row %d: first -> %d.., second -> %d..\n", r, o16[16 * r], o16[64 + 16 * r]);
  int bad = 0;
  for (int r = 0; r < 4; ++r) {
    const int slot = ((r & 1) << 1) | (r >> 1);
    for (int u = 0; u < 2; ++u) {
      const double want = (4 * u + slot + 1) * 2080.0;
      if (rs[u * 64 + 16 * r] != want) ++bad;
      printf("row %d v[%d] = %.0f (value %d, want %.0f)\n", r, u, rs[u * 64 + 16 * r], 4 * u + slot, want);
    }
  }
  printf(bad ? "REDUCE-SCATTER MISMATCH\n" : "reduce-scatter mapping OK\n");
  return bad;
}
