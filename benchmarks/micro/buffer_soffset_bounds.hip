// Is the scalar offset of a raw buffer load part of the hardware range check on gfx950?
//   hipcc --offload-arch=gfx950 -O2 benchmarks/micro/buffer_soffset_bounds.hip -o build/micro/soffset && build/micro/soffset
// A 1 KiB descriptor inside a 64 KiB allocation of known words: a load at voffset 0 + soffset 2048
// returns the word at byte 2048 if soffset is excluded from the check, 0 if it is included.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned *base, unsigned *out) {
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(base), 0, 1024, 0x00020000);
  const unsigned lane = threadIdx.x;
  // 0: in range; 1: voffset beyond num_records; 2: soffset beyond; 3: voffset in range + soffset
  // pushing the sum beyond; 4: immediate-free straddle of the last 16 bytes
  out[0 * 64 + lane] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u, 0, 0)[0];
  out[1 * 64 + lane] = __builtin_amdgcn_raw_buffer_load_b128(r, 2048u + lane * 16u, 0, 0)[0];
  out[2 * 64 + lane] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16u, 2048, 0)[0];
  out[3 * 64 + lane] = __builtin_amdgcn_raw_buffer_load_b128(r, 512u + lane * 16u, 768, 0)[0];
  out[4 * 64 + lane] = __builtin_amdgcn_raw_buffer_load_b128(r, 1016u, 0, 0)[0];
}

int main() {
  const int n = 16384;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0x10000000u + 4u * i;  // the byte address of each word
  unsigned *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 5 * 64 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  std::vector<unsigned> r(5 * 64);
  hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
  const char *what[5] = {"in range", "voffset 2048 (beyond 1024)", "soffset 2048 (beyond 1024)",
                         "voffset 512 + soffset 768", "voffset 1016 (straddles the end)"};
  for (int t = 0; t < 5; ++t)
    printf("%-34s lane0 %08x lane40 %08x\n", what[t], r[t * 64], r[t * 64 + 40]);
  return 0;
}
