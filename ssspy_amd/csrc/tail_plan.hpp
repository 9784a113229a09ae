// Two-level schedule of the bin-major pass kernels (ILRMA and FastMNMF).
#pragma once

#include "common.hpp"

namespace ssspy {

// Two-level schedule of the bin-major kernels.  A work item is (mixture, bin group) walking all
// frame tiles; the chip holds SLOTS workgroups at once (2 per CU).  Items that fill whole rounds of
// SLOTS run unsplit and finish their bins in place; the remaining `tail` items are split into
// `split` frame chunks each, so the last round is 1/split as long instead of leaving most CUs idle
// (128 mixtures x 17 groups = 2176 items = 4.25 rounds: 5 rounds unsplit, 4.25 split).  Split blocks
// write partial sums to a scratch area indexed [tail item][chunk]; a small second kernel folds them.
// Small batches are the same formula with full == 0.
constexpr int SLOTS = 512;
struct TailPlan {
  int full, tail, split, groups;  // blocks = full + tail * split; groups = bin groups per mixture
};

// slots: workgroups the chip holds at once for the kernel (512 at two workgroups per CU, 256 for
// a kernel whose register use allows one)
static inline TailPlan make_tail_plan(int B, int groups, int ntiles, int slots = SLOTS) {
  TailPlan p;
  const long long items = (long long)B * groups;
  p.groups = groups;
  p.full = (int)(items / slots) * slots;
  p.tail = (int)(items - p.full);
  p.split = 1;
  if (p.tail > 0) {
    int s = slots / p.tail;
    s = s > 16 ? 16 : s;
    s = s > ntiles ? ntiles : s;
    if (s > 1) {
      const int tpc = (ntiles + s - 1) / s;
      s = (ntiles + tpc - 1) / tpc;  // no empty chunks
    }
    p.split = s < 1 ? 1 : s;
  }
  if (p.split == 1) {
    p.full += p.tail;
    p.tail = 0;
  }
  return p;
}

struct BlockWork {
  int b, group, chunk, nchunks, tail_idx;
};
__device__ __forceinline__ BlockWork block_work(const TailPlan &p) {
  BlockWork w;
  int item = blockIdx.x;
  if (item < p.full) item = xcd_contiguous(item, p.full);
  w.chunk = 0;
  w.nchunks = 1;
  w.tail_idx = 0;
  if (item >= p.full) {
    const int t = item - p.full;
    w.tail_idx = t / p.split;
    w.chunk = t - w.tail_idx * p.split;
    w.nchunks = p.split;
    item = p.full + w.tail_idx;
  }
  w.b = item / p.groups;
  w.group = item - w.b * p.groups;
  return w;
}

}  // namespace ssspy
