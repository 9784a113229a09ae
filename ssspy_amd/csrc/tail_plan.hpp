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

// Pick the number of chunks `s` (1 .. max_split, no empty chunks) an item of `ntiles` tiles is cut
// into when `items` of them are scheduled on `slots` concurrent workgroups: equal-size blocks run in
// ceil(items * s / slots) rounds of ceil(ntiles / s) tiles each plus a fixed cost per block (its
// prologue and the partial-sum record it leaves, ~2 tiles) and per chunk of the fold.  Round 4
// (benchmarks/batch_sweep.py): the old rule -- the largest s that still fits ONE round, none if two
// items do not fit -- left 47 % of the slots idle at 16 mixtures of the configs[1] shape (272 items,
// s = 1) where three chunks in two short rounds take 25 tile times instead of 33.5.
static inline int best_split(long long items, int ntiles, int slots, int max_split,
                             long long max_blocks) {
  if (items <= 0 || ntiles <= 1) return 1;
  const double ovh = 2.0, fold = 0.15;
  int best = 1;
  double best_cost = (double)((items + slots - 1) / slots) * (ntiles + ovh);
  if (max_split > ntiles) max_split = ntiles;
  for (int s = 2; s <= max_split; ++s) {
    const int tpc = (ntiles + s - 1) / s;
    const int se = (ntiles + tpc - 1) / tpc;  // chunks that are not empty
    if (se != s) continue;                    // the same cut as a smaller s
    const long long blocks = items * se;
    if (blocks > max_blocks) break;
    const double cost = (double)((blocks + slots - 1) / slots) * (tpc + ovh) + fold * se;
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = se;
    }
  }
  return best;
}

// slots: workgroups the chip holds at once for the kernel (512 at two workgroups per CU, 256 for
// a kernel whose register use allows one); max_blocks: split blocks the caller's partial-sum
// scratch has room for (default: one round, the pre-round-4 behaviour)
static inline TailPlan make_tail_plan(int B, int groups, int ntiles, int slots = SLOTS,
                                      int max_blocks = 0) {
  TailPlan p;
  const long long items = (long long)B * groups;
  p.groups = groups;
  p.full = (int)(items / slots) * slots;
  p.tail = (int)(items - p.full);
  p.split = p.tail > 0 ? best_split(p.tail, ntiles, slots, 16, max_blocks > 0 ? max_blocks : slots) : 1;
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
