#!/usr/bin/env python3
"""Host <-> device copies of a pageable NumPy array: the plain torch copy against a pipeline through
two page-locked staging buffers (host memcpy of chunk i + 1 under the DMA of chunk i)."""
import sys
import time

import numpy as np
import torch

dev = torch.device("cuda")


def pipe_up(a, stage, chunk):
    src = torch.from_numpy(a.reshape(-1).view(np.uint8))
    dst = torch.empty(src.shape, dtype=torch.uint8, device=dev)
    ev = [None, None]
    n = src.numel()
    for k, off in enumerate(range(0, n, chunk)):
        s = stage[k & 1]
        if ev[k & 1] is not None:
            ev[k & 1].synchronize()
        m = min(chunk, n - off)
        s[:m].copy_(src[off:off + m])
        dst[off:off + m].copy_(s[:m], non_blocking=True)
        e = torch.cuda.Event()
        e.record()
        ev[k & 1] = e
    torch.cuda.synchronize()
    return dst


def pipe_down(t, out, stage, chunk):
    src = t.reshape(-1).view(torch.uint8)
    dst = torch.from_numpy(out.reshape(-1).view(np.uint8))
    n = src.numel()
    offs = list(range(0, n, chunk))
    ev = []
    for k, off in enumerate(offs[:2]):
        m = min(chunk, n - off)
        stage[k & 1][:m].copy_(src[off:off + m], non_blocking=True)
        e = torch.cuda.Event(); e.record(); ev.append(e)
    for k, off in enumerate(offs):
        m = min(chunk, n - off)
        ev[k].synchronize()
        dst[off:off + m].copy_(stage[k & 1][:m])
        if k + 2 < len(offs):
            o2 = offs[k + 2]
            m2 = min(chunk, n - o2)
            stage[k & 1][:m2].copy_(src[o2:o2 + m2], non_blocking=True)
            e = torch.cuda.Event(); e.record(); ev.append(e)
    return out


def huge_empty(nbytes):
    """uint8 array on a fresh anonymous mapping that asked for transparent huge pages"""
    import mmap
    size = (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1)
    mm = mmap.mmap(-1, size)
    mm.madvise(mmap.MADV_HUGEPAGE)
    return np.frombuffer(mm, dtype=np.uint8, count=nbytes)


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


print("torch threads", torch.get_num_threads())
for mb in (34, 269):
    a = np.random.default_rng(0).standard_normal(mb * 1000 * 1000 // 8)
    t = torch.from_numpy(a).to(dev)
    plain_up = best(lambda: torch.from_numpy(a).to(dev, copy=True))
    plain_dn = best(lambda: t.cpu().numpy())
    fresh_up = best(lambda: torch.from_numpy(a.copy()).to(dev, copy=True))
    host_copy = best(lambda: a.copy())
    print("    (a.copy() alone %.2f ms; copy + upload %.2f ms)" % (1e3 * host_copy, 1e3 * fresh_up))
    line = "%3d MB: plain up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s)" % (
        mb, 1e3 * plain_up, a.nbytes / plain_up / 1e9, 1e3 * plain_dn, a.nbytes / plain_dn / 1e9)
    print(line)
    for cmb in (4, 8, 16, 32):
        chunk = cmb << 20
        stage = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        up = best(lambda: pipe_up(a, stage, chunk))
        out = np.empty_like(a)
        dn = best(lambda: pipe_down(t, out, stage, chunk))
        assert np.array_equal(out, a) and torch.equal(pipe_up(a, stage, chunk).view(torch.float64), t)
        dn_fresh = best(lambda: pipe_down(t, np.empty_like(a), stage, chunk))
        dn_huge = best(lambda: pipe_down(t, huge_empty(a.nbytes), stage, chunk))
        print("      down into a fresh huge-page mapping %.2f ms" % (1e3 * dn_huge))
        print("   chunk %2d MB: up %.2f ms (%.1f GB/s), down %.2f ms (%.1f GB/s), down into a fresh "
              "array %.2f ms" % (cmb, 1e3 * up, a.nbytes / up / 1e9, 1e3 * dn, a.nbytes / dn / 1e9,
                                 1e3 * dn_fresh))
