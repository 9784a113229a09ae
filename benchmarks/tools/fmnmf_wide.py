#!/usr/bin/env python3
"""FastGaussMNMF above 4 channels for a profiler: `python benchmarks/tools/fmnmf_wide.py <channels>
[batch] [F] [T]`: 10 iterations, then 5 Wiener filters (`separate`), n_basis = 8."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.mnmf import FastGaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

M = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
F = int(sys.argv[3]) if len(sys.argv) > 3 else 513
T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
X = torch.from_numpy(nmf_mixture(7, M, F, T)).cuda()[None].expand(B, -1, -1, -1).contiguous()
m = FastGaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))
m._bind_input(X)
m._reset()
for _ in range(2):
    m.update_once()
torch.cuda.synchronize()


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("FastGaussMNMF M=%d B=%d F=%d T=%d: %.3f ms per iteration, separate %.3f ms" % (
    M, B, F, T, timed(m.update_once, 10), timed(lambda: m._wiener(m._X), 5)))
