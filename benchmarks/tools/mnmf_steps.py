#!/usr/bin/env python3
"""Per-step device time of FastGaussMNMF at the configs[3] shape (HIP events around 20 calls of one
step of the C ABI each): basis, activation, diagonaliser (covariance pass + IP1), spatial.

    [SSSPY_AMD_LIB=<variant .so>] python benchmarks/tools/mnmf_steps.py [batch] [n_basis]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd import _lib  # noqa: E402
from ssspy_amd.bss.mnmf import FastGaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X = nmf_mixture(4000, 4, 1025, 512)
Xd = torch.from_numpy(X).cuda()[None].expand(B, -1, -1, -1).contiguous()
m = FastGaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
m._bind_input(Xd)
m._reset()
for _ in range(3):
    m.update_once()
steps = [("basis", _lib.MNMF_BASIS), ("activation", _lib.MNMF_ACTIVATION),
         ("diagonalizer", _lib.MNMF_DIAGONALIZER), ("spatial", _lib.MNMF_SPATIAL), ("all", _lib.MNMF_ALL)]
out = []
for name, flag in steps:
    m.update_once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        m._update(flag)
    e1.record()
    torch.cuda.synchronize()
    out.append("%s %.1f" % (name, e0.elapsed_time(e1) / 20 * 1e3))
print(os.path.basename(os.environ.get("SSSPY_AMD_LIB", "default")), "B=%d K=%d us:" % (B, K), "  ".join(out))
