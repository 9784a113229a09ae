#!/usr/bin/env python3
"""Wall time of each of the first iterations of a separator (synchronised per iteration): finds
one-off host costs (lazy module loads, workspace growth) hiding in short timing loops."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.mnmf import FastGaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture

X = nmf_mixture(4000, 4, 1025, 512)
m = FastGaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))
m._bind_input(X)
m._reset()
ts = []
for _ in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.update_once()
    torch.cuda.synchronize(); ts.append(1e6 * (time.perf_counter() - t0))
print("per-iteration us:", " ".join("%.0f" % t for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m.update_once()
torch.cuda.synchronize(); print("20 unsynchronised: %.1f us/iter" % (1e6 * (time.perf_counter() - t0) / 20))
t0 = time.perf_counter()
for _ in range(300): m.update_once()
torch.cuda.synchronize(); print("300 unsynchronised: %.1f us/iter" % (1e6 * (time.perf_counter() - t0) / 300))
