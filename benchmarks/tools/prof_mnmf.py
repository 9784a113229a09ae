#!/usr/bin/env python3
"""Host-side cost of FastGaussMNMF.update_once() (one mixture): time per call while the launch queue
is short, and the split between the Python wrapper and the C-ABI call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd import _ops
from ssspy_amd.bss.mnmf import FastGaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture
X = nmf_mixture(4000, 4, 1025, 512)
m = FastGaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))
m._bind_input(torch.from_numpy(X[None]).cuda()); m._reset()
for _ in range(10): m.update_once()
torch.cuda.synchronize()
for n in (5, 20, 100, 300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.update_once()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("n=%3d: host issue %.1f us/iter, with sync %.1f us/iter" % (n, 1e6*(t1-t0)/n, 1e6*(t2-t0)/n))
orig = _ops.fastmnmf_update_handover
acc = [0.0, 0]
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); acc[0] += time.perf_counter() - t; acc[1] += 1; return r
_ops.fastmnmf_update_handover = timed
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): m.update_once()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("C-ABI call: %.1f us of %.1f us per update_once (%d calls)" % (1e6*acc[0]/50, 1e6*(t1-t0)/50, acc[1]))
