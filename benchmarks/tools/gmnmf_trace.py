#!/usr/bin/env python3
"""GaussMNMF update_once() x iters on 8 mixtures (F = 513, T = 256, K = 8) for rocprofv3:
    rocprofv3 --kernel-trace --stats ... -- python benchmarks/tools/gmnmf_trace.py <channels> [iters]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.mnmf import GaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture
M = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = 8; F, T, K = 513, 256, 8
X = np.stack([nmf_mixture(4000 + b, M, F, T) for b in range(B)])
m = GaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
m._bind_input(X); m._reset()
for _ in range(iters): m.update_once()
torch.cuda.synchronize()
