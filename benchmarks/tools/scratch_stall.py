#!/usr/bin/env python3
"""Is the slow-down of a GaussMNMF of 8 channels that follows one of 4 channels in the same process a
per-dispatch cost or one stall?  Per-iteration wall times (synchronised) of 30 iterations."""
import os, sys, time
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from ssspy_amd.bss.mnmf import GaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture

F, T, K, B = 513, 256, 8, 8
for M in [int(a) for a in sys.argv[1:]] or [4, 8]:
    X = np.stack([nmf_mixture(4000 + b, M, F, T) for b in range(B)])
    m = GaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
    m._bind_input(X); m._reset()
    times = []
    for _ in range(int(os.environ.get('SYNC_ITERS', '30'))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.update_once()
        torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
    print("M=%d ms per iteration:" % M, " ".join("%.1f" % t for t in times), flush=True)
    for n in (6, 30, 60):  # the same without a synchronisation per iteration
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            m.update_once()
        torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
        print("   %2d iterations queued back to back: %.1f ms in all, %.2f ms each" % (n, dt, dt / n), flush=True)
