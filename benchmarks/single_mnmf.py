#!/usr/bin/env python3
"""configs[3] literally: ONE FastGaussMNMF mixture (N=M=4, F=1025, T=512, K=8), update_once loop."""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd.bss.mnmf import FastGaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
X = nmf_mixture(4000, 4, 1025, 512)
m = FastGaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))
if "--numpy" in sys.argv:
    m._bind_input(X)                                   # the reference contract: a NumPy array
else:
    m._bind_input(torch.from_numpy(X[None]).cuda())    # resident input, as bench.py binds it
m._reset()
for _ in range(10): m.update_once()
gc.collect(); gc.freeze()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): m.update_once()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("single FastGaussMNMF mixture: %.1f us/iter, %.0f it/s" % (1e6 * dt / n, n / dt))
