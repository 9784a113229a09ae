#!/usr/bin/env python3
"""configs[1] literally: ONE GaussILRMA-IP1 mixture (N=4, F=1025, T=512, K=16), update_once loop."""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.utils.dataset import nmf_mixture

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
X = nmf_mixture(1000, 4, 1025, 512)
sep = GaussILRMA(n_basis=16, record_loss=False, rng=np.random.default_rng(0))
sep._bind_input(X); sep._reset(flooring_fn=sep.flooring_fn); sep._C()
for _ in range(10): sep.update_once()
gc.collect(); gc.freeze()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): sep.update_once()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("single mixture: %.1f us/iter, %.0f it/s" % (1e6 * dt / n, n / dt))
