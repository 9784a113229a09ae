#!/usr/bin/env python3
"""One GaussILRMA(n_basis=16)(X, n_iter) on configs[1] (record_loss=True, projection back) for a
profiler; prints the wall time of the second call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.ilrma import GaussILRMA  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
X = nmf_mixture(1000, 4, 1025, 512)
GaussILRMA(n_basis=16, rng=np.random.default_rng(0))(X, n_iter=2)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    Y = GaussILRMA(n_basis=16, rng=np.random.default_rng(0))(X, n_iter=n_iter)
    print("__call__ %d iterations: %.2f ms" % (n_iter, 1e3 * (time.perf_counter() - t0)))
