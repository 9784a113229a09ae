#!/usr/bin/env python3
"""GaussMNMF on the configs[3] shape under rocprofv3 --kernel-trace --stats (see profiles/)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd.bss.mnmf import GaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M, F, T, K = 4, 1025, 512, 8
X = np.stack([nmf_mixture(4000, M, F, T)] * B)
m = GaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
m._bind_input(X)
m._reset()
for _ in range(5):
    m.update_once()
torch.cuda.synchronize()
