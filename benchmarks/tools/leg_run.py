#!/usr/bin/env python3
"""One separator leg for a profiler: `python benchmarks/tools/leg_run.py <leg> [batch] [iters]` with
leg in ilrma_{ip1,ip2,iss1,iss2,ipa}, auxiva_{ip1,ip2,iss1,iss2,ipa}, fmnmf_{ip1,ip2} at the
configs[1] / configs[3] shape (N = 4, F = 1025, T = 512)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

leg = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
N = int(os.environ.get("LEG_SOURCES", "4"))
X = torch.from_numpy(nmf_mixture(1000, N, 1025, 512)).cuda()[None].expand(B, -1, -1, -1).contiguous()
family, algo = leg.split("_")
if family == "ilrma":
    from ssspy_amd.bss.ilrma import GaussILRMA
    m = GaussILRMA(n_basis=16, spatial_algorithm=algo.upper(), record_loss=False,
                   rng=np.random.default_rng(2000))
elif family == "auxiva":
    from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast
    m = AuxLaplaceIVA(spatial_algorithm=algo.upper(), record_loss=False)
    m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
else:
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    m = FastGaussMNMF(n_basis=8, diagonalizer_algorithm=algo.upper(), record_loss=False,
                      rng=np.random.default_rng(0))
m._bind_input(X)
m._reset()
for _ in range(2):
    m.update_once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    m.update_once()
e1.record()
torch.cuda.synchronize()
print("%s N=%d B=%d: %.3f ms per iteration (under rocprofv3 when run by profile_round.sh)" % (leg, N, B, e0.elapsed_time(e1) / iters))
