#!/usr/bin/env python3
"""Per-mixture kappa_rms of the implied-filter route over the first iterations of the bench batch
(seeds 1000..1000+B-1, the configs[1] shape), guard off: how many mixtures pass a given limit when."""
import os, sys, warnings
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.bss.iva import AuxLaplaceIVA
from ssspy_amd.utils.dataset import nmf_mixture_batch
warnings.simplefilter("ignore")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 12
X = torch.from_numpy(nmf_mixture_batch(1000, B, 4, 1025, 512)).to("cuda")
for name, make in [("ilrma ISS2", lambda: GaussILRMA(n_basis=16, spatial_algorithm="ISS2", record_loss=False, rng=np.random.default_rng(2000))),
                   ("ilrma IPA", lambda: GaussILRMA(n_basis=16, spatial_algorithm="IPA", record_loss=False, rng=np.random.default_rng(2000))),
                   ("iva ISS2", lambda: AuxLaplaceIVA(spatial_algorithm="ISS2", record_loss=False))]:
    m = make()
    m._implied_amp_limit = float("inf")
    m._amp_every_launch = True
    m._bind_input(X)
    m._reset()
    print(name)
    for it in range(n_iter):
        m.update_once()
        amp = m.__dict__.get("_amp")
        h = amp["dev"][(amp["phase"] - 1) & 1].cpu().numpy()
        k = np.sqrt(h[:, 0] / h[:, 1])
        print("  it %2d  median %.1e  p90 %.1e  max %.1e  >1e5: %d  >1e6: %d  >1e7: %d" % (
            it + 1, np.median(k), np.quantile(k, 0.9), k.max(), (k > 1e5).sum(), (k > 1e6).sum(), (k > 1e7).sum()), flush=True)
