#!/usr/bin/env python3
"""kappa_rms of every launch of the implied-filter route (guard off) on the draws of
tests/test_gpu_parity.py::test_implied_filter_route_is_left_past_its_rounding_bound."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.bss.iva import AuxLaplaceIVA
from ssspy_amd.utils.dataset import nmf_mixture
warnings.simplefilter("ignore")
for family, algo, N, F, T, seed, n_iter in [("ilrma", "ISS2", 3, 33, 8, 0, 12), ("ilrma", "ISS2", 3, 33, 8, 3, 12),
                                    ("ilrma", "ISS2", 4, 31, 11, 6, 12), ("ilrma", "IPA", 4, 31, 11, 4, 12),
                                    ("ilrma", "ISS2", 4, 1025, 512, 1000, 100), ("ilrma", "IPA", 4, 1025, 512, 1000, 100)]:
    K = 8 if F < 100 else 16
    X = nmf_mixture((7000 + seed) if seed < 1000 else seed, N, F, T)
    rng = np.random.default_rng(seed)
    kw = dict(basis=rng.random((N, F, K)), activation=rng.random((N, K, T)))
    m = GaussILRMA(n_basis=K, spatial_algorithm=algo, record_loss=False)
    m._implied_amp_limit = float("inf")
    m._amp_every_launch = True
    trace = []
    class M(type(m)):
        pass
    def look(mm):
        amp = mm.__dict__.get("_amp")
        if amp and amp["phase"]:
            h = amp["dev"][(amp["phase"] - 1) & 1].cpu().numpy()
            trace.append(float(np.sqrt(h[0, 0] / h[0, 1])))
    m.callbacks = [look]
    m(X, n_iter=n_iter, **kw)
    print(family, algo, N, F, T, seed, " ".join("%.0e" % t for t in trace), flush=True)
