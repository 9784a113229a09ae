#!/usr/bin/env python3
"""ms per iteration of every separator family / spatial algorithm at a given source count (a survey
for pathological paths): python benchmarks/tools/leg_survey.py <n_sources> [batch] [F] [T] [substring of
the leg names to run]"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.ilrma import GGDILRMA, TILRMA, GaussILRMA  # noqa: E402
from ssspy_amd.bss.iva import AuxGaussIVA, AuxLaplaceIVA  # noqa: E402
from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

warnings.simplefilter("ignore")
N = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
F = int(sys.argv[3]) if len(sys.argv) > 3 else 513
T = int(sys.argv[4]) if len(sys.argv) > 4 else 256
X = torch.from_numpy(nmf_mixture(7, N, F, T)).cuda()[None].expand(B, -1, -1, -1).contiguous()
legs = []
for algo in ("IP1", "IP2", "ISS1", "ISS2", "IPA"):
    legs.append(("GaussILRMA-" + algo, lambda a=algo: GaussILRMA(n_basis=8, spatial_algorithm=a, record_loss=False, rng=np.random.default_rng(0))))
    legs.append(("AuxLaplaceIVA-" + algo, lambda a=algo: AuxLaplaceIVA(spatial_algorithm=a, record_loss=False)))
for algo in ("IP1", "ISS2"):
    legs.append(("TILRMA-" + algo, lambda a=algo: TILRMA(n_basis=8, dof=4.0, spatial_algorithm=a, record_loss=False, rng=np.random.default_rng(0))))
    legs.append(("GGDILRMA-" + algo, lambda a=algo: GGDILRMA(n_basis=8, beta=1.0, spatial_algorithm=a, record_loss=False, rng=np.random.default_rng(0))))
legs.append(("AuxGaussIVA-IP2", lambda: AuxGaussIVA(spatial_algorithm="IP2", record_loss=False)))
if N <= 8:
    for algo in ("IP1", "IP2"):
        legs.append(("FastGaussMNMF-" + algo, lambda a=algo: FastGaussMNMF(n_basis=8, diagonalizer_algorithm=a, record_loss=False, rng=np.random.default_rng(0))))
    legs.append(("GaussMNMF", lambda: GaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))))
pts = N * F * T * B
if len(sys.argv) > 5:
    legs = [leg for leg in legs if sys.argv[5] in leg[0]]
for name, make in legs:
    try:
        m = make()
        m._bind_input(X)
        m._reset()
        m.update_once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < 0.3 and n < 50):
            m.update_once()
            n += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        extra = ""
        if hasattr(m, "_separate_dev"):
            torch.cuda.synchronize(); t1 = time.perf_counter(); m._separate_dev(); torch.cuda.synchronize()
            extra = "  separate %.3f ms" % (1e3 * (time.perf_counter() - t1))
        print("N=%d B=%d %-22s %8.3f ms/iter  %7.4f ns/point%s" % (N, B, name, 1e3 * dt, 1e9 * dt / pts, extra), flush=True)
        del m
    except Exception as exc:
        print("N=%d B=%d %-22s %s: %s" % (N, B, name, type(exc).__name__, str(exc)[:80]), flush=True)
    torch.cuda.empty_cache()
