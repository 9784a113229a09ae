#!/usr/bin/env python3
"""GaussMNMF per-iteration time against the number of channels (M = N), F = 513, T = 256, K = 8,
`batch` mixtures; and the states after 3 iterations of the packed per-point kernels against the
full-storage ones (rounds 4-5; the switch went in round 6, the column is in profiles/r04_gmnmf_channels.txt).
GM_WARM=<n>: iterations before the states are compared and the 6 timed iterations start (default 3).

    python benchmarks/gmnmf_channels.py [batch] [M ...]
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd.bss.mnmf import GaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

F, T, K = 513, 256, 8


def run(B, M, iters=6):
    X = np.stack([nmf_mixture(4000 + b, M, F, T) for b in range(B)])
    m = GaussMNMF(n_basis=K, record_loss=True, rng=np.random.default_rng(0))
    m._bind_input(X)
    m._reset()
    for _ in range(int(os.environ.get("GM_WARM", "3"))):
        m.update_once()
    loss3 = np.asarray(m.compute_loss())
    state = {k: np.asarray(getattr(m, k)).copy() for k in ("basis", "activation", "spatial")}
    state["loss"] = loss3
    state["output"] = np.asarray(m.separate(m.input))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m.update_once()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters, state


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Ms = [int(a) for a in sys.argv[2:]] or [2, 3, 4, 5, 6, 7, 8]
    warm = os.environ.get("GM_WARM", "3")
    base = None
    for M in Ms:
        ms, state = run(B, M)
        if M == 4:
            base = ms
        print(json.dumps({"channels": M, "batch": B, "timed_after_iterations": int(warm),
                          "ms_per_iter": round(ms, 3),
                          "vs_4_channels": round(ms / base, 2) if base else None}), flush=True)
