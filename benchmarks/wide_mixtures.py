#!/usr/bin/env python3
"""GaussILRMA-IP update_once() beyond 4 sources (grouped tuned NMF passes + generic covariance):

    python benchmarks/wide_mixtures.py [--batch 16] [--iters 10]

One JSON object per source count, with the time per iteration and the time per (source, bin, frame)
relative to nothing -- compare the ns_per_point column across N (N = 4 is the tuned path).
"""
import gc
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ssspy_amd.bss.ilrma import GaussILRMA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--sources", default="4,5,6,7,8")
    args = ap.parse_args()
    F, T, K = 1025, 512, 16
    for N in [int(v) for v in args.sources.split(",")]:
        g = torch.Generator(device="cuda").manual_seed(N)
        X = torch.randn(args.batch, N, F, T, dtype=torch.complex128, device="cuda", generator=g)
        sep = GaussILRMA(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
        sep._bind_input(X)
        sep._reset(flooring_fn=sep.flooring_fn)
        for _ in range(2):
            sep.update_once()
        torch.cuda.synchronize()
        gc.collect(); gc.freeze()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            sep.update_once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        print(json.dumps({"n_sources": N, "batch": args.batch, "ms_per_iter": round(1e3 * dt, 3),
                          "ns_per_point": round(1e9 * dt / (args.batch * N * F * T), 4)}))
        del sep, X
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
