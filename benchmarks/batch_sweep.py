#!/usr/bin/env python3
"""Mixture-iterations/s of GaussILRMA-IP1 update_once() at the configs[1] shape against the batch
size: where the three kernel families hand over (few-mixture kernels up to 640 work items ~ 9
mixtures, tuned kernels with split tail items, whole rounds).  No B may be slower per mixture than a
smaller one.  GPU box:  python benchmarks/batch_sweep.py [--algo IP|ISS] [--class ilrma|iva|mnmf]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture_batch  # noqa: E402

SIZES = (1, 2, 3, 4, 6, 8, 9, 10, 12, 16, 20, 24, 32, 48, 64, 96, 128)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="*", default=list(SIZES))
    ap.add_argument("--kind", default="ilrma", choices=["ilrma", "iva_ip", "iva_iss", "mnmf"])
    args = ap.parse_args()
    torch.cuda.set_device(0)
    N, F, T, K = 4, 1025, 512, 16
    Xall = torch.from_numpy(nmf_mixture_batch(1000, max(args.sizes), N, F, T)).to("cuda:0")
    rows = []
    prev = 0.0
    for B in args.sizes:
        X = Xall[:B].contiguous()
        if args.kind == "ilrma":
            sep = bench.make_separator(X, K, seed=2000)
            nbytes = 3 * 16.0 * N * F * T
        elif args.kind == "mnmf":
            from ssspy_amd.bss.mnmf import FastGaussMNMF

            sep = FastGaussMNMF(n_basis=8, record_loss=False, rng=np.random.default_rng(0))
            sep._bind_input(X)
            sep._reset()
            nbytes = 4 * 16.0 * N * F * T
        else:
            from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast

            algo = "IP" if args.kind == "iva_ip" else "ISS"
            sep = AuxLaplaceIVA(spatial_algorithm=algo, record_loss=False)
            sep._contrast = _device_contrast(sep.contrast_fn, sep.d_contrast_fn)
            sep._bind_input(X)
            sep._reset()
            if algo == "IP":
                sep._C()
            nbytes = 2 * 16.0 * N * F * T
        iters = max(20, min(400, int(3000 / B)))
        for _ in range(max(5, iters // 10)):
            sep.update_once()
        best = min(bench.time_loop(sep.update_once, iters) for _ in range(3))
        sep._check_device_errors()
        rate = B / best
        rows.append({"B": B, "ms_per_step": round(1e3 * best, 4), "mixture_it_per_s": round(rate, 1),
                     "us_per_mixture": round(1e6 * best / B, 2),
                     "frac": round(nbytes * rate / 8e12, 4),
                     "slower_than_smaller_B": bool(rate < prev * 0.995)})
        prev = max(prev, rate)
        print("B {:4d}  {:8.4f} ms/step  {:9.1f} mixture-it/s  {:7.2f} us/mixture  frac {:.3f}{}".format(
            B, 1e3 * best, rate, 1e6 * best / B, nbytes * rate / 8e12,
            "   <-- slower per mixture than a smaller batch" if rows[-1]["slower_than_smaller_B"] else ""),
            flush=True)
        del sep, X
        torch.cuda.empty_cache()
    print(json.dumps({"kind": args.kind, "shape": [N, F, T], "rows": rows}))


if __name__ == "__main__":
    main()
