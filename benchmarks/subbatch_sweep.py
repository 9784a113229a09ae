#!/usr/bin/env python3
"""The headline batch (128 mixtures of configs[1], GaussILRMA-IP1) iterated in cache-sized sub-batches
on one or several streams (SSSPY_AMD_SUBBATCH, bss/ilrma.py::_subbatch_plan) against the whole-batch
launches: mixture-iterations/s, socket power, shader clock (round-4 verdict item 6b).

    python benchmarks/subbatch_sweep.py [--batch 128] [--seconds 1.5]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

_spec = importlib.util.spec_from_file_location("power_profile",
                                               os.path.join(ROOT, "benchmarks", "power_profile.py"))
_pp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_pp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--plans", default="0,2:1,4:1,6:1,8:1,2:2,3:2,4:2,2:3,2:4,3:3,8:2,16:1,16:2,32:2,64:2")
    args = ap.parse_args()
    from ssspy_amd.utils.dataset import nmf_mixture_batch

    B, N, F, T, K = args.batch, 4, 1025, 512, 16
    X = torch.from_numpy(nmf_mixture_batch(1000, B, N, F, T)).cuda()
    rows = []
    for plan in args.plans.split(","):
        if plan == "0":
            os.environ.pop("SSSPY_AMD_SUBBATCH", None)
        else:
            os.environ["SSSPY_AMD_SUBBATCH"] = plan
        sep = bench.make_separator(X, K, seed=2000)
        for _ in range(5):
            sep.update_once()
        torch.cuda.synchronize()
        sampler = _pp.Sampler()
        n = 0
        with sampler:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(10):
                    sep.update_once()
                n += 10
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        sep._check_device_errors()
        pw = sampler.summary(0.5)
        row = {"plan": plan, "ms_per_step": round(1e3 * dt / n, 4),
               "mixture_it_per_s": round(B * n / dt, 1), "socket_w": pw.get("socket_w"),
               "sclk_mhz": pw.get("sclk_mhz")}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del sep
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
