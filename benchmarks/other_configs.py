#!/usr/bin/env python3
"""Timings of the BASELINE.json configs that are not the bench line (parity-test cases):
configs[2] AuxIVA-ISS (N=8, F=2049, T=1024) and configs[3] FastGaussMNMF (N=M=4, F=1025, T=512, K=8).

    python benchmarks/other_configs.py [--iters 20]

Prints one JSON object per config: iterations/s of update_once() (record_loss=False), the
algorithmic-bytes rate of SURVEY.md 8d, and the one-off costs outside the loop.
"""

import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402


def timed(fn, n):
    gc.collect()   # a full collection (42 ms with torch imported) must not land in a 20-iteration loop
    gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--only", default="", help="comma-separated subset of iva_iss,iva_ip,iva_iss4,ilrma_iss,ilrma_models,fastmnmf,gmnmf")
    args = ap.parse_args()
    only = [t for t in args.only.split(",") if t]

    def want(tag):
        return not only or tag in only

    B = args.batch
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF

    if want("iva_iss"):
        # configs[2]: AuxIVA-ISS, 8 sources
        N, F, T = 8, 2049, 1024
        X = nmf_mixture(3000, N, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        m = AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=False)
        m._contrast = 0
        m._bind_input(X)
        m._reset()
        for _ in range(3):
            m.update_once()
        dt = timed(m.update_once, args.iters)
        print(json.dumps({"config": "configs[2] AuxLaplaceIVA-ISS N=8 F=2049 T=1024 batch={}".format(B),
                          "ms_per_iter": round(dt * 1e3, 3), "mixture_iterations_per_s": round(B / dt, 2),
                          "algorithmic_GBs": round(2 * 16 * N * F * T * B / dt / 1e9, 1)}))
        del m
        torch.cuda.empty_cache()

    if want("iva_ip"):
        # the metric's AuxIVA leg: configs[1] shape, AuxLaplaceIVA with IP1
        N, F, T = 4, 1025, 512
        X = nmf_mixture(1000, N, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        m = AuxLaplaceIVA(spatial_algorithm="IP", record_loss=False)
        m._contrast = 0
        m._bind_input(X)
        m._reset()
        m._C()
        for _ in range(3):
            m.update_once()
        dt = timed(m.update_once, args.iters)
        print(json.dumps({"config": "AuxLaplaceIVA-IP N=4 F=1025 T=512 batch={}".format(B),
                          "ms_per_iter": round(dt * 1e3, 3), "mixture_iterations_per_s": round(B / dt, 2),
                          "algorithmic_GBs": round(2 * 16 * N * F * T * B / dt / 1e9, 1)}))
        del m
        torch.cuda.empty_cache()

    if want("iva_iss4"):
        # the metric's AuxIVA leg with ISS: configs[1] shape
        N, F, T = 4, 1025, 512
        X = nmf_mixture(1000, N, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        m = AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=False)
        m._contrast = 0
        m._bind_input(X)
        m._reset()
        for _ in range(3):
            m.update_once()
        dt = timed(m.update_once, args.iters)
        print(json.dumps({"config": "AuxLaplaceIVA-ISS N=4 F=1025 T=512 batch={}".format(B),
                          "ms_per_iter": round(dt * 1e3, 3), "mixture_iterations_per_s": round(B / dt, 2),
                          "algorithmic_GBs": round(2 * 16 * N * F * T * B / dt / 1e9, 1)}))
        del m
        torch.cuda.empty_cache()

    if want("ilrma_iss"):
        # configs[1] shape with the ISS update (state is the separated spectrogram, no filter)
        from ssspy_amd.bss.ilrma import GaussILRMA

        N, F, T, K = 4, 1025, 512, 16
        X = nmf_mixture(1000, N, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        m = GaussILRMA(n_basis=K, spatial_algorithm="ISS", record_loss=False, rng=np.random.default_rng(0))
        m._bind_input(X)
        m._reset(flooring_fn=m.flooring_fn)
        for _ in range(3):
            m.update_once()
        dt = timed(m.update_once, args.iters)
        print(json.dumps({"config": "GaussILRMA-ISS N=4 F=1025 T=512 K=16 batch={}".format(B),
                          "ms_per_iter": round(dt * 1e3, 3), "mixture_iterations_per_s": round(B / dt, 2)}))
        del m
        torch.cuda.empty_cache()

    if want("ilrma_models"):
        # the other source models / domain on the configs[1] shape (IP1)
        from ssspy_amd.bss.ilrma import GGDILRMA, TILRMA, GaussILRMA

        N, F, T, K = 4, 1025, 512, 16
        X = nmf_mixture(1000, N, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        for name, make in (("TILRMA dof=4", lambda: TILRMA(n_basis=K, dof=4.0, record_loss=False, rng=np.random.default_rng(0))),
                           ("GGDILRMA beta=1", lambda: GGDILRMA(n_basis=K, beta=1.0, record_loss=False, rng=np.random.default_rng(0))),
                           ("GaussILRMA domain=1", lambda: GaussILRMA(n_basis=K, domain=1, record_loss=False, rng=np.random.default_rng(0)))):
            m = make()
            m._bind_input(X)
            m._reset(flooring_fn=m.flooring_fn)
            m._C()
            for _ in range(3):
                m.update_once()
            dt = timed(m.update_once, args.iters)
            print(json.dumps({"config": "{}-IP N=4 F=1025 T=512 K=16 batch={}".format(name, B),
                              "ms_per_iter": round(dt * 1e3, 3),
                              "mixture_iterations_per_s": round(B / dt, 2)}))
            del m
            torch.cuda.empty_cache()

    if want("fastmnmf"):
        # configs[3]: FastGaussMNMF
        M, F, T, K = 4, 1025, 512, 8
        X = nmf_mixture(4000, M, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        m = FastGaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
        m._bind_input(X)
        t0 = time.perf_counter()
        m._reset()
        torch.cuda.synchronize()
        t_reset = time.perf_counter() - t0
        for _ in range(3):
            m.update_once()
        dt = timed(m.update_once, args.iters)
        ds = timed(m._separate_dev, 3)
        print(json.dumps({"config": "configs[3] FastGaussMNMF-IP1 N=M=4 F=1025 T=512 K=8 batch={}".format(B),
                          "ms_per_iter": round(dt * 1e3, 3), "mixture_iterations_per_s": round(B / dt, 2),
                          "algorithmic_GBs": round(4 * 16 * M * F * T * B / dt / 1e9, 1),
                          "wiener_separate_ms": round(ds * 1e3, 3), "reset_s": round(t_reset, 3)}))
        del m
        torch.cuda.empty_cache()

    if want("gmnmf"):
        # GaussMNMF (full-rank SCM) on the configs[3] shape; the CPU figure is the oracle on a
        # 1/16-size slice of the same mixture scaled up (the full shape needs ~10 GB of temporaries)
        M, F, T, K = 4, 1025, 512, 8
        X = nmf_mixture(4000, M, F, T)
        if args.batch > 1:
            X = np.stack([X] * args.batch)
        m = GaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
        m._bind_input(X)
        m._reset()
        for _ in range(2):
            m.update_once()
        dt = timed(m.update_once, max(3, args.iters // 4))
        ds = timed(m._separate_dev, 3)
        out = {"config": "GaussMNMF N=M=4 F=1025 T=512 K=8 batch={}".format(B),
               "ms_per_iter": round(dt * 1e3, 3), "mixture_iterations_per_s": round(B / dt, 2),
               "wiener_separate_ms": round(ds * 1e3, 3)}
        if args.batch == 1:
            from oracle.gmnmf import GaussMNMFOracle

            Fs = 64
            ref = GaussMNMFOracle(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
            ref.reset(X[:, :Fs, :])
            ref.update_once()
            t0 = time.perf_counter()
            ref.update_once()
            out["cpu_oracle_s_per_iter_extrapolated"] = round((time.perf_counter() - t0) * F / Fs, 2)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
