#!/usr/bin/env python3
"""Benchmark of the demixing hot path: GaussILRMA-IP1 update_once() on MI355X.

Contract (one JSON line on rank 0): see the task statement.  A *step* is one
``update_once()`` (basis, activation, weighted covariance, IP1, normalisation) over the
rank's batch of independent synthetic mixtures of BASELINE.json configs[1] shape
(N=4 sources/channels, F=1025 bins, T=512 frames, n_basis=16), fp64/complex128, inputs
resident in HBM before the timed region.  With --gpus N every rank owns its own
``--batch`` mixtures (configs[4]: 128 per GPU); there is no data-path collective, only the
barrier and the max-over-ranks of the elapsed time.

    python bench.py                       # 1 GPU, 128 mixtures, finishes in ~1-2 min
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="independent mixtures per GPU")
    ap.add_argument("--sources", type=int, default=4)
    ap.add_argument("--bins", type=int, default=1025)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--basis", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--no-single", action="store_true", help="skip the batch=1 (configs[1]) leg")
    return ap.parse_args()


def device_mixtures(B, N, F, T, seed, dev):
    """Structured NMF-source mixtures (SURVEY.md 8d formula) generated in HBM with torch's
    generator (allocation/plumbing only); mixture 0 of rank 0 is replaced by the host-seeded
    one the CPU baseline uses."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    X = torch.empty((B, N, F, T), dtype=torch.complex128, device=dev)
    chunk = 8
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        R = (torch.rand((nb, N, F, 4), generator=gen, device=dev, dtype=torch.float64) ** 4) @ (
            torch.rand((nb, N, 4, T), generator=gen, device=dev, dtype=torch.float64) ** 4) + 1e-3
        g = torch.randn((nb, N, F, T, 2), generator=gen, device=dev, dtype=torch.float64)
        S = torch.sqrt(R / 2).unsqueeze(-1) * g
        S = torch.view_as_complex(S.contiguous())
        A = torch.view_as_complex(
            torch.randn((nb, F, N, N, 2), generator=gen, device=dev, dtype=torch.float64))
        X[b0:b0 + nb] = (A @ S.permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    return X


def make_separator(X, K, seed):
    """GaussILRMA bound to device-resident mixtures, with seeded NMF initial state."""
    from ssspy_amd.bss.ilrma import GaussILRMA

    B, N, F, T = X.shape
    sep = GaussILRMA(n_basis=K, spatial_algorithm="IP", record_loss=False,
                     rng=np.random.default_rng(seed))
    sep._bind_input(X)
    sep._reset(flooring_fn=sep.flooring_fn)
    return sep


def timed_steps(sep, steps, stepwise_events):
    """Run `steps` update_once() rounds; with stepwise_events, bracket every kernel group with
    HIP events on the launch stream (torch's current stream is the one the C ABI receives)."""
    names = ("basis", "activation", "wcov", "ip1", "normalize")
    ev = []
    for _ in range(steps):
        if not stepwise_events:
            sep.update_once()
            continue
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        marks[0].record()
        sep.update_basis_mm()
        marks[1].record()
        sep.update_activation_mm()
        marks[2].record()
        from ssspy_amd import _ops

        _ops.ilrma_weighted_covariance(sep._X, sep._state_dev("basis"),
                                       sep._state_dev("activation"), float(sep.domain),
                                       sep._ws, sep._ws_bytes, out=sep._U)
        marks[3].record()
        _ops.update_by_ip1(sep._state_dev("demix_filter"), sep._U, sep._floor, sep._info_tensor())
        sep._state_touch("demix_filter")
        marks[4].record()
        sep.normalize()
        marks[5].record()
        ev.append(marks)
    return names, ev


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # Dry run of the N > 1 control flow on a one-GPU box (development only): every rank on device 0,
    # gloo for the barrier and the timing max.  The driver's runs use neither variable.
    backend = os.environ.get("SSSPY_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("SSSPY_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist

        from ssspy_amd import parallel

        if backend == "nccl":
            parallel.init_from_env(backend="nccl")  # RCCL; the barrier and the timing max only
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world if distributed else 1

    from ssspy_amd import _device as dv
    from ssspy_amd import _ops
    from ssspy_amd.utils.dataset import nmf_mixture

    N, F, T, K, B = args.sources, args.bins, args.frames, args.basis, args.batch
    X = device_mixtures(B, N, F, T, seed=1000 + rank, dev=dev)
    x0_host = None
    if rank == 0:
        x0_host = nmf_mixture(1000, N, F, T)
        X[0] = torch.from_numpy(x0_host).to(dev)
    sep = make_separator(X, K, seed=2000 + rank)
    # scratch of the covariance -> IP1 pair, which timed_steps() launches one kernel group at a time
    sep._U = dv.empty((B, F, N, N, N), dv.c128, dev)
    sep._C()  # static covariance, computed once per call (outside the iteration loop)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed)
    timed_steps(sep, args.warmup, stepwise_events=True)
    fence()
    t0 = time.perf_counter()
    names, events = timed_steps(sep, args.steps, stepwise_events=True)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        elapsed = parallel.max_over_ranks(elapsed, dev if backend == "nccl" else None)
    sep._check_device_errors()

    # per-kernel-group durations from the HIP events of the timed region (this rank)
    dur_ms = {name: 0.0 for name in names}
    for marks in events:
        for k, name in enumerate(names):
            dur_ms[name] += marks[k].elapsed_time(marks[k + 1])
    avg_ms = {name: dur_ms[name] / max(1, len(events)) for name in names}

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    units = B * n_gpus * args.steps  # mixture-iterations
    value = units / elapsed
    pass_bytes = 16.0 * N * F * T * B  # one compulsory pass over the rank's X
    dominant = max(("basis", "activation", "wcov"), key=lambda k: avg_ms[k])
    achieved = pass_bytes / (avg_ms[dominant] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_ilrma_" + dominant, {}).get(
                "hbm_bytes_per_launch_batch{}".format(B))
        except Exception:
            traffic = None
    roofline = {
        # the name rocprofv3 reports for this launch (tuned path of ilrma_fast.hip at these shapes)
        "bound": "hbm", "kernel": {"basis": "k_basis_fast", "activation": "k_activation_fast",
                                   "wcov": "k_wcov_fast"}[dominant],
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "bytes_per_launch": pass_bytes, "avg_launch_ms": round(avg_ms[dominant], 4),
        "per_kernel_ms": {k: round(v, 4) for k, v in avg_ms.items()},
        "iteration_achieved": round(3 * pass_bytes / (elapsed / args.steps) / 1e9, 1),
        "iteration_frac": round(3 * pass_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
    }

    out = {
        "metric": "GaussILRMA-IP1 update_once mixture-iterations/sec (F=1025,T=512,N=4,K=16)",
        "value": round(value, 2),
        "unit": "iterations/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1] GaussILRMA-IP1/MM (N=M={}, F={}, T={}, n_basis={}) x {} "
                        "independent mixtures per GPU (configs[4] shard), record_loss=False"
                        .format(N, F, T, K, B),
            "batch_per_gpu": B, "global_batch": B * n_gpus, "n_sources": N, "n_bins": F,
            "n_frames": T, "n_basis": K, "parallelism": "mixtures sharded, no collective",
        },
        "roofline": roofline,
    }

    # ---- the same batch with record_loss=True semantics (SURVEY 8d asks for both): every iteration
    # is followed by compute_loss(), which syncs the host once per iteration as the reference does
    if not args.no_single and n_gpus == 1:
        nl = max(3, min(10, args.steps))
        sep.compute_loss()
        torch.cuda.synchronize()
        tl = time.perf_counter()
        for _ in range(nl):
            sep.update_once()
            sep.compute_loss()
        torch.cuda.synchronize()
        dtl = (time.perf_counter() - tl) / nl
        out["with_record_loss"] = {
            "workload": "same batch, update_once() + compute_loss() per iteration, {} iterations".format(nl),
            "ms_per_step": round(1e3 * dtl, 4), "iterations_per_s": round(B / dtl, 2),
        }

    # ---- the metric's AuxIVA leg: AuxLaplaceIVA (IP1) on the same resident batch, two passes over X
    # per iteration (frame powers, weighted covariance)
    if not args.no_single and n_gpus == 1:
        from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast

        iva = AuxLaplaceIVA(spatial_algorithm="IP", record_loss=False)
        iva._contrast = _device_contrast(iva.contrast_fn, iva.d_contrast_fn)
        iva._bind_input(X)
        iva._reset()
        iva._C()
        for _ in range(3):
            iva.update_once()
        torch.cuda.synchronize()
        ni = max(5, args.steps)
        ti = time.perf_counter()
        for _ in range(ni):
            iva.update_once()
        torch.cuda.synchronize()
        dti = (time.perf_counter() - ti) / ni
        iva._check_device_errors()
        out["auxiva_ip"] = {
            "workload": "AuxLaplaceIVA-IP1, same batch ({} x N={} F={} T={}), {} iterations".format(
                B, N, F, T, ni),
            "ms_per_step": round(1e3 * dti, 4), "iterations_per_s": round(B / dti, 2),
            "achieved_GBs": round(2 * pass_bytes / dti / 1e9, 1),
            "frac_of_hbm_peak": round(2 * pass_bytes / dti / 1e9 / HBM_PEAK_GBS, 4),
        }
        del iva

    # ---- configs[1] exactly: ONE mixture, fused update_once (one C-ABI call per iteration)
    if not args.no_single and n_gpus == 1:
        sep1 = make_separator(X[:1].clone(), K, seed=2000)
        sep1._C()
        for _ in range(10):
            sep1.update_once()
        torch.cuda.synchronize()
        n1 = 200
        t1 = time.perf_counter()
        for _ in range(n1):
            sep1.update_once()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        out["single_mixture"] = {
            "workload": "configs[1]: 1 mixture, {} iterations".format(n1),
            "iterations_per_s": round(n1 / dt1, 1), "ms_per_iter": round(1e3 * dt1 / n1, 4),
            "achieved_GBs": round(3 * 16.0 * N * F * T * n1 / dt1 / 1e9, 1),
        }

    # ---- CPU baseline: the NumPy oracle (reference expression structure) on the host cores
    if not args.no_cpu_baseline and n_gpus == 1:
        from oracle.ilrma import GaussILRMAOracle

        ref = GaussILRMAOracle(n_basis=K, spatial_algorithm="IP", record_loss=False)
        ref.reset(x0_host, basis=np.random.default_rng(1).random((N, F, K)),
                  activation=np.random.default_rng(2).random((N, K, T)))
        ref.update_once()  # warm-up
        times = []
        for _ in range(args.cpu_iters):
            c0 = time.perf_counter()
            ref.update_once()
            times.append(time.perf_counter() - c0)
        med = float(np.median(times))
        blas_threads = 1
        try:
            from threadpoolctl import threadpool_info

            blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        except Exception:
            pass
        out["cpu_baseline"] = {
            "value": round(1.0 / med, 4), "unit": "iterations/s", "cores": blas_threads,
            "kind": "port",
            "sample": "oracle.ilrma.GaussILRMAOracle.update_once (NumPy restatement of the "
                      "reference, same broadcast structure) on 1 mixture of configs[1], median of "
                      "{} iterations after 1 warm-up; host has {} logical CPUs; NumPy ufuncs are "
                      "single-threaded, BLAS may use {} threads".format(
                          args.cpu_iters, os.cpu_count(), blas_threads),
            "s_per_iter_median": round(med, 4),
        }
        out["speedup_vs_cpu_per_mixture_iteration"] = round(value * med, 1)

    print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
