#!/usr/bin/env python3
"""Benchmark of the demixing hot path: GaussILRMA-IP1 update_once() on MI355X.

Contract (one JSON line on rank 0): see the task statement.  A *step* is one
``update_once()`` -- the product's fused C-ABI call: basis, activation, weighted covariance, IP1,
normalisation -- over the rank's batch of independent synthetic mixtures of BASELINE.json
configs[1] shape (N=4 sources/channels, F=1025 bins, T=512 frames, n_basis=16),
fp64/complex128, inputs resident in HBM before the timed region.  Mixture b of rank r is
``nmf_mixture(seed = 1000 + r*batch + b)``, regenerated on the host; the SHA-256 of
mixture 0 is checked against tests/golden/input_sha256.json.  With --gpus N every rank owns its
own ``--batch`` mixtures (configs[4]: 128 per GPU); there is no data-path collective, only the
barrier and the max-over-ranks of the elapsed time.

After the timed region, on rank 0 at N=1 (none of it is part of ``value``):
  * a second loop over the same steps with HIP events around every kernel group -> ``roofline``;
  * ``with_record_loss`` (the reference default), the AuxIVA-IP leg of the metric;
  * ``configs``: BASELINE configs[1] literally (one mixture), configs[2] (AuxIVA-ISS, N=8,
    F=2049, T=1024) and configs[3] (FastGaussMNMF, N=M=4, K=8), one mixture and a batch each,
    every one with its own CPU baseline (the matching oracle class on this box's host cores).

    python bench.py                       # 1 GPU, 128 mixtures, finishes in ~2-3 min
    python bench.py --gpus 8              # starts 8 ranks itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""

import argparse
import gc
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
FP64_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz; fp64 MFMA and VALU share it
#                          (profiles/r01_f64_rates.txt: v_mfma_f64_16x16x4 = 64 clocks, no overlap)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of exactly --steps steps each, run back to back; `value` is "
                         "the MEDIAN region, all of them are listed (min / median / max)")
    ap.add_argument("--batch", type=int, default=128, help="independent mixtures per GPU")
    ap.add_argument("--sources", type=int, default=4)
    ap.add_argument("--bins", type=int, default=1025)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--basis", type=int, default=16)
    ap.add_argument("--power-seconds", type=float, default=2.5,
                    help="length of the power / clock sampling leg (rank 0, N=1, not with --no-extra)")
    ap.add_argument("--e2e-mixtures", type=int, default=256,
                    help="host-resident mixtures of the pipelined end-to-end leg (0 = skip)")
    ap.add_argument("--e2e-sub-batch", type=int, default=64)
    ap.add_argument("--e2e-iters", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--no-extra", "--no-single", dest="no_extra", action="store_true",
                    help="only the headline (skip the loss / AuxIVA / other-config legs)")
    ap.add_argument("--no-pairwise", action="store_true",
                    help="skip the IP2 / ISS2 / IPA legs (`pairwise_ipa`)")
    ap.add_argument("--only-pairwise", action="store_true",
                    help="profiling: after the headline, run only the `pairwise_ipa` legs")
    ap.add_argument("--other-batch", type=int, default=32,
                    help="mixtures in the batched configs[2] / configs[3] legs")
    ap.add_argument("--mnmf-batch", type=int, default=128,
                    help="mixtures in the second batched configs[3] leg (the headline's shard size)")
    return ap.parse_args()


def make_separator(X, K, seed, record_loss=False):
    """GaussILRMA bound to device-resident mixtures, with seeded NMF initial state."""
    from ssspy_amd.bss.ilrma import GaussILRMA

    sep = GaussILRMA(n_basis=K, spatial_algorithm="IP", record_loss=record_loss,
                     rng=np.random.default_rng(seed))
    sep._bind_input(X)
    sep._reset(flooring_fn=sep.flooring_fn)
    sep._C()  # static covariance, computed once per call (outside the iteration loop)
    return sep


def settle_host():
    """A full collection of CPython's oldest generation walks every object torch and NumPy created at
    import (42 ms here) and lands in whichever short loop crosses the allocation threshold
    (benchmarks/tools/iter_times.py: iteration 21 of a 160 us loop took 42 ms).  Collect now and move
    the survivors to the permanent generation; the collector stays enabled in the timed loops."""
    gc.collect()
    gc.freeze()


def time_loop(fn, n, reps=3):
    """Seconds per call: the median of `reps` regions of exactly `n` calls each (the side legs only;
    the headline has its own five regions).  One region was what rounds 1-5 timed, and a 20 ms leg
    that starts right after seconds of host work (input generation, a pageable upload) caught the
    GPU's clocks on their way back up: configs[3] at 32 mixtures read 1.06 / 1.99 / 3.12 ms per
    iteration on three boxes of the same code while benchmarks/other_configs.py gave 1.08 each time."""
    settle_host()
    times = []
    for _ in range(max(1, reps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / n)
    return sorted(times)[len(times) // 2]


def kernel_group_events(sep, steps):
    """The same `steps` iterations as separate kernel groups (equal to the fused call:
    tests/test_gpu_parity.py::test_gauss_ilrma_step_methods_match_fused_update), with HIP events on
    the launch stream (torch's current stream is the one the C ABI receives) around each."""
    from ssspy_amd import _device as dv
    from ssspy_amd import _ops

    names = ("basis", "activation", "wcov", "ip1", "normalize")
    B, N, F, T = sep._X.shape
    if sep._U is None:
        sep._U = dv.empty((B, F, N, N, N), dv.c128, sep._X.device)
    ev = []
    for _ in range(steps):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        marks[0].record()
        sep.update_basis_mm()
        marks[1].record()
        sep.update_activation_mm()
        marks[2].record()
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
    torch.cuda.synchronize()
    dur = {name: 0.0 for name in names}
    for marks in ev:
        for k, name in enumerate(names):
            dur[name] += marks[k].elapsed_time(marks[k + 1])
    return {name: dur[name] / max(1, len(ev)) for name in names}


def power_leg(sep, seconds):
    """update_once() back to back for `seconds` while a host thread samples the amdgpu hwmon power
    sensor and shader clock (benchmarks/power_profile.py): the evidence for roofline.bound."""
    try:
        import importlib.util

        spec = importlib.util.spec_from_file_location(
            "power_profile", os.path.join(ROOT, "benchmarks", "power_profile.py"))
        pp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pp)
        sampler = pp.Sampler()
        if not sampler.cards:
            return {"error": "no amdgpu hwmon power sensor visible"}
        n = 0
        with sampler:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                for _ in range(20):
                    sep.update_once()
                n += 20
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        rec = sampler.summary(0.5)  # second half: the sensor is a moving average
        rec["ms_per_step"] = round(1e3 * dt / n, 4)
        rec["workload"] = "the headline batch, update_once() back to back for {:.1f} s".format(dt)
        rec["sclk_max_mhz"] = 2400
        rec["note"] = ("socket power at the cap with the shader clock below its maximum = the pass "
                       "rate is set by the energy of a tile; a read-only stream of the same bytes "
                       "draws 1087 W at 6.0 TB/s and full clock (profiles/r04_power_cap.md)")
        return rec
    except Exception as exc:  # never cost the headline line
        return {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}


def joint_power_leg(sep, seconds, rank, fence):
    """All ranks run update_once() back to back for `seconds` between two barriers while rank 0
    samples the power sensor and shader clock of EVERY amdgpu card of the node.  Every rank passes
    both barriers whatever happens to the sampler (a rank that skipped one would hang the others)."""
    sampler, err = None, None
    if rank == 0:
        try:
            import importlib.util

            spec = importlib.util.spec_from_file_location(
                "power_profile", os.path.join(ROOT, "benchmarks", "power_profile.py"))
            pp = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(pp)
            sampler = pp.Sampler()
            if not sampler.cards:
                sampler, err = None, "no amdgpu hwmon power sensor visible"
        except Exception as exc:
            sampler, err = None, "{}: {}".format(type(exc).__name__, str(exc)[:200])
    fence()
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            sep.update_once()
        torch.cuda.synchronize()
    if sampler is not None:
        sampler.__exit__()
    fence()
    if rank != 0:
        return None
    if sampler is None:
        return {"error": err}
    try:
        cards = [sampler.summary(0.5, card=i) for i in range(len(sampler.cards))]
        return {"cards": cards, "seconds": seconds,
                "note": "every rank iterating at once; one entry per amdgpu card with a power sensor "
                        "(sysfs order, not rank order)"}
    except Exception as exc:
        return {"error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}


def blas_threads():
    try:
        from threadpoolctl import threadpool_info

        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return 1


def cpu_leg(make_ref, n_iters, what):
    """Median seconds per update_once() of an oracle instance on the host (1 warm-up iteration)."""
    ref = make_ref()
    ref.update_once()
    times = []
    for _ in range(n_iters):
        c0 = time.perf_counter()
        ref.update_once()
        times.append(time.perf_counter() - c0)
    med = float(np.median(times))
    threads = blas_threads()
    return {
        "value": round(1.0 / med, 4), "unit": "iterations/s", "cores": threads, "kind": "port",
        "sample": "{} (NumPy restatement of the reference, same broadcast structure) on 1 mixture, "
                  "median of {} iterations after 1 warm-up; host has {} logical CPUs; NumPy ufuncs "
                  "are single-threaded, BLAS may use {} threads".format(what, n_iters, os.cpu_count(),
                                                                       threads),
        "s_per_iter_median": round(med, 4),
    }


def rate_entry(workload, dt, n_mixtures, bytes_per_mixture_iter):
    gbs = bytes_per_mixture_iter * n_mixtures / dt / 1e9
    return {"workload": workload, "ms_per_step": round(1e3 * dt, 4),
            "iterations_per_s": round(n_mixtures / dt, 2), "achieved_GBs": round(gbs, 1),
            "frac": round(gbs / HBM_PEAK_GBS, 4)}


def other_configs(args, dev, x0_host, pins, cpu_configs1):
    """BASELINE configs[1] literally, configs[2] and configs[3]: update_once() with
    record_loss=False, one mixture and a batch, algorithmic bytes of SURVEY.md 8d, CPU baseline."""
    from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture_batch, sha256_of

    out = {}
    cpu = not args.no_cpu_baseline
    Bo = args.other_batch

    # ---- configs[1]: ONE mixture, GaussILRMA-IP1 (33.6 MB: the working set sits in the caches)
    N, F, T, K = 4, 1025, 512, 16
    sep1 = make_separator(torch.from_numpy(x0_host[None]).to(dev), K, seed=2000)
    for _ in range(20):
        sep1.update_once()
    dt = time_loop(sep1.update_once, 300)
    sep1._check_device_errors()
    ent = rate_entry("configs[1]: GaussILRMA-IP1 N=4 F=1025 T=512 n_basis=16, 1 mixture, 300 "
                     "iterations", dt, 1, 3 * 16.0 * N * F * T)
    if cpu_configs1 is not None:
        ent["cpu_baseline"] = cpu_configs1  # same oracle on the same mixture as the headline's
    out["configs1_single"] = ent
    del sep1
    # ... and the metric's AuxIVA leg on the same single mixture (IP1: two passes over X; ISS: one
    # read + one write of the separated spectrogram)
    for key, algo in (("auxiva_ip_single", "IP"), ("auxiva_iss_single", "ISS")):
        iva = AuxLaplaceIVA(spatial_algorithm=algo, record_loss=False)
        iva._contrast = _device_contrast(iva.contrast_fn, iva.d_contrast_fn)
        iva._bind_input(torch.from_numpy(x0_host[None]).to(dev))
        iva._reset()
        if algo == "IP":
            iva._C()
        for _ in range(20):
            iva.update_once()
        dt = time_loop(iva.update_once, 300)
        iva._check_device_errors()
        out[key] = rate_entry("AuxLaplaceIVA-{} N=4 F=1025 T=512, 1 mixture, 300 iterations".format(
            "IP1" if algo == "IP" else "ISS"), dt, 1, 2 * 16.0 * N * F * T)
        del iva

    # ---- configs[2]: AuxLaplaceIVA-ISS, N=8, F=2049, T=1024 (2 passes: read Y, write Y)
    N, F, T = 8, 2049, 1024
    Xh = nmf_mixture_batch(3000, Bo, N, F, T)
    sha_ok = sha256_of(Xh[0]) == pins["configs2_seed3000_N8_F2049_T1024"]["sha256"]
    ent = {"input_sha256_ok": sha_ok}
    for tag, nb, iters in (("single", 1, 100), ("batch", Bo, 20)):
        m = AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=False)
        m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
        m._bind_input(torch.from_numpy(Xh[:nb]).to(dev))
        m._reset()
        for _ in range(3):
            m.update_once()
        dt = time_loop(m.update_once, iters)
        m._check_device_errors()
        ent[tag] = rate_entry("configs[2]: AuxLaplaceIVA-ISS N=8 F=2049 T=1024, {} mixture(s), {} "
                              "iterations".format(nb, iters), dt, nb, 2 * 16.0 * N * F * T)
        del m
        torch.cuda.empty_cache()
    if cpu:
        from oracle.iva import AuxIVAOracle

        def make():
            ref = AuxIVAOracle(spatial_algorithm="ISS", contrast="laplace", record_loss=False)
            ref.reset(Xh[0])
            return ref

        ent["cpu_baseline"] = cpu_leg(make, 2, "oracle.iva.AuxIVAOracle.update_once (ISS)")
    out["configs2"] = ent
    del Xh

    # ---- configs[3]: FastGaussMNMF-IP1, N=M=4, F=1025, T=512, n_basis=8 (4 passes)
    M, F, T, K = 4, 1025, 512, 8
    Bm = max(Bo, args.mnmf_batch)
    Xh = nmf_mixture_batch(4000, Bm, M, F, T)
    sha_ok = sha256_of(Xh[0]) == pins["configs3_seed4000_N4_F1025_T512"]["sha256"]
    ent = {"input_sha256_ok": sha_ok}
    legs = [("single", 1, 100), ("batch", Bo, 20)]
    if Bm > Bo:
        legs.append(("batch{}".format(Bm), Bm, 10))
    for tag, nb, iters in legs:
        m = FastGaussMNMF(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
        m._bind_input(torch.from_numpy(Xh[:nb]).to(dev))
        m._reset()
        for _ in range(3):
            m.update_once()
        dt = time_loop(m.update_once, iters)
        ent[tag] = rate_entry("configs[3]: FastGaussMNMF-IP1 N=M=4 F=1025 T=512 n_basis=8, {} "
                              "mixture(s), {} iterations".format(nb, iters), dt, nb,
                              4 * 16.0 * M * F * T)
        ent[tag]["wiener_separate_ms"] = round(1e3 * time_loop(m._separate_dev, 3), 3)
        m._check_device_errors()
        if tag == "batch":
            # the separator's own record_loss=True loop (the reference's default): loss terms from the
            # |Qx|^2 hand-over, kept in HBM until the end of the run
            m.record_loss, m.loss = True, []
            t0 = time.perf_counter()
            assert m._iterate_with_resident_loss(iters, True)
            torch.cuda.synchronize()
            dtl = (time.perf_counter() - t0) / iters
            ent["batch_with_record_loss"] = {
                "ms_per_step": round(1e3 * dtl, 4), "iterations_per_s": round(nb / dtl, 2),
                "note": "{} iterations + the initial loss, {} mixtures".format(iters, nb)}
        del m
        torch.cuda.empty_cache()
    if cpu:
        from oracle.mnmf import FastGaussMNMFOracle

        def make():
            ref = FastGaussMNMFOracle(n_basis=K, record_loss=False)
            ref.reset(Xh[0], basis=np.random.default_rng(1).random((M, F, K)),
                      activation=np.random.default_rng(2).random((M, K, T)),
                      spatial=np.random.default_rng(4).random((F, M, M)))
            return ref

        ent["cpu_baseline"] = cpu_leg(make, 3, "oracle.mnmf.FastGaussMNMFOracle.update_once")
    out["configs3"] = ent
    return out


def pairwise_ipa_legs(args, dev, Xbatch_host):
    """Round 5 (round-4 verdict item 2): the pairwise and IPA spatial updates had no timing anywhere.
    update_once() of GaussILRMA (IP2 / ISS2 / IPA), AuxLaplaceIVA (IP2 / ISS2 / IPA) at the configs[1]
    shape and FastGaussMNMF with the IP2 diagonaliser at the configs[3] shape, on 1 / 32 / 128
    mixtures of the headline's batch.  `bytes` is the algorithmic traffic per mixture-iteration in
    passes of A = 16 N F T over the spectrograms (stated per leg; IP1 / ISS1 beside them for scale)."""
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast
    from ssspy_amd.bss.mnmf import FastGaussMNMF

    N, F, T = Xbatch_host.shape[1:]
    A = 16.0 * N * F * T

    def ilrma(algo):
        return lambda: GaussILRMA(n_basis=16, spatial_algorithm=algo, record_loss=False,
                                  rng=np.random.default_rng(2000))

    def iva(algo):
        def make():
            m = AuxLaplaceIVA(spatial_algorithm=algo, record_loss=False)
            m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
            return m
        return make

    def fmnmf(algo):
        return lambda: FastGaussMNMF(n_basis=8, diagonalizer_algorithm=algo, record_loss=False,
                                     rng=np.random.default_rng(0))

    legs = [
        ("ilrma_ip1", ilrma("IP1"), 3, "basis, activation, covariance passes over X"),
        ("ilrma_ip2", ilrma("IP2"), 3, "basis, activation, covariance passes over X"),
        ("ilrma_iss1", ilrma("ISS1"), lambda nb: 4 if nb * F < 4096 else 3,
         "one mixture: basis, activation passes over Y, fused sweep read + write Y (4); batches: "
         "basis, activation, covariance passes over X read through the filters the updates imply, "
         "W <- G W per bin, Y formed when `output` is read (3)"),
        ("ilrma_iss2", ilrma("ISS2"), 3, "basis, activation, covariance passes over X read through the "
                                         "filters the updates imply (W <- G W per bin; Y formed when "
                                         "`output` is read)"),
        ("ilrma_ipa", ilrma("IPA"), 3, "basis, activation, covariance passes over X read through the "
                                       "filters the updates imply (W <- G W per bin; Y formed when "
                                       "`output` is read)"),
        ("auxiva_ip2", iva("IP2"), 2 * N, "per pair (N of them: (0,1), (1,2), ... , (N-1,0)) a frame-power and "
                                          "a covariance pass over X: the weights are recomputed from the "
                                          "current filters before every pair (ssspy/bss/iva.py:1795-1915)"),
        ("auxiva_iss2", iva("ISS2"), 2, "frame-power and covariance passes over X read through the filters "
                                        "the updates imply (W <- G W per bin; Y formed when `output` is "
                                        "read)"),
        ("auxiva_ipa", iva("IPA"), 2, "frame-power and covariance passes over X read through the filters "
                                      "the updates imply (W <- G W per bin; Y formed when `output` is "
                                      "read)"),
        ("fmnmf_ip2", fmnmf("IP2"), 4, "basis, activation, covariance, spatial passes (as IP1)"),
    ]
    out = {"shape": "N=M={} F={} T={}; ILRMA n_basis=16, FastGaussMNMF n_basis=8".format(N, F, T),
           "A_bytes": A}
    batches = [b for b in (1, 32, 128) if b <= Xbatch_host.shape[0]]
    for nb in batches:
        X = torch.from_numpy(Xbatch_host[:nb]).to(dev)
        for key, make, passes, what in legs:
            try:
                m = make()
                m._bind_input(X)
                m._reset()
                iters = 40 if nb == 1 else (10 if nb <= 32 else 5)
                for _ in range(3):
                    m.update_once()
                dt = time_loop(m.update_once, iters)
                m._check_device_errors()
                np_ = passes(nb) if callable(passes) else passes
                ent = out.setdefault(key, {"passes": what})
                ent["b{}".format(nb)] = {
                    "ms_per_step": round(1e3 * dt, 4), "iterations_per_s": round(nb / dt, 1),
                    "passes_of_A": np_, "achieved_GBs": round(np_ * A * nb / dt / 1e9, 1),
                    "frac": round(np_ * A * nb / dt / 1e9 / HBM_PEAK_GBS, 4)}
                del m
            except Exception as exc:  # an extra leg must never cost the headline line
                out.setdefault(key, {})["b{}".format(nb)] = {
                    "error": "{}: {}".format(type(exc).__name__, str(exc)[:200])}
            torch.cuda.empty_cache()
        del X
        torch.cuda.empty_cache()
    return out


def end_to_end(args, dev, Xbatch_host, pins, cpu):
    """`call_end_to_end`: host NumPy in -> separator(X, n_iter) -> host NumPy out, PCIe and every
    reset / restore / separate step included (round-3 verdict item 8).  One mixture of configs[1],
    [2], [3] through the plain `__call__`, and a host-resident configs[4]-style batch through
    `parallel.separate_pipelined` (upload of sub-batch k + 1 and download of k - 1 overlap the
    iterations of k) next to the same batch processed serially.  The CPU figure beside each is the
    per-iteration median of the matching oracle leg x n_iter (an extrapolation, stated as such:
    100 oracle iterations take 45 / 160 / 58 s)."""
    from ssspy_amd import parallel
    from ssspy_amd.bss.ilrma import GaussILRMA
    from ssspy_amd.bss.iva import AuxLaplaceIVA
    from ssspy_amd.bss.mnmf import FastGaussMNMF
    from ssspy_amd.utils.dataset import nmf_mixture

    n_iter = args.e2e_iters
    out = {"n_iter": n_iter}

    def one(tag, make, X, cpu_key):
        make()(X, n_iter=2)  # warm-up: allocator, kernel images
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        Y = make()(X, n_iter=n_iter)
        dt = time.perf_counter() - t0
        assert Y.shape == X.shape and np.isfinite(Y).all()
        ent = {"workload": tag, "seconds": round(dt, 4), "iterations_per_s": round(n_iter / dt, 1),
               "host_bytes_in_out": 2 * X.nbytes}
        c = cpu.get(cpu_key) if cpu else None
        if c:
            ent["cpu_extrapolated_s"] = round(n_iter * c["s_per_iter_median"], 1)
            ent["speedup_vs_cpu"] = round(ent["cpu_extrapolated_s"] / dt, 1)
        return ent

    X1 = Xbatch_host[0]
    out["configs1"] = one("configs[1] GaussILRMA-IP1 N=4 F=1025 T=512 n_basis=16, record_loss=True, "
                          "projection back: __call__(X, n_iter={})".format(n_iter),
                          lambda: GaussILRMA(n_basis=16, rng=np.random.default_rng(0)), X1, "configs1")
    X2 = nmf_mixture(3000, 8, 2049, 1024)
    out["configs2"] = one("configs[2] AuxLaplaceIVA-ISS N=8 F=2049 T=1024: __call__(X, n_iter={})".format(n_iter),
                          lambda: AuxLaplaceIVA(spatial_algorithm="ISS"), X2, "configs2")
    del X2
    X3 = nmf_mixture(4000, 4, 1025, 512)
    out["configs3"] = one("configs[3] FastGaussMNMF-IP1 N=M=4 F=1025 T=512 n_basis=8 incl. the Wiener "
                          "filter: __call__(X, n_iter={})".format(n_iter),
                          lambda: FastGaussMNMF(n_basis=8, rng=np.random.default_rng(0)), X3, "configs3")
    del X3
    torch.cuda.empty_cache()

    Be = args.e2e_mixtures
    if Be > 0:
        nb = Xbatch_host.shape[0]
        shape = (Be,) + tuple(Xbatch_host.shape[1:])
        Xp = torch.empty(shape, dtype=torch.complex128, pin_memory=True)  # the host-resident batch
        for lo in range(0, Be, nb):  # the generated mixtures, repeated to the requested count
            n = min(nb, Be - lo)
            Xp[lo:lo + n].copy_(torch.from_numpy(Xbatch_host[:n]))
        Yp = torch.empty(shape, dtype=torch.complex128, pin_memory=True)

        def make():
            return GaussILRMA(n_basis=16, record_loss=False, rng=np.random.default_rng(0))

        sub = min(args.e2e_sub_batch, Be)
        parallel.separate_pipelined(make, Xp[:sub], sub, n_iter=2, out=Yp[:sub])  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parallel.separate_pipelined(make, Xp, sub, n_iter=n_iter, out=Yp)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        t0 = time.perf_counter()
        for lo in range(0, Be, sub):  # the same sub-batches, one after the other through __call__
            Yp[lo:lo + sub].copy_(torch.from_numpy(make()(Xp[lo:lo + sub].numpy(), n_iter=n_iter)))
        dts = time.perf_counter() - t0
        out["configs4_host_batch"] = {
            "workload": "{} host-resident (pinned) mixtures of the configs[1] shape ({} generated ones "
                        "repeated), GaussILRMA-IP1 record_loss=False, {} iterations each, sub-batches "
                        "of {}: parallel.separate_pipelined vs one __call__ per sub-batch".format(
                            Be, nb, n_iter, sub),
            "pipelined_seconds": round(dtp, 3), "serial_seconds": round(dts, 3),
            "pipelined_mixture_iterations_per_s": round(Be * n_iter / dtp, 1),
            "serial_mixture_iterations_per_s": round(Be * n_iter / dts, 1),
            "host_GB_in_plus_out": round(2 * Xp.numel() * 16 / 1e9, 2),
            "pcie_inclusive_GBs": round(2 * Xp.numel() * 16 / dtp / 1e9, 1),
        }
        del Xp, Yp
    return out


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script (one process per
    GPU, RCCL) through torch.distributed.run on the loopback address and hand back its exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus {} but WORLD_SIZE={} (launch one rank per GPU, or run "
                         "`python bench.py --gpus N` and let it start them)".format(args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # Dry run of the N > 1 control flow on a one-GPU box (development / tests only): every rank on
    # device 0, gloo for the barrier and the timing max.  The driver's runs use neither variable.
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

    from ssspy_amd.utils.dataset import nmf_mixture_batch, sha256_of

    N, F, T, K, B = args.sources, args.bins, args.frames, args.basis, args.batch
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "input_sha256.json")))
    first_seed = 1000 + rank * B
    Xh = nmf_mixture_batch(first_seed, B, N, F, T)
    x0_host = Xh[0].copy()
    x0_sha = sha256_of(x0_host)
    pin = pins.get("configs1_seed1000_N4_F1025_T512")
    sha_ok = None
    if rank == 0 and (N, F, T) == tuple(pin["shape"]):
        sha_ok = x0_sha == pin["sha256"]
    X = torch.from_numpy(Xh).to(dev)
    if not (rank == 0 and world == 1 and not args.no_extra):
        del Xh  # (the end-to-end legs of the N = 1 run start from the host copy)
        Xh = None
    sep = make_separator(X, K, seed=2000 + rank)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: exactly `steps` update_once() calls (one fused C-ABI call each),
    # bracketed by barrier + synchronize on both sides.  It is run `--repeats` times back to back
    # (round-3 verdict: a single 66 ms shot on a box whose clocks are still settling): every region is
    # reported, `value` is the median one.
    for _ in range(args.warmup):
        sep.update_once()
    settle_host()
    regions, own_regions = [], []
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sep.update_once()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0  # this rank's own work, before it waits for the others
        fence()
        dt = time.perf_counter() - t0
        if distributed:
            dt = parallel.max_over_ranks(dt, dev if backend == "nccl" else None)
        regions.append(dt)
        own_regions.append(own)
    elapsed = float(np.median(regions))
    sep._check_device_errors()

    # ---- N > 1: what each rank did on its own, and the power of every card while all ranks run
    # (round-4 verdict item 10: the first real 8-GPU run should show at a glance whether a slow rank
    # or a shared power budget limits scaling).  After the timed regions; none of it is `value`.
    per_rank = joint_power = None
    if distributed:
        # (a plain all_gather of one double per rank, on the device for RCCL: the same kind of
        # collective as the all-reduce of max_over_ranks)
        mine = torch.tensor([float(np.median(own_regions))], dtype=torch.float64,
                            device=dev if backend == "nccl" else torch.device("cpu"))
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        joint_power = joint_power_leg(sep, min(args.power_seconds, 1.5), rank, fence)
        if rank == 0:
            rates = [B * args.steps / float(g.item()) for g in gathered]
            per_rank = {
                "mixture_iterations_per_s": [round(r, 1) for r in rates],
                "min": round(min(rates), 1), "median": round(float(np.median(rates)), 1),
                "max": round(max(rates), 1), "slowest_rank": int(np.argmin(rates)),
                "sum": round(sum(rates), 1),
                "note": "each rank's own steps / its own time to drain them (median region), "
                        "before the closing barrier; `value` divides by the slowest rank's time",
            }

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    # ---- second loop, same steps as separate kernel groups under HIP events -> roofline
    avg_ms = kernel_group_events(sep, args.steps)
    sep._check_device_errors()

    units = B * n_gpus * args.steps  # mixture-iterations
    value = units / elapsed
    pass_bytes = 16.0 * N * F * T * B  # one compulsory pass over the rank's X
    dominant = max(("basis", "activation", "wcov"), key=lambda k: avg_ms[k])
    achieved = pass_bytes / (avg_ms[dominant] * 1e-3) / 1e9
    kernel_names = {"basis": "k_basis_fast", "activation": "k_activation_fast", "wcov": "k_wcov_fast"}
    traffic = None
    traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("k_ilrma_" + dominant, {}).get("hbm_bytes_per_launch_batch{}".format(B))
            traffic_source = "profiles/roofline_traffic.json ({})".format(tj.get("_round", "r01"))
            from ssspy_amd.utils.dataset import kernel_sources_sha256
            if tj.get("_kernel_sources_sha256") not in (None, kernel_sources_sha256()):
                # the counters were collected on other kernel sources: do not quote them
                traffic = None
                traffic_source += " -- stale: the pass kernels changed since it was collected"
        except Exception:
            traffic = None
    # fp64 work of a pass per (source, bin, frame): both NMF passes run GEMM1 (16 FMA on the matrix
    # pipe), GEMM2 for num and den (32), y = W x (16), |y|^2 (2), the Newton reciprocal (5), the
    # numerator factor (2) = 73 FMA = 146 flop (the SQ counters give 40.45 / 38.44 Gflop per launch at
    # 128 mixtures against 39.2 from this count; the covariance pass counts 26.72: its figure is the
    # counters', profiles/r03_pmc_sq_digest.md).
    elems = float(N) * F * T * B
    flop = {"basis": 146.0 * elems, "activation": 146.0 * elems, "wcov": 26.72e9 * elems / (4.0 * 1025 * 512 * 128)}
    roofline = {
        # the name rocprofv3 reports for this launch (tuned path of ilrma_fast.hip at these shapes)
        # bound: what limits the pass.  Neither roof of the classic model is reached (frac / fp64_frac
        # below): the passes run at the socket power cap with the shader clock throttled -- see
        # "power" below (sampled live in this run) and profiles/r04_power_cap.md.  achieved / peak /
        # frac stay on the HBM axis, as the contract asks.
        "bound": "power", "kernel": kernel_names[dominant],
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "fp64_frac": round(flop[dominant] / (avg_ms[dominant] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
        "traffic": traffic, "traffic_source": traffic_source,
        "bytes_per_launch": pass_bytes, "avg_launch_ms": round(avg_ms[dominant], 4),
        "per_kernel_ms": {k: round(v, 4) for k, v in avg_ms.items()},
        "per_kernel_frac": {k: round(pass_bytes / (avg_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                            for k in ("basis", "activation", "wcov")},
        "per_kernel_fp64_frac": {k: round(flop[k] / (avg_ms[k] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)
                                 for k in ("basis", "activation", "wcov")},
        "measured": "HIP events around each kernel group in a second loop of the same steps "
                    "(the timed region runs the fused update_once())",
        "iteration_achieved": round(3 * pass_bytes / (elapsed / args.steps) / 1e9, 1),
        "iteration_frac": round(3 * pass_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
    }
    if n_gpus == 1 and not args.no_extra:
        roofline["power"] = power_leg(sep, args.power_seconds)
    if joint_power is not None:
        roofline["power_all_ranks"] = joint_power
    out = {
        "metric": "GaussILRMA-IP1 update_once mixture-iterations/sec (F=1025,T=512,N=4,K=16)",
        "value": round(value, 2),
        "unit": "iterations/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "timed_regions": {
            "count": len(regions), "steps_each": args.steps,
            "ms_per_step": [round(1e3 * r / args.steps, 4) for r in regions],
            "value_min": round(units / max(regions), 2), "value_median": round(value, 2),
            "value_max": round(units / min(regions), 2), "value_first_region": round(units / regions[0], 2),
            "note": "each region = exactly `steps` update_once() calls between barrier + synchronize; "
                    "`value` is the median region",
            "side_legs": "every ms_per_step outside this block (configs, pairwise_ipa, auxiva_*) is the "
                         "median of 3 regions of the stated number of iterations (time_loop)",
        },
        "per_rank": per_rank,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1] GaussILRMA-IP1/MM (N=M={}, F={}, T={}, n_basis={}) x {} "
                        "independent mixtures per GPU (configs[4] shard), record_loss=False, "
                        "update_once() = one fused C-ABI call".format(N, F, T, K, B),
            "batch_per_gpu": B, "global_batch": B * n_gpus, "n_sources": N, "n_bins": F,
            "n_frames": T, "n_basis": K, "parallelism": "mixtures sharded, no collective",
            "input": "nmf_mixture(seed=1000+b) per mixture b (SURVEY 8d)",
            "input_sha256_mixture0": x0_sha, "input_sha256_ok": sha_ok,
        },
        "roofline": roofline,
    }

    extra = not args.no_extra and n_gpus == 1
    # ---- the same batch with record_loss=True semantics (SURVEY 8d asks for both): the separator's
    # own loop, update_once() then the loss bookkeeping of IterativeMethodBase
    if extra:
        try:
            nl = max(3, args.steps)
            sep.record_loss, sep.loss = True, []
            assert sep._iterate_with_deferred_loss(2, True)  # warm-up of the loss variants
            sep.loss = []
            torch.cuda.synchronize()
            tl = time.perf_counter()
            ok = sep._iterate_with_deferred_loss(nl, True)  # the loop __call__ runs with record_loss=True
            torch.cuda.synchronize()
            dtl = (time.perf_counter() - tl) / nl
            assert ok and len(sep.loss) == nl + 1
            sep.record_loss, sep.loss = False, None
            out["with_record_loss"] = {
                "workload": "same batch, the separator's record_loss=True loop ({} iterations + the "
                            "initial and final loss): loss of iteration t as a by-product of the basis "
                            "pass of iteration t+1, one dedicated loss pass at the end".format(nl),
                "ms_per_step": round(1e3 * dtl, 4), "iterations_per_s": round(B / dtl, 2),
                "frac": round(3 * pass_bytes / dtl / 1e9 / HBM_PEAK_GBS, 4),
            }
        except Exception as exc:  # an extra leg must never cost the headline line
            out['with_record_loss'] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}
            torch.cuda.empty_cache()

    # ---- the metric's AuxIVA leg: AuxLaplaceIVA (IP1) on the same resident batch, two passes over X
    # per iteration (frame powers, weighted covariance)
    if extra:
        try:
            from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast

            iva = AuxLaplaceIVA(spatial_algorithm="IP", record_loss=False)
            iva._contrast = _device_contrast(iva.contrast_fn, iva.d_contrast_fn)
            iva._bind_input(X)
            iva._reset()
            iva._C()
            for _ in range(3):
                iva.update_once()
            ni = max(5, args.steps)
            dti = time_loop(iva.update_once, ni)
            iva._check_device_errors()
            out["auxiva_ip"] = rate_entry(
                "AuxLaplaceIVA-IP1, same batch ({} x N={} F={} T={}), {} iterations".format(B, N, F, T, ni),
                dti, B, 2 * 16.0 * N * F * T)
            del iva
            # ... and with ISS (the fused sweep kernel: one read and one write of the separated batch)
            iva = AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=False)
            iva._contrast = _device_contrast(iva.contrast_fn, iva.d_contrast_fn)
            iva._bind_input(X)
            iva._reset()
            for _ in range(3):
                iva.update_once()
            dts = time_loop(iva.update_once, ni)
            iva._check_device_errors()
            out["auxiva_iss"] = rate_entry(
                "AuxLaplaceIVA-ISS, same batch ({} x N={} F={} T={}), {} iterations".format(B, N, F, T, ni),
                dts, B, 2 * 16.0 * N * F * T)
            del iva
            torch.cuda.empty_cache()
        except Exception as exc:  # an extra leg must never cost the headline line
            out['auxiva'] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}
            torch.cuda.empty_cache()

    # ---- CPU baseline of the headline: the NumPy oracle (reference expression structure)
    if not args.no_cpu_baseline and n_gpus == 1:
        from oracle.ilrma import GaussILRMAOracle

        def make():
            ref = GaussILRMAOracle(n_basis=K, spatial_algorithm="IP", record_loss=False)
            ref.reset(x0_host, basis=np.random.default_rng(1).random((N, F, K)),
                      activation=np.random.default_rng(2).random((N, K, T)))
            return ref

        out["cpu_baseline"] = cpu_leg(make, args.cpu_iters,
                                      "oracle.ilrma.GaussILRMAOracle.update_once on configs[1]")
        out["speedup_vs_cpu_per_mixture_iteration"] = round(
            value * out["cpu_baseline"]["s_per_iter_median"], 1)

    # ---- the other BASELINE configs, each with its own CPU baseline
    if extra:
        del sep, X
        torch.cuda.empty_cache()
    if extra and not args.only_pairwise:
        try:
            out["configs"] = other_configs(args, dev, x0_host, pins, out.get("cpu_baseline"))
        except Exception as exc:  # an extra leg must never cost the headline line
            out["configs"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}

    # ---- the pairwise and IPA spatial updates (configs[1] / [3] shapes, 1 / 32 / 128 mixtures)
    if extra and Xh is not None and not args.no_pairwise:
        try:
            out["pairwise_ipa"] = pairwise_ipa_legs(args, dev, Xh)
        except Exception as exc:  # an extra leg must never cost the headline line
            out["pairwise_ipa"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}

    # ---- end to end: host NumPy in -> __call__ -> host NumPy out (PCIe inclusive; never `value`)
    if extra and Xh is not None and not args.only_pairwise:
        try:
            cfg = out.get("configs", {}) if isinstance(out.get("configs"), dict) else {}
            cpu = {"configs1": out.get("cpu_baseline"),
                   "configs2": (cfg.get("configs2") or {}).get("cpu_baseline"),
                   "configs3": (cfg.get("configs3") or {}).get("cpu_baseline")}
            out["call_end_to_end"] = end_to_end(args, dev, Xh, pins, cpu)
        except Exception as exc:  # an extra leg must never cost the headline line
            out["call_end_to_end"] = {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}

    print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
