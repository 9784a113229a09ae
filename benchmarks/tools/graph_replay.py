#!/usr/bin/env python3
"""Does replaying update_once() from a captured HIP graph beat issuing its launches from Python?
One mixture of configs[1] (GaussILRMA-IP1), AuxLaplaceIVA-IP / -ISS N=4 F=1025 T=512, configs[2]
(N=8 ISS) and configs[3] (FastGaussMNMF): eager loop vs torch.cuda.CUDAGraph replay of `unroll`
iterations per graph; the states after the same number of iterations are compared bit for bit."""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast
from ssspy_amd.bss.mnmf import FastGaussMNMF
from ssspy_amd.utils.dataset import nmf_mixture

unroll = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = 30


def prepare(make, X):
    m = make()
    if isinstance(m, AuxLaplaceIVA):
        m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
    m._bind_input(X)
    m._reset(flooring_fn=m.flooring_fn) if not isinstance(m, AuxLaplaceIVA) else m._reset()
    if hasattr(m, "_C"):
        m._C()
    for _ in range(5):
        m.update_once()
    torch.cuda.synchronize()
    return m


def run(tag, make, X, names):
    a = prepare(make, X)
    gc.collect()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps * unroll):
        a.update_once()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / (reps * unroll)
    b = prepare(make, X)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=side):
                for _ in range(unroll):
                    b.update_once()
        except Exception as e:  # noqa: BLE001
            print("%-34s capture failed: %s" % (tag, str(e).splitlines()[0][:120]))
            return
        torch.cuda.synchronize()
        # capture does not execute: the state is still the one after the warm-up
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / (reps * unroll)
    same = all(np.array_equal(np.asarray(getattr(a, n)), np.asarray(getattr(b, n))) for n in names)
    print("%-34s eager %7.1f us/iter | graph (x%d) %7.1f us/iter | %+.1f %% | states equal: %s"
          % (tag, 1e6 * eager, unroll, 1e6 * graph, 100 * (eager / graph - 1), same))


X4 = nmf_mixture(1000, 4, 1025, 512)
run("configs[1] GaussILRMA-IP1", lambda: GaussILRMA(n_basis=16, record_loss=False,
                                                     rng=np.random.default_rng(0)), X4,
    ("basis", "activation", "demix_filter"))
run("AuxLaplaceIVA-IP N=4", lambda: AuxLaplaceIVA(spatial_algorithm="IP", record_loss=False), X4,
    ("demix_filter",))
run("AuxLaplaceIVA-ISS N=4", lambda: AuxLaplaceIVA(spatial_algorithm="ISS", record_loss=False), X4,
    ("output",))
run("configs[3] FastGaussMNMF-IP1", lambda: FastGaussMNMF(n_basis=8, record_loss=False,
                                                          rng=np.random.default_rng(0)), X4,
    ("basis", "activation", "diagonalizer", "spatial"))
X8 = nmf_mixture(3000, 8, 2049, 1024)
run("configs[2] AuxLaplaceIVA-ISS N=8", lambda: AuxLaplaceIVA(spatial_algorithm="ISS",
                                                               record_loss=False), X8, ("output",))
