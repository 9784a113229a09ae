#!/usr/bin/env python3
"""Timeline of one ``GaussILRMA(n_basis=16)(X, n_iter=100)`` on configs[1] (one mixture, host NumPy in
and out, record_loss=True, projection back): wall time of every phase of ``__call__`` with a device
synchronisation at the phase boundaries (so the phases add up to a serialised call; the unsynchronised
call is printed beside it), plus the finer split of ``_reset`` and of the tail.

    python benchmarks/tools/call_timeline.py [n_iter]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd import _ops  # noqa: E402
from ssspy_amd.bss import ilrma as ilrma_mod  # noqa: E402
from ssspy_amd.bss.base import IterativeMethodBase  # noqa: E402
from ssspy_amd.bss.ilrma import GaussILRMA  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
X = nmf_mixture(1000, 4, 1025, 512)


def make():
    return GaussILRMA(n_basis=16, rng=np.random.default_rng(0))


make()(X, n_iter=2)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    Y = make()(X, n_iter=n_iter)
    print("plain __call__: %.2f ms" % (1e3 * (time.perf_counter() - t0)))

marks = []


def mark(name):
    torch.cuda.synchronize()
    marks.append((name, time.perf_counter()))


m = make()
mark("start")
m._bind_input(X)
mark("_bind_input (upload X)")
m._reset(flooring_fn=m.flooring_fn)
mark("_reset")
if not m._iterate_with_deferred_loss(n_iter, True):
    IterativeMethodBase.__call__(m, n_iter=n_iter, initial_call=True)
mark("loop (%d iterations, deferred loss)" % n_iter)
m.restore_scale()
mark("restore_scale")
m._state_set_dev("output", _ops.separate(m._X, m._state_dev("demix_filter")))
mark("separate")
Yt = m._final_output()
mark("_final_output (download)")
for (a, ta), (b, tb) in zip(marks, marks[1:]):
    print("  %-40s %8.3f ms" % (b, 1e3 * (tb - ta)))
print("  %-40s %8.3f ms" % ("sum", 1e3 * (marks[-1][1] - marks[0][1])))

# finer: host-side pieces of _reset
import cProfile
import pstats

m2 = make()
m2._bind_input(X)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
m2._reset(flooring_fn=m2.flooring_fn)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)
m3 = make()
m3._bind_input(X)
m3._reset(flooring_fn=m3.flooring_fn)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
m3._iterate_with_deferred_loss(n_iter, True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
