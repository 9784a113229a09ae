#!/usr/bin/env python3
"""Phases of one separator(X, n_iter) with a device synchronisation at the phase boundaries:
`call_phases.py auxiva|fmnmf [n_iter]` (configs[2] AuxLaplaceIVA-ISS, configs[3] FastGaussMNMF;
call_timeline.py is the GaussILRMA form)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd import _ops  # noqa: E402
from ssspy_amd.bss.base import IterativeMethodBase  # noqa: E402
from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast  # noqa: E402
from ssspy_amd.bss.mnmf import FastGaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

kind = sys.argv[1]
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if kind == "auxiva":
    X = nmf_mixture(1000, 8, 2049, 1024)
    make = lambda: AuxLaplaceIVA(spatial_algorithm="ISS")  # noqa: E731
else:
    X = nmf_mixture(1000, 4, 1025, 512)
    make = lambda: FastGaussMNMF(n_basis=8, rng=np.random.default_rng(0))  # noqa: E731
make()(X, n_iter=2)
marks = []


def mark(name):
    torch.cuda.synchronize()
    marks.append((name, time.perf_counter()))


m = make()
mark("start")
if kind == "auxiva":
    m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
m._bind_input(X)
mark("_bind_input (upload X, %.0f MB)" % (X.nbytes / 1e6))
m._reset()
mark("_reset")
if not m._iterate_with_resident_loss(n_iter, True):
    IterativeMethodBase.__call__(m, n_iter=n_iter, initial_call=True)
mark("loop (%d iterations)" % n_iter)
if kind == "auxiva":
    if m.scale_restoration:
        m.restore_scale()
    mark("restore_scale")
    if m._uses_filter():
        m._state_set_dev("output", _ops.separate(m._X, m._state_dev("demix_filter")))
else:
    m._separate_dev()
mark("separate")
Y = m._final_output()
mark("_final_output (download %.0f MB)" % (Y.nbytes / 1e6))
for (a, ta), (b, tb) in zip(marks, marks[1:]):
    print("  %-44s %8.3f ms" % (b, 1e3 * (tb - ta)))
print("  %-44s %8.3f ms" % ("sum", 1e3 * (marks[-1][1] - marks[0][1])))
