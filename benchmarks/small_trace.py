#!/usr/bin/env python3
"""Per-wave phase timeline of k_activation_small (build with SSSPY_AMD_EXTRA_CXXFLAGS=-DSSSPY_SMALL_TRACE).
Runs a few single-mixture iterations, then reads the stamps of the LAST activation launch."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd import _lib
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.utils.dataset import nmf_mixture

X = nmf_mixture(1000, 4, 1025, 512)
sep = GaussILRMA(n_basis=16, record_loss=False, rng=np.random.default_rng(0))
sep._bind_input(X); sep._reset(flooring_fn=sep.flooring_fn); sep._C()
for _ in range(20): sep.update_once()
torch.cuda.synchronize()
lib = _lib.load()
n = 8 * 4096
buf = (ctypes.c_longlong * n)()
rc = lib.ssspy_debug_small_trace(buf, n)
a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 8)
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
rel = (a - t0).astype(float)
names = ["entry", "V staged+sync", "x/T/W landed+parked", "|y|^2 done", "GEMMs done", "WG sync", "end"]
print("waves", len(a), "(cycles of the shader clock; ~2.1-2.4 GHz)")
for k, nm in enumerate(names):
    col = rel[:, k]
    print("{:22s} mean {:9.0f}  min {:9.0f}  max {:9.0f}".format(nm, col.mean(), col.min(), col.max()))
d = np.diff(rel[:, :7], axis=1)
print("phase durations (mean):", np.round(d.mean(axis=0)).astype(int).tolist())
print("phase durations (p90): ", np.round(np.percentile(d, 90, axis=0)).astype(int).tolist())
print("kernel span (first entry -> last end):", rel[:, 6].max())
