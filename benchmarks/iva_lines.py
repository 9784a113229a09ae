#!/usr/bin/env python3
"""AuxLaplaceIVA lines of bench.py in isolation: N=4 F=1025 T=512 (IP / ISS, 1 and 128 mixtures),
configs[2] N=8 F=2049 T=1024 (ISS, 1 and 32 mixtures).  ms per update_once()."""
import gc
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssspy_amd.bss.iva import AuxLaplaceIVA, _device_contrast
from ssspy_amd.utils.dataset import nmf_mixture_batch

def run(X, algo, iters):
    m = AuxLaplaceIVA(spatial_algorithm=algo, record_loss=False)
    m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
    m._bind_input(X); m._reset()
    if algo == "IP": m._C()
    for _ in range(5): m.update_once()
    gc.collect(); gc.freeze()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): m.update_once()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters

dev = torch.device("cuda", 0)
X4 = torch.from_numpy(nmf_mixture_batch(1000, 128, 4, 1025, 512)).to(dev)
for algo in ("IP", "ISS"):
    print("N=4 %-3s   1 mixture : %.4f ms" % (algo, run(X4[:1].contiguous(), algo, 200)))
    print("N=4 %-3s 128 mixtures: %.4f ms" % (algo, run(X4, algo, 10)))
del X4; torch.cuda.empty_cache()
X8 = torch.from_numpy(nmf_mixture_batch(3000, 32, 8, 2049, 1024)).to(dev)
print("N=8 ISS   1 mixture : %.4f ms" % run(X8[:1].contiguous(), "ISS", 100))
print("N=8 ISS  32 mixtures: %.4f ms" % run(X8, "ISS", 10))
