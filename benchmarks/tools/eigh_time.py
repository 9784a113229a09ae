#!/usr/bin/env python3
"""linalg.eigh / special.to_psd on n Hermitian M x M matrices: ms per call, check against NumPy."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ssspy_amd import _lib, _device as dv
from ssspy_amd._device import ptr


def eigh_dev(Ad):
    n, M, _ = Ad.shape
    lamb = dv.empty((n, M), dv.f64, Ad.device)
    V = dv.empty((n, M, M), dv.c128, Ad.device)
    _lib.check(_lib.load().ssspy_eigh(ptr(Ad), ptr(lamb), ptr(V), n, M, dv.stream_handle()), 'eigh')
    return lamb, V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
for M in ([int(a) for a in sys.argv[2:]] or (4, 5, 6, 7, 8)):
    A = rng.standard_normal((n, M, M)) + 1j * rng.standard_normal((n, M, M))
    A = A @ A.conj().transpose(0, 2, 1) + 0.1 * np.eye(M)
    Ad = dv.to_device(A)
    lam, V = eigh_dev(Ad)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        lam, V = eigh_dev(Ad)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 5
    lam_h, V_h = dv.to_host(lam), dv.to_host(V)
    ref = np.linalg.eigvalsh(A[:2000])
    err = np.max(np.abs(lam_h[:2000] - ref) / np.max(np.abs(ref)))
    rec = np.max(np.abs((V_h[:2000] * lam_h[:2000, None, :]) @ V_h[:2000].conj().transpose(0, 2, 1) - A[:2000]))
    print("M=%d  n=%d  eigh %.3f ms  (%.1f ns per matrix)  eigenvalue err %.1e  rebuild err %.1e"
          % (M, n, ms, 1e6 * ms / n, err, rec))
