import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ssspy_amd.bss.iva import AuxLaplaceIVA
from ssspy_amd.utils.dataset import nmf_mixture
for shape in ((2, 257, 128), (4, 1025, 512)):
    X = nmf_mixture(1000, *shape)
    AuxLaplaceIVA()(X, n_iter=2)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter(); AuxLaplaceIVA()(X, n_iter=100); dt = time.perf_counter() - t0
        print("AuxLaplaceIVA-IP1 %s __call__ 100 iterations: %.2f ms" % (shape, 1e3 * dt))
    m = AuxLaplaceIVA(record_loss=False); m._bind_input(X); m._reset()
    from ssspy_amd.bss.iva import _device_contrast
    m._contrast = _device_contrast(m.contrast_fn, m.d_contrast_fn)
    for _ in range(5): m.update_once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): m.update_once()
    torch.cuda.synchronize(); print("  update_once: %.1f us" % (1e6 * (time.perf_counter() - t0) / 300))
