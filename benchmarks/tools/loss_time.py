import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from ssspy_amd.bss.ilrma import GaussILRMA
from ssspy_amd.utils.dataset import nmf_mixture
B, N, F, T = 32, 4, 1025, 512
X = torch.from_numpy(np.stack([nmf_mixture(1000, N, F, T)] * B)).cuda()
for K in (16, 32, 64, 128):
    m = GaussILRMA(n_basis=K, record_loss=False, rng=np.random.default_rng(0))
    m._bind_input(X); m._reset(flooring_fn=m.flooring_fn); m._C()
    for _ in range(2): m.update_once(); m.compute_loss()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.compute_loss()
    torch.cuda.synchronize()
    print("n_basis %3d: compute_loss %.3f ms" % (K, 1e3 * (time.perf_counter() - t0) / 5))
