"""Worker for tests/test_gpu_paths_agree.py::test_two_ranks_real_separators_share_one_device: two
gloo ranks, both on HIP device 0, each running the REAL GaussILRMA separator (the HIP library) on its
block of a 5-mixture batch through ssspy_amd.parallel.run_sharded."""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ssspy_amd import parallel  # noqa: E402
from ssspy_amd.bss.ilrma import GaussILRMA  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

N_MIX, N, F, T, K, N_ITER = 5, 3, 33, 48, 4, 4


def inputs(lo, hi):
    X = np.stack([nmf_mixture(1000 + b, N, F, T) for b in range(lo, hi)])
    basis = np.stack([np.random.default_rng(b).random((N, F, K)) for b in range(lo, hi)])
    act = np.stack([np.random.default_rng(100 + b).random((N, K, T)) for b in range(lo, hi)])
    return X, basis, act


def process(lo, hi):
    """Filters and loss lists of mixtures [lo, hi): ONE batched call of the real separator."""
    X, basis, act = inputs(lo, hi)
    m = GaussILRMA(n_basis=K)
    m(X, n_iter=N_ITER, basis=basis, activation=act)
    loss = np.asarray(m.loss).reshape(N_ITER + 1, -1).T  # (mixture, iteration)
    return np.concatenate([m.demix_filter.reshape(hi - lo, -1),
                           loss.astype(np.complex128)], axis=1)


def main():
    torch.cuda.set_device(0)  # both ranks share the one device of the test box
    rank, world, _ = parallel.init_from_env(backend="gloo")
    full = parallel.run_sharded(process, N_MIX, gather=True)
    slowest = parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
    parallel.barrier()
    if rank == 0:
        np.savez(sys.argv[1], full=full, world=world, slowest=slowest)


if __name__ == "__main__":
    main()
