"""Worker for tests/test_parallel_gloo.py: launched by torch.distributed.run with 2 CPU processes."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ilrma import GaussILRMAOracle  # noqa: E402  (the checker stands in for the device path)
from ssspy_amd import parallel  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402

N_MIX, N, F, T, K = 5, 2, 9, 16, 2


def process(lo, hi):
    out = []
    for b in range(lo, hi):
        m = GaussILRMAOracle(n_basis=K, record_loss=False)
        m.run(nmf_mixture(1000 + b, N, F, T), n_iter=2,
              basis=np.random.default_rng(b).random((N, F, K)),
              activation=np.random.default_rng(100 + b).random((N, K, T)))
        out.append(m.demix_filter)
    return np.stack(out) if out else np.zeros((0, F, N, N), dtype=complex)


def main():
    rank, world, _ = parallel.init_from_env(backend="gloo")
    lo, hi = parallel.shard_bounds(N_MIX, rank, world)
    full = parallel.run_sharded(process, N_MIX, gather=True)
    local = parallel.run_sharded(process, N_MIX, gather=False)
    slowest = parallel.max_over_ranks(1.0 + rank)
    parallel.barrier()
    if rank == 0:
        np.savez(sys.argv[1], full=full, local=local, lo=lo, hi=hi, world=world, slowest=slowest)


if __name__ == "__main__":
    main()
