"""N > 1 path on CPU: 2 gloo processes shard 5 mixtures (3 + 2), no data-path collective; the gathered
result must equal the serial one.  The per-shard compute is the CPU oracle (test stand-in): what is
under test is the sharding, ordering, gather and the max-over-ranks timing of ssspy_amd.parallel."""

import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from ssspy_amd import parallel


def test_shard_bounds_cover_and_balance():
    for n, world in [(1024, 8), (5, 2), (3, 4), (0, 2), (7, 7)]:
        blocks = [parallel.shard_bounds(n, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_bounds(1024, 3, 8) == (384, 512)


def test_single_process_passthrough():
    out = parallel.run_sharded(lambda lo, hi: np.arange(lo, hi), 6)
    assert out.tolist() == [0, 1, 2, 3, 4, 5]
    assert parallel.max_over_ranks(0.25) == 0.25


def test_two_process_gloo_sharding(tmp_path):
    out = tmp_path / "result.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(out)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.load(out)
    assert int(got["world"]) == 2 and (int(got["lo"]), int(got["hi"])) == (0, 3)
    assert float(got["slowest"]) == 2.0

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _gloo_worker as w

    serial = w.process(0, w.N_MIX)
    assert got["full"].shape == serial.shape
    assert np.array_equal(got["full"], serial)       # same code per mixture: bit-identical
    assert np.array_equal(got["local"], serial[0:3])
