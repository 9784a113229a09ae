"""Oracle runs for tests/test_gpu_benchmark_sizes.py in their own processes.

The checker is plain NumPy on the host: a hundred iterations of the ISS / ISS2 / IPA oracles at the
configs[1] shape take one to three minutes each, so the tests that need them start them all at once
(spawned processes: nothing of the HIP runtime of the test process is inherited) and collect the
results while the device runs.  Test infrastructure only.
"""

import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N, F, T, K = 4, 1025, 512, 16


def initial(seed):
    """The explicit NMF state of a configs[1]-shaped mixture with the given seed."""
    return (np.random.default_rng(seed + 1).random((N, F, K)),
            np.random.default_rng(seed + 2).random((N, K, T)))


def run(family, algo, seed, n_iter):
    """(loss list, output after scale restoration) of the oracle on nmf_mixture(seed, 4, 1025, 512)."""
    from oracle.ilrma import GaussILRMAOracle
    from oracle.iva import AuxIVAOracle
    from ssspy_amd.utils.dataset import nmf_mixture

    X = nmf_mixture(seed, N, F, T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (IPA: "Newton-Raphson method did not converge in 1 iterations")
        if family == "ilrma":
            basis, act = initial(seed)
            ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
            Y = ref.run(X, n_iter=n_iter, basis=basis, activation=act)
        else:
            ref = AuxIVAOracle(spatial_algorithm=algo, contrast="laplace")
            Y = ref.run(X, n_iter=n_iter)
    return np.asarray(ref.loss), Y
