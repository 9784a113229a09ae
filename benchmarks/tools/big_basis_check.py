#!/usr/bin/env python3
"""n_basis far above the tuned range against the oracle (tiny shapes): `big_basis_check.py [K ...]`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.gmnmf import GaussMNMFOracle  # noqa: E402
from oracle.ilrma import GaussILRMAOracle  # noqa: E402
from oracle.mnmf import FastGaussMNMFOracle  # noqa: E402
from ssspy_amd.bss.ilrma import GaussILRMA  # noqa: E402
from ssspy_amd.bss.mnmf import FastGaussMNMF, GaussMNMF  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


only = os.environ.get("ONLY", "")
for K in [int(a) for a in sys.argv[1:]] or [1500, 3000]:
    for N, F, T in ((2, 20, 30), (3, 33, 17)):
        X = nmf_mixture(5, N, F, T)
        basis = np.random.default_rng(1).random((N, F, K))
        act = np.random.default_rng(2).random((N, K, T))
        if not only or only == "ilrma":
            for algo in ("IP", "ISS"):
                ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
                Yr = ref.run(X, n_iter=3, basis=basis, activation=act)
                m = GaussILRMA(n_basis=K, spatial_algorithm=algo)
                Y = m(X, n_iter=3, basis=basis, activation=act)
                print("ILRMA", algo, K, N, F, T, rel(Y, Yr),
                      np.max(np.abs(np.array(m.loss) / np.array(ref.loss) - 1)))
        if not only or only == "fmnmf":
            sp0 = np.random.default_rng(3).random((F, N, N)) + 0.05
            ref = FastGaussMNMFOracle(n_basis=K)
            Yr = ref.run(X, n_iter=3, basis=basis, activation=act, spatial=sp0.copy())
            m = FastGaussMNMF(n_basis=K)
            Y = m(X, n_iter=3, basis=basis, activation=act, spatial=sp0)
            print("FastMNMF", K, N, F, T, rel(Y, Yr), rel(m.basis, ref.basis),
                  np.max(np.abs(np.array(m.loss) / np.array(ref.loss) - 1)))
        if not only or only == "gmnmf":
            ref = GaussMNMFOracle(n_basis=K)
            Yr = ref.run(X, n_iter=2, basis=basis, activation=act)
            m = GaussMNMF(n_basis=K)
            Y = m(X, n_iter=2, basis=basis, activation=act)
            print("GaussMNMF", K, N, F, T, rel(Y, Yr),
                  np.max(np.abs(np.array(m.loss) / np.array(ref.loss) - 1)))
