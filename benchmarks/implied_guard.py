#!/usr/bin/env python3
"""How the implied-filter route's rounding bound (kappa of ssspy_covariance_congruence_tracked)
relates to its distance from the oracle: short, badly conditioned draws (few frames per source)
with the guard at its limit, off, and the on-Y route.  A development tool (round 6).

    python benchmarks/implied_guard.py [n_seeds]
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle.ilrma import GaussILRMAOracle  # noqa: E402
from oracle.iva import AuxIVAOracle  # noqa: E402
from ssspy_amd import _routes  # noqa: E402
from ssspy_amd.bss.ilrma import GaussILRMA  # noqa: E402
from ssspy_amd.bss.iva import AuxLaplaceIVA  # noqa: E402
from ssspy_amd.utils.dataset import nmf_mixture  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    warnings.simplefilter("ignore")
    print("family algo N F T seed | err guarded (left route?) | err unguarded  kappa | err on-Y")
    for family, algo, N, F, T, n_iter in [("ilrma", "ISS2", 4, 31, 11, 8), ("ilrma", "IPA", 4, 31, 11, 8),
                                          ("ilrma", "ISS2", 4, 31, 16, 8), ("ilrma", "ISS2", 4, 31, 64, 8),
                                          ("iva", "ISS2", 4, 31, 11, 8), ("iva", "IPA", 4, 31, 11, 8),
                                          ("iva", "ISS2", 4, 31, 64, 8), ("ilrma", "ISS2", 3, 33, 8, 12),
                                          ("ilrma", "ISS2", 4, 1025, 512, 8)] if n_seeds else []:
        for seed in range(n_seeds if F < 1000 else 1):
            X = nmf_mixture(7000 + seed, N, F, T)
            K = 8
            rng = np.random.default_rng(seed)
            kw = dict(basis=rng.random((N, F, K)), activation=rng.random((N, K, T))) if family == "ilrma" else {}

            def make():
                if family == "ilrma":
                    return GaussILRMA(n_basis=K, spatial_algorithm=algo)
                return AuxLaplaceIVA(spatial_algorithm=algo)

            if family == "ilrma":
                ref = GaussILRMAOracle(n_basis=K, spatial_algorithm=algo)
            else:
                ref = AuxIVAOracle(spatial_algorithm=algo, contrast="laplace")
            try:
                Yr = ref.run(X, n_iter=n_iter, **{k: v.copy() for k, v in kw.items()})
                m = make()
                Yg = m(X, n_iter=n_iter, **{k: v.copy() for k, v in kw.items()})
                left = m._implied is None
                m2 = make()
                m2._implied_amp_limit = float("inf")
                Yu = m2(X, n_iter=n_iter, **{k: v.copy() for k, v in kw.items()})
                import torch
                torch.cuda.synchronize()
                kappa = m2._amp_kappa_rms()
                m3 = make()
                m3._implied_amp_limit = 0.0  # leaves at the second iteration
                with _routes.override(implied_filter=False):
                    Yy = m3(X, n_iter=n_iter, **{k: v.copy() for k, v in kw.items()})
                print(family, algo, N, F, T, seed, "| %.1e %s | %.1e %.1e | %.1e" % (
                    rel(Yg, Yr), left, rel(Yu, Yr), kappa, rel(Yy, Yr)), flush=True)
            except Exception as exc:
                print(family, algo, N, F, T, seed, "EXC", type(exc).__name__, str(exc)[:80], flush=True)
    # the benchmark shape, 100 iterations: implied route (guard off) against the on-Y route
    import torch
    for family, algo in [("ilrma", "ISS1"), ("ilrma", "ISS2"), ("ilrma", "IPA"), ("iva", "ISS2"), ("iva", "IPA")]:
        N, F, T, K = 4, 1025, 512, 16
        for seed in (1000, 1017):
            X = nmf_mixture(seed, N, F, T)
            kw = dict(basis=np.random.default_rng(seed + 1).random((N, F, K)),
                      activation=np.random.default_rng(seed + 2).random((N, K, T))) if family == "ilrma" else {}
            res = []
            for implied in (True, False):
                m = (GaussILRMA(n_basis=K, spatial_algorithm=algo) if family == "ilrma"
                     else AuxLaplaceIVA(spatial_algorithm=algo))
                m._implied_amp_limit = float("inf")
                trace = []
                m.callbacks = [lambda mm: trace.append(mm._amp_kappa_rms())]
                with _routes.override(iss1_statistics=True, implied_filter=implied):
                    Y = m(X, n_iter=100, **{k: v.copy() for k, v in kw.items()})
                torch.cuda.synchronize()
                res.append((Y, np.asarray(m.loss), m._amp_kappa_rms(), trace, m._implied_iterations()))
            print("full-size 100 it", family, algo, seed, "implied iterations", res[0][4], "| implied vs on-Y: Y %.1e loss %.1e | kappa_rms %.2e | trace %s" % (
                rel(res[0][0], res[1][0]), float(np.max(np.abs(res[0][1] / res[1][1] - 1))), res[0][2],
                " ".join("%.1e" % t for t in res[0][3][::10])), flush=True)


if __name__ == "__main__":
    sys.exit(main())
