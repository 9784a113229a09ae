#!/usr/bin/env python3
"""Randomised differential run of the ssspy.linalg / to_psd operators against LAPACK (NumPy / SciPy)
and of lqpqm2 against the oracle, sizes 1 .. 16 (a development tool; the fixed cases live in tests/).

    python benchmarks/fuzz_operators.py [n_cases] [seed]
"""
import functools
import os
import sys
import warnings

import numpy as np
import scipy.linalg

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import oracle.spatial as osp  # noqa: E402
from oracle.ipa import lqpqm2 as oracle_lqpqm2  # noqa: E402
from ssspy_amd.algorithm import minimal_distortion_principle, projection_back  # noqa: E402
from ssspy_amd.linalg import eigh, gmeanmh, invsqrtmh, lqpqm2, solve, sqrtmh  # noqa: E402
from ssspy_amd.special.flooring import add_flooring, max_flooring  # noqa: E402
from ssspy_amd.special.psd import to_psd  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def psd(rng, n, M, T, cplx):
    x = rng.standard_normal((n, M, T))
    if cplx:
        x = x + 1j * rng.standard_normal((n, M, T))
    return np.mean(x[:, :, None, :] * x[:, None, :, :].conj(), axis=-1)


def projectors(z):
    return z[..., :, None, :] * z[..., None, :, :].conj()


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n_cases):
        M = int(rng.integers(1, 17))
        n = int(rng.choice([1, 2, 63, 64, 65, 130, 300]))
        cplx = bool(rng.random() < 0.7)
        T = M + int(rng.integers(2, 3 * M + 4))
        op = str(rng.choice(["solve", "eigh", "to_psd", "sqrt", "invsqrt", "gmean", "geigh", "lqpqm2",
                             "restore"]))
        tag = (case, op, M, n, cplx, T)
        errs = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            A, B = psd(rng, n, M, T, cplx), psd(rng, n, M, T, cplx)
            if op == "solve":
                G = rng.standard_normal((n, M, M)) + (1j * rng.standard_normal((n, M, M)) if cplx else 0)
                nrhs = int(rng.integers(1, 5))
                b = rng.standard_normal((n, M, nrhs)) + (1j * rng.standard_normal((n, M, nrhs)) if cplx else 0)
                errs["x"] = (rel(solve(G, b), np.linalg.solve(G, b)), 1e-8)
            elif op == "eigh":
                H = A - 0.3 * np.eye(M)
                lam, V = eigh(H)
                lam0, V0 = np.linalg.eigh(H)
                errs["lam"] = (float(np.max(np.abs(lam - lam0)) / np.max(np.abs(lam0))), 1e-11)
                errs["res"] = (rel(H @ V, V * lam[:, None, :]), 1e-11)
                errs["unit"] = (rel(V.swapaxes(-2, -1).conj() @ V, np.broadcast_to(np.eye(M), V.shape)), 1e-11)
            elif op == "to_psd":
                H = A - float(rng.random()) * np.eye(M)
                eps = float(rng.choice([1e-10, 0.05, 0.5]))
                fl = str(rng.choice(["max", "add"]))
                fn = functools.partial(max_flooring if fl == "max" else add_flooring, eps=eps)
                lam0, V0 = np.linalg.eigh((H + H.swapaxes(-2, -1).conj()) / 2)
                ref = (V0 * fn(lam0)[:, None, :]) @ V0.swapaxes(-2, -1).conj()
                errs["psd"] = (rel(to_psd(H, flooring_fn=fn), ref), 1e-10)
            elif op == "sqrt":
                S = sqrtmh(A)
                errs["sq"] = (rel(S @ S, A), 1e-10)
            elif op == "invsqrt":
                eps = float(rng.choice([1e-10, 0.3]))
                lam0, V0 = np.linalg.eigh(A)
                ref = (V0 / np.maximum(np.sqrt(lam0), eps)[:, None, :]) @ V0.swapaxes(-2, -1).conj()
                errs["isq"] = (rel(invsqrtmh(A, flooring_fn=functools.partial(max_flooring, eps=eps)), ref), 1e-8)
            elif op == "gmean":
                t = int(rng.integers(1, 4))
                G = gmeanmh(A, B, type=t)
                Ai, Bi = np.linalg.inv(A), np.linalg.inv(B)
                lhs, rhs = {1: (G @ Ai @ G, B), 2: (G @ A @ G, B), 3: (G @ Ai @ G, Bi)}[t]
                errs["ric%d" % t] = (rel(lhs, rhs), 1e-6)
            elif op == "geigh":
                t = int(rng.integers(1, 4))
                lam, z = eigh(A, B, type=t)
                i = int(rng.integers(n))
                lam0 = scipy.linalg.eigh(A[i], B[i], type=t, eigvals_only=True)
                errs["lam%d" % t] = (float(np.max(np.abs(lam[i] - lam0) / np.abs(lam0))), 1e-8)
                lhs, rhs = {1: (A @ z, lam[:, None, :] * (B @ z)), 2: (A @ B @ z, lam[:, None, :] * z),
                            3: (B @ A @ z, lam[:, None, :] * z)}[t]
                errs["res%d" % t] = (rel(lhs, rhs), 1e-8)
            elif op == "restore":
                # projection back (filter and spectrogram form) and the minimal distortion principle,
                # one reference channel or all of them (reference_id=None)
                N, F, Tn = M, int(rng.choice([1, 3, 17, 33])), M + int(rng.integers(3, 40))
                Y = rng.standard_normal((N, F, Tn)) + 1j * rng.standard_normal((N, F, Tn))
                Xm = rng.standard_normal((N, F, Tn)) + 1j * rng.standard_normal((N, F, Tn))
                W = np.eye(N) + 0.3 * (rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N)))
                rid = [0, N - 1, int(rng.integers(N)), None][int(rng.integers(4))]
                errs["pb_w"] = (rel(projection_back(W, reference_id=rid),
                                    osp.projection_back_filter(W, rid)), 1e-9)
                errs["pb_y"] = (rel(projection_back(Y, reference=Xm, reference_id=rid),
                                    osp.projection_back_output(Y, Xm, rid)), 1e-8)
                errs["mdp"] = (rel(minimal_distortion_principle(Y, reference=Xm, reference_id=rid),
                                   osp.minimal_distortion_output(Y, Xm, rid)), 1e-8)
            else:
                L = min(M, 15)
                H = psd(rng, n, L, T, True)
                H = H / np.real(np.trace(H, axis1=-2, axis2=-1))[:, None, None]
                v = rng.standard_normal((n, L)) + 1j * rng.standard_normal((n, L))
                z = rng.random(n) * 2.0
                it = int(rng.choice([1, 3, 10]))
                errs["y"] = (rel(lqpqm2(H, v, z, max_iter=it),
                                 np.asarray(oracle_lqpqm2(H, v, z, ("max", 1e-10), it))), 1e-8)
        for k, (e, tol) in errs.items():
            if not e < tol:
                bad += 1
                print("MISMATCH", tag, k, e)
        if os.environ.get("FUZZ_VERBOSE"):
            print("case", tag, {k: "%.1e" % e for k, (e, _) in errs.items()})
    print("cases", n_cases, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
