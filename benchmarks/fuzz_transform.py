#!/usr/bin/env python3
"""Randomised STFT / ISTFT lengths, hops, windows and signal lengths against scipy.signal
(a development tool; the fixed cases live in tests/).  python benchmarks/fuzz_transform.py [n] [seed]"""
import os
import sys
import warnings

import numpy as np
import scipy.signal as ss

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ssspy_amd.transform import istft, stft  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n_cases):
        kind = rng.random()
        if kind < 0.3:
            n_fft = int(2 ** rng.integers(1, 15))
        elif kind < 0.8:
            n_fft = int(rng.integers(2, 4200))
        else:
            n_fft = int(rng.integers(4200, 20000))
        hop = int(rng.integers(1, n_fft + 1)) if rng.random() < 0.5 else max(1, n_fft // int(rng.choice([2, 3, 4, 8])))
        window = [("hann"), ("hamming"), ("blackman"), ("boxcar"), ("kaiser", 8.6), ("tukey", 0.5)][int(rng.integers(6))]
        L = int(rng.integers(max(1, n_fft // 2), 6 * n_fft + 50))
        C = int(rng.integers(1, 4))
        tag = (case, n_fft, hop, window, L, C)
        x = rng.standard_normal((C, L))
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    _, _, Zr = ss.stft(x, window=window, nperseg=n_fft, noverlap=n_fft - hop)
                except ValueError as ref_exc:  # (a signal shorter than the overlap): the same refusal
                    try:
                        stft(x, n_fft=n_fft, hop_length=hop, window=window)
                        raise AssertionError("SciPy raised, stft did not: " + str(ref_exc))
                    except ValueError as exc:
                        assert str(exc) == str(ref_exc), (str(exc), str(ref_exc))
                    continue
                Z = stft(x, n_fft=n_fft, hop_length=hop, window=window)
                n_fft, hop = min(n_fft, L), min(n_fft, L) - (n_fft - hop)  # (what SciPy used)
                ok = Z.shape == Zr.shape and rel(Z, Zr) < 1e-10
                try:
                    _, yr = ss.istft(Zr, window=window, nperseg=n_fft, noverlap=n_fft - hop)
                except ValueError:  # (SciPy refuses windows that fail NOLA)
                    yr = None
                if yr is not None:
                    y = istft(Zr, n_fft=n_fft, hop_length=hop, window=window)
                    ok = ok and y.shape == yr.shape and rel(y, yr) < 1e-10
            if not ok:
                bad += 1
                print("MISMATCH", tag, Z.shape, Zr.shape)
        except Exception as exc:
            bad += 1
            print("EXC", tag, type(exc).__name__, str(exc)[:120])
    print("cases", n_cases, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
