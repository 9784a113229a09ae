"""Seeded synthetic mixtures (the reference downloads speech; there is no network here).

``iid_mixture`` and ``nmf_mixture`` are the generators of SURVEY.md section 8d.  ``nmf_mixture``
is written so that the same seed gives the same BYTES on every host: the two small contractions
(rank-4 variance model, N x N mixing) are accumulated term by term with separate real multiplies
and adds, never through BLAS / ``**`` / complex ufuncs, whose rounding depends on the CPU's SIMD
dispatch.  ``tests/golden/input_sha256.json`` holds the SHA-256 of the benchmark inputs and
``bench.py`` / the tests verify it on the box they run on.
"""

import concurrent.futures
import hashlib

import numpy as np


def iid_mixture(seed, n_channels, n_bins, n_frames):
    """i.i.d. circular complex Gaussian tensor (n_channels, n_bins, n_frames)."""
    rng = np.random.default_rng(seed)
    shape = (n_channels, n_bins, n_frames)
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def nmf_mixture(seed, n_sources, n_bins, n_frames, n_latent=4):
    """Low-rank-variance sources through a random per-bin mixing matrix:
    S_n = sqrt(R_n / 2) (g1 + 1j g2),  R_n = (U^4)(V^4) + 1e-3,  X_i = A_i S_i."""
    rng = np.random.default_rng(seed)
    N, F, T = n_sources, n_bins, n_frames
    u = rng.random((N, F, n_latent))
    v = rng.random((N, n_latent, T))
    u = (u * u) * (u * u)
    v = (v * v) * (v * v)
    R = np.zeros((N, F, T))
    for k in range(n_latent):
        R += u[:, :, k, None] * v[:, None, k, :]
    R += 1e-3
    g1 = rng.standard_normal((N, F, T))
    g2 = rng.standard_normal((N, F, T))
    amp = np.sqrt(R / 2)
    Sr, Si = amp * g1, amp * g2
    Ar = rng.standard_normal((F, N, N))
    Ai = rng.standard_normal((F, N, N))
    X = np.empty((N, F, T), dtype=np.complex128)
    for m in range(N):
        xr = np.zeros((F, T))
        xi = np.zeros((F, T))
        for n in range(N):
            ar, ai = Ar[:, m, n, None], Ai[:, m, n, None]
            xr += ar * Sr[n] - ai * Si[n]
            xi += ar * Si[n] + ai * Sr[n]
        X[m].real = xr
        X[m].imag = xi
    return X


def nmf_mixture_batch(first_seed, n_mixtures, n_sources, n_bins, n_frames, workers=None):
    """Mixtures ``first_seed .. first_seed + n_mixtures - 1`` stacked on a leading axis (the seeding
    rule of SURVEY.md 8d: mixture b of a batch uses seed 1000 + b); generated on a thread pool (the
    NumPy generators and ufuncs release the GIL)."""
    import os

    workers = workers or min(32, os.cpu_count() or 1, n_mixtures)
    out = np.empty((n_mixtures, n_sources, n_bins, n_frames), dtype=np.complex128)

    def fill(b):
        out[b] = nmf_mixture(first_seed + b, n_sources, n_bins, n_frames)

    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(fill, range(n_mixtures)))
    return out


def sha256_of(array):
    """SHA-256 of the C-contiguous bytes of an array."""
    return hashlib.sha256(np.ascontiguousarray(array).tobytes()).hexdigest()


def kernel_sources_sha256():
    """SHA-256 over the sources of the three ILRMA pass kernels (the kernels the committed PMC traffic
    figures describe): lets bench.py see when `profiles/roofline_traffic.json` predates a change."""
    import hashlib
    import os

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    h = hashlib.sha256()
    for name in ("ilrma_fast.hip", "fast_model.hpp", "fast_tiles.hpp", "tail_plan.hpp", "common.hpp"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()
