"""Seeded synthetic mixtures (the reference downloads speech; there is no network here).

``iid_mixture`` and ``nmf_mixture`` are the generators of SURVEY.md section 8d; the golden
vectors under tests/golden were produced from the same formulas.
"""

import numpy as np


def iid_mixture(seed, n_channels, n_bins, n_frames):
    """i.i.d. circular complex Gaussian tensor (n_channels, n_bins, n_frames)."""
    rng = np.random.default_rng(seed)
    shape = (n_channels, n_bins, n_frames)
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def nmf_mixture(seed, n_sources, n_bins, n_frames, n_latent=4):
    """Low-rank-variance sources through a random per-bin mixing matrix."""
    rng = np.random.default_rng(seed)
    N, F, T = n_sources, n_bins, n_frames
    R = (rng.random((N, F, n_latent)) ** 4) @ (rng.random((N, n_latent, T)) ** 4) + 1e-3
    g1 = rng.standard_normal((N, F, T))
    g2 = rng.standard_normal((N, F, T))
    S = np.sqrt(R / 2) * (g1 + 1j * g2)
    A = rng.standard_normal((F, N, N)) + 1j * rng.standard_normal((F, N, N))
    return (A @ S.transpose(1, 0, 2)).transpose(1, 0, 2)
