"""Flooring functions for numerical stability (host-side definitions).

Mirrors the interface of the reference's ``ssspy.special.flooring``
(ssspy/special/flooring.py:6-18).  The separators recognise these functions (also wrapped
in ``functools.partial(..., eps=...)``) and run the equivalent floor inside the HIP kernels.
"""

import numpy as np

EPS = 1e-10


def identity(input: np.ndarray) -> np.ndarray:
    """Return the input unchanged."""
    return input


def max_flooring(input: np.ndarray, eps: float = EPS) -> np.ndarray:
    """Clip from below: ``max(input, eps)`` element-wise."""
    return np.maximum(input, eps)


def add_flooring(input: np.ndarray, eps: float = EPS) -> np.ndarray:
    """Shift: ``input + eps``."""
    return input + eps
