"""Minimal distortion principle (scale restoration), NumPy in / NumPy out, computed on the device.

ref: ssspy/algorithm/minimal_distortion_principle.py:6-43.
"""

from typing import Optional

import numpy as np

from .. import _device as dv
from .. import _ops


def minimal_distortion_principle(
    estimated: np.ndarray,
    reference: Optional[np.ndarray] = None,
    reference_id: Optional[int] = 0,
) -> np.ndarray:
    """Scale every (source, bin) of ``estimated`` (n_sources, n_bins, n_frames) by conj(z),
    z = <y, x_ref> / <y, y>, against channel ``reference_id`` of ``reference``
    (n_channels, n_bins, n_frames); ``reference_id=None`` = every channel in turn, stacked on a new
    leading axis of length n_channels (ref: minimal_distortion_principle.py:34-35)."""
    if reference_id is None:
        return np.stack([minimal_distortion_principle(estimated, reference=reference, reference_id=c)
                         for c in range(reference.shape[0])], axis=0)
    Y = dv.to_device(estimated[None], dtype=np.complex128)
    X = dv.to_device(reference[None], dtype=np.complex128)
    G = _ops.mdp_scale(_ops.cross_covariance(Y, X), _ops.cross_covariance(Y, Y), reference_id)
    return dv.to_host(_ops.separate(Y, G))[0]
