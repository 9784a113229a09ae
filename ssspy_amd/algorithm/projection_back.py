"""Projection back (scale restoration), NumPy in / NumPy out, computed on the device.

ref: ssspy/algorithm/projection_back.py:6-121.
"""

from typing import Optional

import numpy as np

from .. import _device as dv
from .. import _lib, _ops


def projection_back(
    data_or_filter: np.ndarray,
    reference: Optional[np.ndarray] = None,
    reference_id: Optional[int] = 0,
) -> np.ndarray:
    """Restore the scale of demixing filters or of separated spectrograms.

    Args:
        data_or_filter: demixing filters (n_bins, n_sources, n_channels) when ``reference``
            is None, else separated spectrograms (n_sources, n_bins, n_frames).
        reference: the mixture (n_channels, n_bins, n_frames) for the spectrogram form.
        reference_id: reference channel; ``None`` = every channel in turn, stacked on a new leading
            axis of length n_channels (ref: projection_back.py:92-95, :113-116).
    """
    if reference_id is None:
        n_channels = data_or_filter.shape[-1] if reference is None else reference.shape[0]
        return np.stack([projection_back(data_or_filter, reference=reference, reference_id=c)
                         for c in range(n_channels)], axis=0)
    info = dv.zeros((1,), dv.i32)
    if reference is None:
        W = dv.to_device(data_or_filter[None], dtype=np.complex128)
        _ops.projection_back_filter(W, reference_id, info)
        _lib.raise_if_singular(int(info.item()), "projection_back")
        return dv.to_host(W)[0]
    Y = dv.to_device(data_or_filter[None], dtype=np.complex128)
    X = dv.to_device(reference[None], dtype=np.complex128)
    G = _ops.projection_back_scale(_ops.cross_covariance(X, Y), _ops.cross_covariance(Y, Y),
                                   reference_id, info)
    out = _ops.separate(Y, G)
    _lib.raise_if_singular(int(info.item()), "projection_back")
    return dv.to_host(out)[0]
