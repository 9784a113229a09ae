"""Projection onto (numerically) positive semidefinite Hermitian matrices, on the device."""

import functools
from typing import Callable, Optional

import numpy as np

from .. import _device as dv
from .. import _lib
from .._device import ptr
from .flooring import max_flooring

EPS = 1e-10


def to_psd(
    X: np.ndarray,
    axis1: int = -2,
    axis2: int = -1,
    flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
        max_flooring, eps=EPS
    ),
) -> np.ndarray:
    """Hermitise, floor the eigenvalues, rebuild, Hermitise (ref: ssspy/special/psd.py:11-71).

    Complex Jacobi on the device, M <= 16: a lane per matrix up to 6 x 6 and from 9 x 9 (there with the
    size at run time, csrc/hermitian_rt.hip), a matrix on 8 lanes (a row per lane,
    csrc/hermitian_rows.hip) at 7 x 7 and 8 x 8.  ``flooring_fn`` as for the separators.
    """
    from ..utils.flooring import device_flooring

    X = np.asarray(X)
    nd = X.ndim
    axis1 = nd + axis1 if axis1 < 0 else axis1
    axis2 = nd + axis2 if axis2 < 0 else axis2
    assert axis1 == nd - 2 and axis2 == nd - 1, "axis1 == -2 and axis2 == -1"
    floor = device_flooring(flooring_fn, allow_host=True)
    if floor.host is not None:
        # an arbitrary callable: Hermitise, eigh on the device, the callable on the eigenvalues on the
        # host (it sees the reference's (..., M) array), rebuild + Hermitise on the device
        from ..linalg import eigh, herm_rebuild

        Xh = (X + X.swapaxes(-2, -1).conj()) / 2 if np.iscomplexobj(X) else (X + X.swapaxes(-2, -1)) / 2
        lamb, P = eigh(Xh)
        return herm_rebuild(P, np.asarray(floor.host(lamb), dtype=np.float64), hermitise=True)
    lead = X.shape[:-2]
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    M = X.shape[-1]
    dA = dv.to_device(X.reshape((n, M, M)), dtype=np.complex128)
    out = dv.empty((n, M, M), dv.c128, dA.device)
    _lib.check(_lib.load().ssspy_to_psd(ptr(dA), ptr(out), n, M, floor[0], floor[1],
                                        dv.stream_handle()), "to_psd")
    res = dv.to_host(out).reshape(X.shape)
    return res if np.iscomplexobj(X) else res.real
