"""Batched small dense linear algebra on the device (NumPy in, NumPy out).

Device counterparts of the reference's ``ssspy.linalg`` functions that the demixing hot path
sits on (ssspy/linalg/_solve.py, inv.py, eigh.py): leading axes are batch axes, the trailing
two are the matrix; one lane of a wavefront owns one matrix.
"""

from typing import Optional, Tuple, Union

import numpy as np

from .. import _device as dv
from .. import _lib
from .._device import ptr

__all__ = ["solve", "inv2", "eigh", "eigh2"]


def _flat(a, tail):
    a = np.asarray(a)
    lead = a.shape[: a.ndim - tail]
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    return lead, n, np.ascontiguousarray(a.reshape((n,) + a.shape[a.ndim - tail:]), dtype=np.complex128)


def solve(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Solve ``a x = b`` for batches of N x N complex systems (N <= 8).

    ``b`` may be a stack of vectors (``b.ndim == a.ndim - 1``) or of matrices, as in the
    reference (ref: ssspy/linalg/_solve.py:9-21).  Raises LinAlgError on a singular matrix.
    """
    a = np.asarray(a)
    b = np.asarray(b)
    vector = a.ndim == b.ndim + 1
    if vector:
        b = b[..., None]
    lead = np.broadcast_shapes(a.shape[:-2], b.shape[:-2])
    a = np.broadcast_to(a, lead + a.shape[-2:])
    b = np.broadcast_to(b, lead + b.shape[-2:])
    _, n, A = _flat(a, 2)
    _, _, Bm = _flat(b, 2)
    N, nrhs = A.shape[-1], Bm.shape[-1]
    dA, dB = dv.to_device(A), dv.to_device(Bm)
    dX = dv.empty((n, N, nrhs), dv.c128, dA.device)
    info = dv.zeros((1,), dv.i32)
    _lib.check(_lib.load().ssspy_solve(ptr(dA), ptr(dB), ptr(dX), n, N, nrhs, ptr(info),
                                       dv.stream_handle()), "solve")
    _lib.raise_if_singular(int(info.item()), "solve")
    x = dv.to_host(dX).reshape(lead + (N, nrhs))
    return x[..., 0] if vector else x


def inv2(X: np.ndarray) -> np.ndarray:
    """Closed-form inverse of 2 x 2 matrices (ref: ssspy/linalg/inv.py:4-54)."""
    X = np.asarray(X)
    assert X.shape[-2:] == (2, 2), "2x2 matrix is expected, but given shape of {}.".format(X.shape)
    lead, n, A = _flat(X, 2)
    dA = dv.to_device(A)
    out = dv.empty((n, 2, 2), dv.c128, dA.device)
    _lib.check(_lib.load().ssspy_inv2(ptr(dA), ptr(out), n, dv.stream_handle()), "inv2")
    res = dv.to_host(out).reshape(lead + (2, 2))
    return res if np.iscomplexobj(X) else res.real


def eigh(A: np.ndarray, B: Optional[np.ndarray] = None, type: int = 1
         ) -> Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]:
    """Hermitian (generalised when ``B`` is given) eigen-decomposition, eigenvalues ascending.

    Standard problem: M <= 8, cyclic complex Jacobi on the device.  Generalised problem: built
    for 2 x 2 (``eigh2``).  Eigenvectors are unit-norm (standard) with an arbitrary phase that
    differs from LAPACK's.  ref: ssspy/linalg/eigh.py:8-81.
    """
    if B is not None:
        if np.asarray(A).shape[-1] != 2:
            raise NotImplementedError("generalised eigh is built for 2 x 2 matrices (eigh2) only.")
        return eigh2(A, B, type=type)
    A = np.asarray(A)
    lead, n, Af = _flat(A, 2)
    M = Af.shape[-1]
    dA = dv.to_device(Af)
    lamb = dv.empty((n, M), dv.f64, dA.device)
    V = dv.empty((n, M, M), dv.c128, dA.device)
    _lib.check(_lib.load().ssspy_eigh(ptr(dA), ptr(lamb), ptr(V), n, M, dv.stream_handle()), "eigh")
    vec = dv.to_host(V).reshape(lead + (M, M))
    return dv.to_host(lamb).reshape(lead + (M,)), (vec if np.iscomplexobj(A) else vec.real)


def eigh2(A: np.ndarray, B: Optional[np.ndarray] = None, type: int = 1
          ) -> Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]:
    """(Generalised) eigen-decomposition of 2 x 2 Hermitian matrices (ref: eigh.py:84-207).

    type 1: ``A z = lamb B z``; type 2: ``A B z = lamb z``; type 3: ``B A z = lamb z``.
    """
    A = np.asarray(A)
    assert A.shape[-2:] == (2, 2), "2x2 matrix is expected, but given shape of {}.".format(A.shape)
    if B is None:
        return eigh(A)
    if type not in (1, 2, 3):
        raise ValueError("Invalid type={} is given.".format(type))
    B = np.asarray(B)
    lead = np.broadcast_shapes(A.shape[:-2], B.shape[:-2])
    _, n, Af = _flat(np.broadcast_to(A, lead + (2, 2)), 2)
    _, _, Bf = _flat(np.broadcast_to(B, lead + (2, 2)), 2)
    dA, dB = dv.to_device(Af), dv.to_device(Bf)
    lamb = dv.empty((n, 2), dv.f64, dA.device)
    Z = dv.empty((n, 2, 2), dv.c128, dA.device)
    info = dv.zeros((1,), dv.i32)
    _lib.check(_lib.load().ssspy_eigh2(ptr(dA), ptr(dB), ptr(lamb), ptr(Z), n, type, ptr(info),
                                       dv.stream_handle()), "eigh2")
    if int(info.item()):
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    z = dv.to_host(Z).reshape(lead + (2, 2))
    complex_in = np.iscomplexobj(A) or np.iscomplexobj(B)
    return dv.to_host(lamb).reshape(lead + (2,)), (z if complex_in else z.real)
