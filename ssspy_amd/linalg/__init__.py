"""Batched small dense linear algebra on the device (NumPy in, NumPy out).

Device counterparts of the reference's ``ssspy.linalg`` functions that the demixing hot path
sits on (ssspy/linalg/_solve.py, inv.py, eigh.py): leading axes are batch axes, the trailing
two are the matrix; one lane of a wavefront owns one matrix, and from 7 x 7 on the Hermitian
functions (eigh, sqrtmh, invsqrtmh, gmeanmh) spread a matrix over 8 lanes, a row per lane
(csrc/herm_rows8.hpp); 9 x 9 .. 16 x 16 run with the size as a kernel argument, a lane per matrix
on private memory (csrc/hermitian_rt.hip) -- correct, not tuned.
"""

from typing import Optional, Tuple, Union

import numpy as np

from .. import _device as dv
from .. import _lib
from .._device import ptr

__all__ = ["solve", "inv2", "eigh", "eigh2", "sqrtmh", "invsqrtmh", "gmeanmh", "lqpqm2"]


def _flat(a, tail):
    a = np.asarray(a)
    lead = a.shape[: a.ndim - tail]
    n = int(np.prod(lead, dtype=np.int64)) if lead else 1
    return lead, n, np.ascontiguousarray(a.reshape((n,) + a.shape[a.ndim - tail:]), dtype=np.complex128)


def solve(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Solve ``a x = b`` for batches of N x N complex systems (N <= 16).

    ``b`` may be a stack of vectors (``b.ndim == a.ndim - 1``) or of matrices, as in the
    reference (ref: ssspy/linalg/_solve.py:9-21).  Raises LinAlgError on a singular matrix.
    """
    a = np.asarray(a)
    b = np.asarray(b)
    real = not (np.iscomplexobj(a) or np.iscomplexobj(b))  # np.linalg.solve keeps real systems real
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
    if real:
        x = np.ascontiguousarray(x.real)
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

    M <= 16, cyclic complex Jacobi on the device; the generalised problem goes through the Cholesky
    factor of ``B`` like the reference (type 1: ``A z = lamb B z``; 2: ``A B z = lamb z``;
    3: ``B A z = lamb z``).  Eigenvectors carry an arbitrary phase that differs from LAPACK's.
    ref: ssspy/linalg/eigh.py:8-81, :164-207.
    """
    if B is not None:
        if np.asarray(A).shape[-1] == 2:
            return eigh2(A, B, type=type)
        if type not in (1, 2, 3):
            raise ValueError("Invalid type={} is given.".format(type))
        A, B = np.asarray(A), np.asarray(B)
        lead = np.broadcast_shapes(A.shape[:-2], B.shape[:-2])
        M = A.shape[-1]
        _, n, Af = _flat(np.broadcast_to(A, lead + (M, M)), 2)
        _, _, Bf = _flat(np.broadcast_to(B, lead + (M, M)), 2)
        dA, dB = dv.to_device(Af), dv.to_device(Bf)
        lamb = dv.empty((n, M), dv.f64, dA.device)
        Z = dv.empty((n, M, M), dv.c128, dA.device)
        info = dv.zeros((1,), dv.i32)
        _lib.check(_lib.load().ssspy_eigh_general(ptr(dA), ptr(dB), ptr(lamb), ptr(Z), n, M, type,
                                                  ptr(info), dv.stream_handle()), "eigh")
        if int(info.item()):
            raise np.linalg.LinAlgError("Matrix is not positive definite")
        z = dv.to_host(Z).reshape(lead + (M, M))
        complex_in = np.iscomplexobj(A) or np.iscomplexobj(B)
        return dv.to_host(lamb).reshape(lead + (M,)), (z if complex_in else z.real)
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


def herm_rebuild(P: np.ndarray, w: np.ndarray, hermitise: bool = False) -> np.ndarray:
    """``P diag(w) P^H`` on the device (optionally Hermitised): the rebuild half of ``to_psd`` /
    ``invsqrtmh`` when the eigenvalue map is a host callable."""
    P, w = np.asarray(P), np.asarray(w, dtype=np.float64)
    lead, n, Pf = _flat(P, 2)
    M = Pf.shape[-1]
    dP = dv.to_device(Pf, dtype=np.complex128)
    dw = dv.to_device(w.reshape(n, M))
    out = dv.empty((n, M, M), dv.c128, dP.device)
    _lib.check(_lib.load().ssspy_herm_rebuild(ptr(dP), ptr(dw), ptr(out), n, M, int(bool(hermitise)),
                                              dv.stream_handle()), "herm_rebuild")
    res = dv.to_host(out).reshape(lead + (M, M))
    return res if np.iscomplexobj(P) else res.real


def _hermitian_fn(X, inverse, flooring):
    X = np.asarray(X)
    host = getattr(flooring, "host", None)
    if host is not None:  # invsqrtmh with an arbitrary callable: eigh -> host map -> rebuild
        lamb, P = eigh(X)
        return herm_rebuild(P, 1.0 / np.asarray(host(np.sqrt(lamb)), dtype=np.float64))
    lead, n, Xf = _flat(X, 2)
    M = Xf.shape[-1]
    dX = dv.to_device(Xf)
    out = dv.empty((n, M, M), dv.c128, dX.device)
    _lib.check(_lib.load().ssspy_sqrtmh(ptr(dX), ptr(out), n, M, int(inverse), flooring[0],
                                        flooring[1], dv.stream_handle()), "sqrtmh")
    res = dv.to_host(out).reshape(lead + (M, M))
    return res if np.iscomplexobj(X) else res.real


def sqrtmh(X: np.ndarray) -> np.ndarray:
    """Square root of positive semidefinite Hermitian matrices (ref: ssspy/linalg/sqrtm.py:8-24)."""
    return _hermitian_fn(X, False, (_lib.FLOOR_NONE, 0.0))


def invsqrtmh(X: np.ndarray, flooring_fn=None) -> np.ndarray:
    """Inverse square root, ``P diag(1 / flooring_fn(sqrt(lamb))) P^H`` (ref: sqrtm.py:27-64)."""
    from ..utils.flooring import device_flooring

    return _hermitian_fn(X, True, device_flooring(flooring_fn, allow_host=True))


def gmeanmh(A: np.ndarray, B: np.ndarray, type: int = 1) -> np.ndarray:
    """Geometric mean of Hermitian positive definite matrices (ref: ssspy/linalg/mean.py:6-83):
    type 1: ``A # B``; type 2: ``A^-1 # B``; type 3: ``A # B^-1``."""
    if type not in (1, 2, 3):
        raise ValueError("Invalid type={} is given.".format(type))
    A, B = np.asarray(A), np.asarray(B)
    lead = np.broadcast_shapes(A.shape[:-2], B.shape[:-2])
    M = A.shape[-1]
    _, n, Af = _flat(np.broadcast_to(A, lead + (M, M)), 2)
    _, _, Bf = _flat(np.broadcast_to(B, lead + (M, M)), 2)
    dA, dB = dv.to_device(Af), dv.to_device(Bf)
    G = dv.empty((n, M, M), dv.c128, dA.device)
    _lib.check(_lib.load().ssspy_gmeanmh(ptr(dA), ptr(dB), ptr(G), n, M, type, dv.stream_handle()),
               "gmeanmh")
    res = dv.to_host(G).reshape(lead + (M, M))
    return res if (np.iscomplexobj(A) or np.iscomplexobj(B)) else res.real


def lqpqm2(H: np.ndarray, v: np.ndarray, z: np.ndarray, flooring_fn="default",
           singular_fn="flooring", max_iter: int = 10) -> np.ndarray:
    """Log-quadratically penalised quadratic minimisation, type 2 (ref: ssspy/linalg/lqpqm.py:13-110).

    H (n_bins, L, L) Hermitian, v (n_bins, L), z (n_bins,) -> y (n_bins, L).  The Newton loop stops
    when every problem has converged and warns when they have not after ``max_iter`` steps, like
    the reference (lqpqm.py:196-213).  ``singular_fn``: "flooring" (default), None or a callable on the
    norms of ``v``, as in the reference (lqpqm.py:61-78).  A singular problem follows the reference's
    indexing literally (lqpqm.py:84-93: ``scale * sigma[:, -1]`` on the ``(n_bins, L, L)`` eigenvector
    array): component ``a`` of the solution is the LAST entry of the eigenvector of the ``a``-th
    smallest eigenvalue of ``H`` -- the last row of the eigenvector matrix, not its last column.
    Every component carries a different eigenvector's arbitrary phase, so only the moduli are
    comparable between implementations.
    """
    import functools

    from ..special.flooring import max_flooring
    from ..utils.flooring import device_flooring

    if isinstance(flooring_fn, str) and flooring_fn == "default":
        flooring_fn = functools.partial(max_flooring, eps=1e-10)
    floor = device_flooring(flooring_fn, what="lqpqm2")
    # singular_fn (lqpqm.py:61-78): "flooring" -> ||v|| < flooring_fn(0), decided in the kernel; None
    # -> ||v|| == 0; a callable -> its verdict on the (n_bins,) norms.  The last two are evaluated here
    # on the host (n numbers from the caller's own v) and handed to the kernel as a mask.
    mask = None
    if singular_fn is None:
        mask = np.linalg.norm(np.asarray(v), axis=-1) == 0
    elif not (isinstance(singular_fn, str) and singular_fn == "flooring"):
        assert callable(singular_fn), "singular_fn should be callable."
        mask = np.asarray(singular_fn(np.linalg.norm(np.asarray(v), axis=-1)), dtype=bool)
    H, v, z = np.asarray(H), np.asarray(v), np.asarray(z, dtype=np.float64)
    n, L = v.shape
    dH = dv.to_device(np.ascontiguousarray(H, dtype=np.complex128))
    dvv = dv.to_device(np.ascontiguousarray(v, dtype=np.complex128))
    dz = dv.to_device(np.ascontiguousarray(z))
    y = dv.empty((n, L), dv.c128, dH.device)
    newton_ws = dv.empty((1,), dv.i64, dH.device)
    not_converged = dv.zeros((1,), dv.i32, dH.device)
    if mask is None:
        _lib.check(_lib.load().ssspy_lqpqm2(ptr(dH), ptr(dvv), ptr(dz), ptr(y), n, L, int(max_iter),
                                            floor[0], floor[1], ptr(newton_ws), ptr(not_converged),
                                            dv.stream_handle()), "lqpqm2")
    else:
        if mask.shape != (n,):
            raise ValueError("singular_fn must return one flag per problem, got shape {}".format(mask.shape))
        dmask = dv.to_device(mask.astype(np.int32))
        _lib.check(_lib.load().ssspy_lqpqm2_masked(
            ptr(dH), ptr(dvv), ptr(dz), ptr(y), n, L, int(max_iter), floor[0], floor[1],
            ptr(newton_ws), ptr(not_converged), ptr(dmask), dv.stream_handle()), "lqpqm2")
    out = dv.to_host(y)
    if int(not_converged.item()):
        import warnings

        warnings.warn("Newton-Raphson method did not converge in {} iterations.".format(max_iter),
                      UserWarning)
    return out
