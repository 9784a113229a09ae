"""Oracle: per-bin spatial update operators, flooring, scale restoration.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Each function restates one reference function; the citation gives the
reference file and line range it follows.  Floors are described by a
``(kind, eps)`` pair instead of a Python callable so that the oracle and the
HIP path share one vocabulary: kind is ``"max"`` (``np.maximum(x, eps)``),
``"add"`` (``x + eps``) or ``"none"``.
"""

import numpy as np

EPS = 1e-10
DEFAULT_FLOOR = ("max", EPS)


def floor(x, flooring=DEFAULT_FLOOR):
    """ref: ssspy/special/flooring.py:6-18 (identity / max_flooring / add_flooring)."""
    kind, eps = flooring
    if kind == "max":
        return np.maximum(x, eps)
    if kind == "add":
        return x + eps
    if kind == "none":
        return x
    raise ValueError("unknown flooring kind {!r}".format(kind))


def separate(X, W):
    """y_ij = W_i x_ij.  ref: ssspy/bss/ilrma.py:272-295, ssspy/bss/iva.py:171-194.

    X (N, F, T) complex, W (F, N, N) complex -> (N, F, T).
    """
    return np.matmul(W, X.transpose(1, 0, 2)).transpose(1, 0, 2)


def solve(a, b):
    """ref: ssspy/linalg/_solve.py:9-21 (vector right-hand side when b has one axis less)."""
    if a.ndim == b.ndim + 1:
        return np.linalg.solve(a, b[..., None])[..., 0]
    return np.linalg.solve(a, b)


def weighted_covariance(X, weight):
    """U_in = mean_j weight_{n,i,j} x_ij x_ij^H.

    ref: ssspy/bss/ilrma.py:1500-1505 (weight (N,F,T)), ssspy/bss/iva.py:1785-1791
    (weight (N,T), broadcast over bins).  Returns (F, N_src, N_ch, N_ch).
    """
    XX = X[:, None, :, :] * X[None, :, :, :].conj()  # (N, N, F, T)
    XX = XX.transpose(2, 0, 1, 3)  # (F, N, N, T)
    if weight.ndim == 2:
        GXX = weight[:, None, None, :] * XX[:, None, :, :, :]
    else:
        GXX = weight.transpose(1, 0, 2)[:, :, None, None, :] * XX[:, None, :, :, :]
    return GXX.mean(axis=-1)


def update_by_ip1(W, U, flooring=DEFAULT_FLOOR):
    """Iterative projection, one sweep over the sources, in place on a copy.

    ref: ssspy/bss/_update_spatial_model.py:17-78.
    W (F, N, N) complex, U (F, N, N, N) complex -> new W (F, N, N).
    """
    W = W.copy()
    F, N, M = W.shape
    eye = np.eye(N, M)
    for n in range(N):
        U_n = U[:, n]
        e_n = np.broadcast_to(eye[n], (F, M))
        w = solve(W @ U_n, e_n)  # (F, M)
        q = np.einsum("fa,fab,fb->f", w.conj(), U_n, w).real
        d = floor(np.sqrt(np.maximum(q, 0)), flooring)
        W[:, n, :] = w.conj() / d[:, None]
    return W


def update_by_iss1(Y, varphi, flooring=DEFAULT_FLOOR):
    """Iterative source steering, N sequential rank-1 sweeps.

    ref: ssspy/bss/_update_spatial_model.py:146-194.
    Y (N, F, T) complex, varphi (N, F, T) or (N, 1, T) real -> new Y.
    """
    N = Y.shape[0]
    for n in range(N):
        Y_n = Y[n]
        num = np.mean(varphi * (Y * Y_n.conj()), axis=-1)  # (N, F)
        den = floor(np.mean(varphi * (np.abs(Y_n) ** 2), axis=-1), flooring)
        v = num / den
        v[n] = 1 - 1 / np.sqrt(den[n])
        Y = Y - v[:, :, None] * Y_n
    return Y


def projection_back_filter(W, reference_id=0):
    """W <- W * (W^-1)[ref, :]^T (row scaling).  ref: ssspy/algorithm/projection_back.py:87-99."""
    scale = np.linalg.inv(W)[..., reference_id, :]
    return W * scale[..., None]


def projection_back_output(Y, X, reference_id=0):
    """Least-squares scale of Y on the reference channel of X.

    ref: ssspy/algorithm/projection_back.py:100-121.  Y, X (N, F, T) -> (N, F, T).
    """
    Yf = Y.transpose(1, 0, 2)
    Xf = X.transpose(1, 0, 2)
    YH = Yf.transpose(0, 2, 1).conj()
    scale = (Xf @ YH) @ np.linalg.inv(Yf @ YH)  # (F, N_ch, N_src)
    scale = scale[..., reference_id, :]
    return (Yf * scale[..., None]).swapaxes(-3, -2)


def demix_from_output(Y, X):
    """W_i = Y_i X_i^H (X_i X_i^H)^-1.  ref: ssspy/bss/ilrma.py:1938-1944, ssspy/bss/iva.py:2180-2185."""
    Xf, Yf = X.transpose(1, 0, 2), Y.transpose(1, 0, 2)
    XH = Xf.transpose(0, 2, 1).conj()
    return Yf @ XH @ np.linalg.inv(Xf @ XH)


def logdet(W):
    """ref: ssspy/bss/ilrma.py:524-536, ssspy/bss/iva.py:224-236, ssspy/bss/mnmf.py:1263-1276."""
    return np.linalg.slogdet(W)[1]


def to_psd(X, flooring=DEFAULT_FLOOR):
    """Hermitise, floor the eigenvalues, rebuild, Hermitise.  ref: ssspy/special/psd.py:11-71."""
    X = (X + X.swapaxes(-2, -1).conj()) / 2
    lamb, P = np.linalg.eigh(X)
    lamb = floor(lamb, flooring)
    X = (P * lamb[..., None, :]) @ P.swapaxes(-2, -1).conj()
    return (X + X.swapaxes(-2, -1).conj()) / 2


def inv2(X):
    """Closed-form 2x2 inverse.  ref: ssspy/linalg/inv.py:4-54."""
    a, b, c, d = X[..., 0, 0], X[..., 0, 1], X[..., 1, 0], X[..., 1, 1]
    det = a * d - b * c
    out = np.stack([np.stack([d, -b], axis=-1), np.stack([-c, a], axis=-1)], axis=-2)
    return out / det[..., None, None]
