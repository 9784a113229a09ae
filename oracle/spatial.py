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
    if callable(flooring):
        # an arbitrary callable, as the reference accepts (ssspy/bss/ilrma.py:70-89)
        return flooring(x)
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
    """W <- W * (W^-1)[ref, :]^T (row scaling).  ref: ssspy/algorithm/projection_back.py:87-99;
    reference_id=None: every channel, stacked on a new leading axis (:92-95)."""
    if reference_id is None:
        return np.stack([projection_back_filter(W, c) for c in range(W.shape[-1])])
    scale = np.linalg.inv(W)[..., reference_id, :]
    return W * scale[..., None]


def projection_back_output(Y, X, reference_id=0):
    """Least-squares scale of Y on the reference channel of X.

    ref: ssspy/algorithm/projection_back.py:100-121.  Y, X (N, F, T) -> (N, F, T);
    reference_id=None: every channel, (n_channels, N, F, T) (:113-116).
    """
    if reference_id is None:
        return np.stack([projection_back_output(Y, X, c) for c in range(X.shape[0])])
    Yf = Y.transpose(1, 0, 2)
    Xf = X.transpose(1, 0, 2)
    YH = Yf.transpose(0, 2, 1).conj()
    scale = (Xf @ YH) @ np.linalg.inv(Yf @ YH)  # (F, N_ch, N_src)
    scale = scale[..., reference_id, :]
    return (Yf * scale[..., None]).swapaxes(-3, -2)


def minimal_distortion_output(Y, X, reference_id=0):
    """conj(z) y with z = <y, x_ref> / <y, y> per (source, bin).

    ref: ssspy/algorithm/minimal_distortion_principle.py:6-43 (reference_id=None: every channel, :34-35).
    """
    if reference_id is None:
        return np.stack([minimal_distortion_output(Y, X, c) for c in range(X.shape[0])])
    num = np.sum(Y * X[reference_id].conj(), axis=-1, keepdims=True)
    den = np.sum(np.abs(Y) ** 2, axis=-1, keepdims=True)
    return (num / den).conj() * Y


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


def eigh2(A, B):
    """Generalised 2x2 Hermitian eigenproblem A z = lamb B z via Cholesky of B.

    ref: ssspy/linalg/eigh.py:84-207 (eigh2 -> _eigh, type=1, inv=inv2).
    """
    L = np.linalg.cholesky(B)
    Li = inv2(L)
    C = Li @ A @ Li.swapaxes(-2, -1).conj()
    lamb, y = np.linalg.eigh(C)
    return lamb, Li.swapaxes(-2, -1).conj() @ y


def sequential_pairs(n_sources, stop=None, step=1):
    """ref: ssspy/utils/select_pair.py:5-44."""
    stop = n_sources if stop is None else stop
    return [(m % n_sources, (m + 1) % n_sources) for m in range(0, stop, step)]


def combination_pairs(n_sources):
    """ref: ssspy/utils/select_pair.py:47-78."""
    return [(m, n) for m in range(n_sources) for n in range(m + 1, n_sources)]


def update_by_ip2_one_pair(W, U_pair, pair, flooring=DEFAULT_FLOOR):
    """Pairwise iterative projection for sources (m, n).  ref: _update_spatial_model.py:317-395.

    W (F, N, N), U_pair (F, 2, N, N) -> new rows (F, 2, N).
    """
    m, n = pair
    F, N, M = W.shape
    U_m, U_n = U_pair[:, 0], U_pair[:, 1]
    E_mn = np.tile(np.eye(M, N)[:, (m, n)], (F, 1, 1))
    P_m = solve(W @ U_m, E_mn)
    P_n = solve(W @ U_n, E_mn)
    PUP_m = P_m.transpose(0, 2, 1).conj() @ U_m @ P_m
    PUP_n = P_n.transpose(0, 2, 1).conj() @ U_n @ P_n
    _, H = eigh2(PUP_m, PUP_n)
    H = H[..., ::-1]
    h_m, h_n = H.transpose(2, 0, 1)
    q = np.maximum(np.einsum("fa,fab,fb->f", h_m.conj(), PUP_m, h_m).real, 0)
    h_m = h_m / floor(np.sqrt(q), flooring)[:, None]
    q = np.maximum(np.einsum("fa,fab,fb->f", h_n.conj(), PUP_n, h_n).real, 0)
    h_n = h_n / floor(np.sqrt(q), flooring)[:, None]
    w_m = P_m @ h_m[..., None]
    w_n = P_n @ h_n[..., None]
    return np.concatenate([w_m, w_n], axis=-1).transpose(0, 2, 1).conj()


def update_by_ip2(W, U, flooring=DEFAULT_FLOOR, pairs=None):
    """ref: ssspy/bss/_update_spatial_model.py:81-143.  pairs: list of (m, n); default sequential."""
    W = W.copy()
    N = W.shape[1]
    if pairs is None:
        pairs = sequential_pairs(N)
    for m, n in pairs:
        W[:, (m, n), :] = update_by_ip2_one_pair(W, U[:, (m, n)], (m, n), flooring)
    return W


def update_by_iss2(Y, varphi, flooring=DEFAULT_FLOOR, pairs=None):
    """Pairwise iterative source steering.  ref: ssspy/bss/_update_spatial_model.py:197-314."""
    N = Y.shape[0]
    if pairs is None:
        pairs = sequential_pairs(N, stop=N, step=2)
    for m, n in pairs:
        m, n = m % N, n % N
        sub = [s for s in range(N) if s not in (m, n)]
        Y_main = Y[[m, n]]  # the pair in the order given (the reference's ascend / descend branches)
        v_main = varphi[[m, n]]
        YY_main = (Y_main[:, None] * Y_main[None].conj()).transpose(2, 0, 1, 3)  # (F, 2, 2, T)
        Ym = Y_main.transpose(1, 0, 2)  # (F, 2, T)
        Y_new = Y.copy()
        if sub:
            Y_sub = Y[sub]
            v_sub = varphi[sub]
            YY_sub = (Y_main[:, None] * Y_sub[None].conj()).transpose(1, 2, 0, 3)  # (S, F, 2, T)
            G_sub = np.mean(v_sub[:, :, None, None, :] * YY_main[None], axis=-1)
            Fv = np.mean(v_sub[:, :, None, :] * YY_sub, axis=-1)
            Q = (-inv2(G_sub) @ Fv[..., None])[..., 0].transpose(1, 0, 2)  # (F, S, 2)
            Y_new[sub] = Y_sub + (Q.conj() @ Ym).transpose(1, 0, 2)
        G_main = np.mean(v_main[:, :, None, None, :] * YY_main[None], axis=-1)  # (2, F, 2, 2)
        _, H = eigh2(G_main[0], G_main[1])
        h = H.transpose(2, 0, 1)  # (2, F, 2)
        q = np.einsum("kfa,kfab,kfb->kf", h.conj(), G_main, h).real
        P = h / floor(np.sqrt(np.maximum(q, 0)), flooring)[..., None]
        Y_new[[m, n]] = (P.transpose(1, 0, 2).conj() @ Ym).transpose(1, 0, 2)
        Y = Y_new
    return Y
