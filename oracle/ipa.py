"""Oracle: iterative projection with adjustment (IPA) and the LQPQM solver it sits on.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``update_by_ipa`` (ssspy/bss/_update_spatial_model.py:398-513) and ``lqpqm2`` /
``solve_equation`` / ``_find_largest_root`` (ssspy/linalg/lqpqm.py:13-352, linalg/cubic.py) one
frequency bin at a time -- the unit the device kernel works on -- except for the Newton loop of
``solve_equation``, which runs over all (non-singular) bins of the call at once and stops at the
first step at which every one of them has converged (lqpqm.py:196-213), warning when they have not
after ``max_iter`` steps.  One caveat: the degenerate branch ``||v|| < floor(0)`` of ``lqpqm2``
returns a scaled eigenvector whose phase is LAPACK's (lqpqm.py:78-92); it is restated, but no parity
is claimed for it.
"""

import warnings


import numpy as np

from . import spatial as sp


def floor0(flooring):
    """flooring_fn(0) of the reference: the threshold of its "is zero" tests."""
    return float(sp.floor(np.zeros(()), flooring))


def largest_cubic_root(A, B, C):
    """Largest real root of x^3 + A x^2 + B x + C as the reference's Cardano code returns it.

    ref: ssspy/linalg/lqpqm.py:292-331, ssspy/linalg/cubic.py (polar complex cube root).
    """
    P = -(A**2) / 3 + B
    Q = (2 * A**3) / 27 - (A * B) / 3 + C
    disc = (Q / 2) ** 2 + (P / 3) ** 3
    w = -Q / 2 + np.sqrt(complex(disc))
    U = np.cbrt(abs(w)) * np.exp(1j * np.angle(w) / 3)
    if U == 0:
        X1 = np.cbrt(-Q)
        V = -P / 3  # U replaced by 1 in the reference
        U = 1.0
    else:
        V = -P / (3 * U)
        X1 = U + V
    omega = (-1 + 1j * np.sqrt(3)) / 2
    X2 = (U * omega + V * omega.conjugate()).real
    X3 = (U * omega.conjugate() + V * omega).real
    roots = [np.real(X1)]
    if P < 0 and not (disc > 0):
        roots += [X2, X3]
    return max(roots) - A / 3


def _secular_setup(phi, v, z, flooring):
    """Masked, normalised coefficients and the Cardano start value of one bin.
    ref: ssspy/linalg/lqpqm.py:157-194 (normalization=True)."""
    f0 = floor0(flooring)
    mask = phi * np.abs(v) ** 2 >= f0
    phi = mask * phi
    v = mask * v
    l_max = int(np.argmax(phi))
    phi_max = float(sp.floor(phi[l_max], flooring))
    v_max = v[l_max] / phi_max
    phi, v, z = phi / phi_max, v / phi_max, z / phi_max
    A = -(abs(v_max) ** 2 + 2 + z)
    B = 1 + 2 * z
    C = -z
    lamb = largest_cubic_root(A, B, C)
    if not lamb > 1:
        lamb = 1 + f0
    lamb = max(lamb, z)
    return dict(phi=phi, w2=np.abs(v) ** 2, z=z, lamb=lamb, phi_max=phi_max)


def solve_equations(problems, flooring, max_iter):
    """Largest roots lambda_i of  lambda^2 sum_l phi_l |v_l|^2 / (lambda - phi_l)^2 - lambda + z = 0
    for a list of (phi, v, z): Newton steps on ALL of them until every |f_i| <= floor(0) at the same
    step, at most ``max_iter``; UserWarning when they have not converged by then.
    ref: ssspy/linalg/lqpqm.py:112-216."""
    f0 = floor0(flooring)
    st = [_secular_setup(phi, v, z, flooring) for phi, v, z in problems]
    if not st:
        return []

    def fn(s):
        return s["lamb"] ** 2 * np.sum(s["phi"] * s["w2"] / (s["lamb"] - s["phi"]) ** 2) - s["lamb"] + s["z"]

    broke = False
    for _ in range(max_iter):
        f = [fn(s) for s in st]
        if all(abs(fi) <= f0 for fi in f):
            broke = True
            break
        for s, fi in zip(st, f):
            df = -2 * s["lamb"] * np.sum(s["phi"] ** 2 * s["w2"] / (s["lamb"] - s["phi"]) ** 3) - 1
            mu = s["lamb"] - fi / df
            s["lamb"] = mu if mu > 1 else (1 + s["lamb"]) / 2
    if not broke and max_iter > 0 and not all(abs(fn(s)) <= f0 for s in st):
        warnings.warn("Newton-Raphson method did not converge in {} iterations.".format(max_iter),
                      UserWarning)
    return [s["lamb"] * s["phi_max"] for s in st]


def lqpqm2(H, v, z, flooring, max_iter, singular_fn="flooring"):
    """argmin of the log-quadratically penalised quadratic (type 2), a batch: H (n, L, L), v (n, L),
    z (n,) -> y (n, L).  ref: lqpqm.py:13-110.  singular_fn: "flooring" (||v|| < floor(0)), None
    (||v|| == 0) or a callable on the norms (lqpqm.py:61-78)."""
    n = len(H)
    y = [None] * n
    todo, problems = [], []
    norms = np.linalg.norm(np.asarray(v), axis=-1)
    if singular_fn is None:
        singular = norms == 0
    elif isinstance(singular_fn, str):
        singular = norms < floor0(flooring)
    else:
        singular = np.asarray(singular_fn(norms), dtype=bool)
    for i in range(n):
        phi, sigma = np.linalg.eigh(H[i])
        if singular[i]:
            lamb = max(z[i], phi[-1])
            # the reference indexes the (n_bins, L, L) eigenvector array with [:, -1] (lqpqm.py:87):
            # the LAST ROW of the eigenvector matrix, not the top eigenvector -- restated as is
            y[i] = np.sqrt(max((lamb - z[i]) / phi[-1], 0.0)) * sigma[-1, :]
            continue
        v_t = sigma.conj().T @ v[i]
        todo.append((i, phi, sigma, v_t))
        problems.append((phi, v_t, z[i]))
    for (i, phi, sigma, v_t), lamb in zip(todo, solve_equations(problems, flooring, max_iter)):
        y[i] = sigma @ (phi * v_t / (lamb - phi))
    return np.stack(y)


def _ipa_prepare_bin(U, s, flooring, normalization):
    """Everything of one bin up to the LQPQM problem (H, v, z).  U (N, N, N): U[n] = mean_j varphi_nj
    y_j y_j^H (not yet PSD-floored).  ref: ssspy/bss/_update_spatial_model.py:425-486."""
    N = U.shape[0]
    U = sp.to_psd(U, flooring)
    rest = [m for m in range(N) if m != s]
    U_s = U[s]
    lam, P = np.linalg.eigh(U_s)
    U_s_inv = (P / sp.floor(lam, flooring)) @ P.conj().T  # _psd_inv: a second floor on top of to_psd
    a = np.real(U[rest, s, s])
    b = U[rest, s, rest]
    Ui = U_s_inv.conj()
    C = Ui[np.ix_(rest, rest)]
    d = Ui[rest, s]
    Cd = np.linalg.solve(C, d)
    zz = np.real(U_s_inv[s, s]) - np.real(np.vdot(d, Cd))
    a_sqrt = np.sqrt(a)
    H = C / np.outer(a_sqrt, a_sqrt)
    v = -b / a_sqrt - a_sqrt * Cd
    if normalization:
        tr = np.real(np.trace(H))
        H, zz = H / tr, zz / tr
    return dict(H=H, v=v, z=zz, a=a, b=b, a_sqrt=a_sqrt, U_s=U_s, rest=rest, N=N, s=s)


def _ipa_finish_bin(st, q_check, flooring):
    """The N x N update matrix of one bin from the LQPQM solution: y <- G y.  ref: :487-511."""
    N, s, rest = st["N"], st["s"], st["rest"]
    q = q_check / st["a_sqrt"] - st["b"] / st["a"]
    q_tilde = np.zeros(N, dtype=np.complex128)
    q_tilde[s] = 1.0
    q_tilde[rest] = -q.conj()
    Uq = np.linalg.solve(st["U_s"], q_tilde)
    den = sp.floor(np.sqrt(max(np.real(np.vdot(q_tilde, Uq)), 0.0)), flooring)
    p = Uq / den
    G = np.eye(N, dtype=np.complex128)
    G[s, :] = p.conj()
    G[rest, s] = q.conj()
    return G


def ipa_transforms(U, s, flooring, normalization, max_iter):
    """Update matrices (F, N, N) of all bins for source ``s``; U (F, N, N, N)."""
    st = [_ipa_prepare_bin(U[i], s, flooring, normalization) for i in range(len(U))]
    y = lqpqm2([b["H"] for b in st], [b["v"] for b in st], [b["z"] for b in st], flooring, max_iter)
    return np.stack([_ipa_finish_bin(b, y[i], flooring) for i, b in enumerate(st)])


def update_by_ipa(Y, varphi, flooring=sp.DEFAULT_FLOOR, normalization=True, max_iter=1):
    """One IPA sweep over the sources.  Y (N, F, T), varphi (N, F, T) or (N, 1, T) -> new Y."""
    N, F, T = Y.shape
    Y = Y.copy()
    for s in range(N):
        YY = Y[:, None] * Y[None, :].conj()  # (a, b, F, T)
        U = np.mean(varphi[:, None, None] * YY, axis=-1).transpose(3, 0, 1, 2)  # (F, n, a, b)
        G = ipa_transforms(U, s, flooring, normalization, max_iter)
        Y = sp.separate(Y, G)
    return Y
