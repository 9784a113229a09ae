"""Oracle: GaussMNMF (multichannel NMF with full-rank spatial covariance matrices).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``GaussMNMF`` of the reference, with and without partitioning:
source model ``lambda_nij = sum_k t_nik v_nkj``, spatial covariance ``H_ni`` (M x M Hermitian),
model covariance ``R_ij = to_psd(sum_n lambda_nij H_ni)``.  All updates are the reference's MM
rules; the matrix geometric mean is written through Hermitian square roots (the reference goes
through a generalised eigenproblem, ``linalg/mean.py:6-83`` type 2 -- the same matrix
``A^-1 # B``).
"""

import numpy as np

from . import spatial as sp


def sqrtmh(A):
    """Hermitian PSD square root.  ref: ssspy/linalg/sqrtm.py:8-24."""
    lamb, P = np.linalg.eigh(A)
    return (P * np.sqrt(np.maximum(lamb, 0.0))[..., None, :]) @ P.swapaxes(-2, -1).conj()


def gmean_inv_a_b(A, B):
    """Geometric mean of A^-1 and B: A^-1/2 (A^1/2 B A^1/2)^1/2 A^-1/2.

    ref: ssspy/linalg/mean.py:6-83 (``gmeanmh(A, B, type=2)`` = ``inv(A) @ (A B)^(1/2)``).
    """
    As = sqrtmh(A)
    Ais = np.linalg.inv(As)
    return Ais @ sqrtmh(As @ B @ As) @ Ais


class GaussMNMFOracle:
    """ref: ssspy/bss/mnmf.py:681-1073 (GaussMNMF), :300-414 (MNMF), :21-297 (MNMFBase)."""

    def __init__(
        self,
        n_basis,
        n_sources=None,
        flooring=sp.DEFAULT_FLOOR,
        normalization=True,
        record_loss=True,
        reference_id=0,
        rng=None,
        partitioning=False,
    ):
        # partitioning: shared basis (F, K) / activation (K, T), latent Z (N, K)
        self.partitioning = partitioning
        self.n_basis = n_basis
        self.n_sources = n_sources
        self.flooring = flooring
        self.normalization = normalization
        self.record_loss = record_loss
        self.reference_id = reference_id
        self.rng = np.random.default_rng() if rng is None else rng
        self.loss = [] if record_loss else None

    def reset(self, X, basis=None, activation=None, spatial=None, latent=None):
        """ref: ssspy/bss/mnmf.py:139-165 (_reset), :167-188, :190-259, :327-353."""
        self.input = X.copy()
        M, F, T = X.shape
        N = M if self.n_sources is None else self.n_sources
        self.n_sources, self.n_channels = N, M
        self.n_bins, self.n_frames = F, T
        XX = (X[:, None] * X[None, :].conj()).transpose(2, 3, 0, 1)  # (F, T, M, M)
        self.instant_covariance = sp.to_psd(XX, self.flooring)
        lead = () if self.partitioning else (N,)
        if basis is None:
            basis = sp.floor(self.rng.random(lead + (F, self.n_basis)), self.flooring)
        else:
            basis = basis.copy()
        if activation is None:
            activation = sp.floor(self.rng.random(lead + (self.n_basis, T)), self.flooring)
        else:
            activation = activation.copy()
        if self.partitioning:
            if latent is None:
                latent = self.rng.random((N, self.n_basis))
                latent = sp.floor(latent / latent.sum(axis=0), self.flooring)
            self.latent = latent.copy()
        if spatial is None:
            spatial = np.tile(np.eye(M, dtype=X.dtype) / M, (N, F, 1, 1))
        else:
            spatial = spatial.copy()
        self.basis, self.activation, self.spatial = basis, activation, spatial
        self.output = self.separate(self.input)

    # shared intermediates -----------------------------------------------------
    def _lamb(self):
        """(N, F, T).  ref: ssspy/bss/mnmf.py:264-297."""
        if self.partitioning:
            return np.einsum("nk,ik,kj->nij", self.latent, self.basis, self.activation)
        return self.basis @ self.activation

    def _model_covariance(self, Lamb):
        """R_ij = to_psd(sum_n lambda_nij H_ni) -> (F, T, M, M).  ref: mnmf.py:355-389, :878-879."""
        R = np.sum(Lamb[:, :, :, None, None] * self.spatial[:, :, None, :, :], axis=0)
        return sp.to_psd(R, self.flooring)

    def _traces(self):
        """tr(R^-1 XX R^-1 H_n) and tr(R^-1 H_n) -> two (N, F, T).  ref: mnmf.py:858-873."""
        R = self._model_covariance(self._lamb())
        Ri = np.linalg.inv(R)  # (F, T, M, M)
        RXX = Ri @ self.instant_covariance
        RH = Ri[None] @ self.spatial[:, :, None, :, :]  # (N, F, T, M, M)
        tr_rxxrh = np.real(np.trace(RXX[None] @ RH, axis1=-2, axis2=-1))
        tr_rh = np.real(np.trace(RH, axis1=-2, axis2=-1))
        return tr_rxxrh, tr_rh

    def update_basis(self):
        """ref: ssspy/bss/mnmf.py:836-901."""
        a, b = self._traces()
        V = self.activation
        if self.partitioning:
            num = np.einsum("nk,kj,nij->ik", self.latent, V, a)
            den = np.einsum("nk,kj,nij->ik", self.latent, V, b)
        else:
            num = np.sum(V[:, None, :, :] * a[:, :, None, :], axis=-1)
            den = np.sum(V[:, None, :, :] * b[:, :, None, :], axis=-1)
        self.basis = sp.floor(self.basis * np.sqrt(num / den), self.flooring)

    def update_activation(self):
        """ref: ssspy/bss/mnmf.py:903-968."""
        a, b = self._traces()
        T = self.basis
        if self.partitioning:
            num = np.einsum("nk,ik,nij->kj", self.latent, T, a)
            den = np.einsum("nk,ik,nij->kj", self.latent, T, b)
        else:
            num = np.sum(T[:, :, :, None] * a[:, :, None, :], axis=1)
            den = np.sum(T[:, :, :, None] * b[:, :, None, :], axis=1)
        self.activation = sp.floor(self.activation * np.sqrt(num / den), self.flooring)

    def update_spatial(self):
        """ref: ssspy/bss/mnmf.py:970-1016."""
        Lamb = self._lamb()
        H = self.spatial
        Ri = np.linalg.inv(self._model_covariance(Lamb))  # (F, T, M, M)
        RXXR = Ri @ self.instant_covariance @ Ri
        P = np.sum(Lamb[:, :, :, None, None] * Ri[None], axis=2)  # (N, F, M, M)
        Q = np.sum(Lamb[:, :, :, None, None] * RXXR[None], axis=2)
        HQH = H @ Q @ H
        P = sp.to_psd(P, self.flooring)
        HQH = sp.to_psd(HQH, self.flooring)
        self.spatial = sp.to_psd(gmean_inv_a_b(P, HQH), self.flooring)

    def normalize(self):
        """Unit trace of H, scale moved into the basis.  ref: ssspy/bss/mnmf.py:391-414."""
        trace = np.real(np.trace(self.spatial, axis1=-2, axis2=-1))  # (N, F)
        self.spatial = self.spatial / trace[..., None, None]
        if not self.partitioning:  # with partitioning the scale cannot move into the shared basis
            self.basis = trace[:, :, None] * self.basis

    def update_latent(self):
        """z_nk <- z_nk sqrt(sum_ij t v a / sum_ij t v b), columns renormalised.

        ref: ssspy/bss/mnmf.py:1018-1073.
        """
        a, b = self._traces()
        num = np.einsum("ik,kj,nij->nk", self.basis, self.activation, a)
        den = np.einsum("ik,kj,nij->nk", self.basis, self.activation, b)
        Z = self.latent * np.sqrt(num / den)
        self.latent = Z / Z.sum(axis=0)

    def update_once(self):
        """ref: ssspy/bss/mnmf.py:806-834."""
        self.update_basis()
        self.update_activation()
        self.update_spatial()
        if self.normalization:
            self.normalize()
        if self.partitioning:
            self.update_latent()

    def compute_loss(self):
        """ref: ssspy/bss/mnmf.py:765-804."""
        R = self._model_covariance(self._lamb())
        trace = np.real(np.trace(np.linalg.solve(R, self.instant_covariance), axis1=-2, axis2=-1))
        logdet = np.linalg.slogdet(R)[1]
        return np.mean(trace + logdet, axis=-1).sum(axis=0).item()

    def separate(self, X):
        """Multichannel Wiener filter.  ref: ssspy/bss/mnmf.py:729-763."""
        Lamb = self._lamb()
        R_n = Lamb[:, :, :, None, None] * self.spatial[:, :, None, :, :]  # (N, F, T, M, M)
        R = sp.to_psd(np.sum(R_n, axis=0), self.flooring)
        WH = np.linalg.solve(R[None], R_n)
        W = WH.swapaxes(-2, -1).conj()
        W_ref = W[:, :, :, self.reference_id, :].transpose(0, 3, 1, 2)  # (N, M, F, T)
        return np.sum(W_ref * X, axis=1)

    def run(self, X, n_iter=100, **init):
        """ref: ssspy/bss/mnmf.py:90-118 and ssspy/bss/base.py:48-77."""
        self.reset(X, **init)
        if self.record_loss:
            self.loss.append(self.compute_loss())
        for _ in range(n_iter):
            self.update_once()
            if self.record_loss:
                self.loss.append(self.compute_loss())
        self.output = self.separate(self.input)
        return self.output
