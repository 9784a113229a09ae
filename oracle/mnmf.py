"""Oracle: FastGaussMNMF with the IP1 and the pairwise (IP2) diagonaliser update.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``FastGaussMNMF`` of the reference (SURVEY.md section 8 rows a16-a21,
Appendix D): jointly diagonalisable full-rank spatial model with per-bin
diagonaliser Q (F, M, M), diagonal spatial D (F, N, M), NMF basis T (N, F, K) and
activation V (N, K, T); multichannel Wiener filter for the output.
"""

import numpy as np

from . import spatial as sp


class FastGaussMNMFOracle:
    """ref: ssspy/bss/mnmf.py:1076-1675 (FastGaussMNMF), :417-678 (FastMNMFBase), :21-297."""

    def __init__(
        self,
        n_basis,
        n_sources=None,
        flooring=sp.DEFAULT_FLOOR,
        normalization=True,
        record_loss=True,
        reference_id=0,
        rng=None,
        diagonalizer_algorithm="IP",
        pairs=None,
    ):
        # ref: ssspy/bss/mnmf.py:1140-1153 -- "IP" / "IP1" / "IP2"; `pairs` is the list the
        # reference's pair_selector(n_channels) yields (None: sequential_pair_selector)
        assert diagonalizer_algorithm in ("IP", "IP1", "IP2")
        self.diagonalizer_algorithm = diagonalizer_algorithm
        self.pairs = pairs
        self.n_basis = n_basis
        self.n_sources = n_sources
        self.flooring = flooring
        self.normalization = normalization
        self.record_loss = record_loss
        self.reference_id = reference_id
        self.rng = np.random.default_rng() if rng is None else rng
        self.loss = [] if record_loss else None

    def reset(self, X, basis=None, activation=None, diagonalizer=None, spatial=None):
        """ref: ssspy/bss/mnmf.py:499-600 (FastMNMFBase._reset and the _init_* helpers).

        The instantaneous covariance of the reference (:167-188) is never read by
        the FastGaussMNMF updates and is not materialised here.
        """
        self.input = X.copy()
        M, F, T = X.shape
        N = M if self.n_sources is None else self.n_sources
        self.n_sources, self.n_channels = N, M
        self.n_bins, self.n_frames = F, T
        # order of the random draws follows the reference: basis, activation, spatial
        if basis is None:
            basis = sp.floor(self.rng.random((N, F, self.n_basis)), self.flooring)
        else:
            basis = basis.copy()
        if activation is None:
            activation = sp.floor(self.rng.random((N, self.n_basis, T)), self.flooring)
        else:
            activation = activation.copy()
        if diagonalizer is None:
            diagonalizer = np.tile(np.eye(M, dtype=np.complex128), (F, 1, 1))
        else:
            diagonalizer = diagonalizer.copy()
        if spatial is None:
            spatial = sp.floor(self.rng.random((F, N, M)), self.flooring)
        self.basis, self.activation = basis, activation
        self.diagonalizer, self.spatial = diagonalizer, spatial
        self.output = self.separate(self.input)

    # shared intermediates -----------------------------------------------------
    def _lamb(self):
        return self.basis @ self.activation  # (N, F, T)

    def _lambD(self, Lamb):
        """R~_{ijm} = sum_n lambda_{nij} d_{inm} -> (F, T, M).  ref: ssspy/bss/mnmf.py:1344-1346."""
        D = self.spatial.transpose(1, 0, 2)  # (N, F, M)
        return np.sum(Lamb[:, :, :, None] * D[:, :, None, :], axis=0)

    def _abs_qx(self):
        """|Q_i x_ij| -> (F, M, T).  ref: ssspy/bss/mnmf.py:1347-1348."""
        return np.abs(self.diagonalizer @ self.input.transpose(1, 0, 2))

    def update_basis(self):
        """ref: ssspy/bss/mnmf.py:1305-1360."""
        T, V = self.basis, self.activation
        D = self.spatial.transpose(1, 0, 2)  # (N, F, M)
        LambD = self._lambD(self._lamb())  # (F, T, M)
        QX = self._abs_qx().transpose(0, 2, 1)  # (F, T, M)
        QXLambD = (QX / LambD) ** 2
        DQXLambD = np.sum(D[:, :, None, :] * QXLambD, axis=-1)  # (N, F, T)
        DLambD = np.sum(D[:, :, None, :] / LambD, axis=-1)
        num = np.sum(V[:, None, :] * DQXLambD[:, :, None], axis=-1)
        den = np.sum(V[:, None, :] * DLambD[:, :, None], axis=-1)
        self.basis = sp.floor(T * np.sqrt(num / den), self.flooring)

    def update_activation(self):
        """ref: ssspy/bss/mnmf.py:1362-1417."""
        T, V = self.basis, self.activation
        D = self.spatial.transpose(1, 0, 2)
        LambD = self._lambD(self._lamb())
        QX = self._abs_qx().transpose(0, 2, 1)
        QXLambD = (QX / LambD) ** 2
        DQXLambD = np.sum(D[:, :, None, :] * QXLambD, axis=-1)
        DLambD = np.sum(D[:, :, None, :] / LambD, axis=-1)
        num = np.sum(T[:, :, :, None] * DQXLambD[:, :, None, :], axis=1)
        den = np.sum(T[:, :, :, None] * DLambD[:, :, None, :], axis=1)
        self.activation = sp.floor(V * np.sqrt(num / den), self.flooring)

    def update_diagonalizer(self):
        """ref: ssspy/bss/mnmf.py:1419-1447 (dispatch), :1449-1514 (IP1), :1516-1633 (IP2: the same
        weighted covariances, then the pairwise projection of _update_spatial_model.py:81-143)."""
        X = self.input
        Lamb = self._lamb().transpose(1, 0, 2)  # (F, N, T)
        LambD = np.sum(Lamb[:, :, None, :] * self.spatial[:, :, :, None], axis=1)  # (F, M, T)
        varphi = 1 / LambD
        XX = (X[:, None, :, :] * X[None, :, :, :].conj()).transpose(2, 0, 1, 3)  # (F, M, M, T)
        U = np.mean(varphi[:, :, None, None, :] * XX[:, None, :, :, :], axis=-1)
        if self.diagonalizer_algorithm == "IP2":
            self.diagonalizer = sp.update_by_ip2(self.diagonalizer, U, self.flooring,
                                                 pairs=self.pairs)
        else:
            self.diagonalizer = sp.update_by_ip1(self.diagonalizer, U, self.flooring)

    def update_spatial(self):
        """ref: ssspy/bss/mnmf.py:1635-1675 (no flooring)."""
        QX2 = self._abs_qx() ** 2  # (F, M, T)
        Lamb = self._lamb().transpose(1, 0, 2)  # (F, N, T)
        D = self.spatial
        LambD = np.sum(Lamb[:, :, None, :] * D[:, :, :, None], axis=1)  # (F, M, T)
        num = np.sum((Lamb[:, :, None] / (LambD**2)[:, None, :]) * QX2[:, None, :, :], axis=-1)
        den = np.sum(Lamb[:, :, None] / LambD[:, None, :], axis=-1)
        self.spatial = np.sqrt(num / den) * D

    def normalize(self):
        """ref: ssspy/bss/mnmf.py:632-678."""
        QX = self.diagonalizer @ self.input.transpose(1, 0, 2)
        psi = sp.floor(np.sqrt(np.mean(np.abs(QX) ** 2, axis=(0, 2))), self.flooring)
        self.diagonalizer = self.diagonalizer / psi[None, :, None]
        self.spatial = self.spatial / (psi**2)

    def update_once(self):
        """ref: ssspy/bss/mnmf.py:1278-1303."""
        self.update_basis()
        self.update_activation()
        self.update_diagonalizer()
        self.update_spatial()
        if self.normalization:
            self.normalize()

    def compute_loss(self):
        """ref: ssspy/bss/mnmf.py:1219-1261."""
        D = self.spatial.transpose(1, 0, 2)
        LambD = np.sum(self._lamb()[:, :, None, :] * D[:, :, :, None], axis=0)  # (F, M, T)
        QX2 = self._abs_qx() ** 2
        loss = np.sum(QX2 / LambD + np.log(LambD), axis=1)  # (F, T)
        loss = np.mean(loss, axis=-1) - 2 * sp.logdet(self.diagonalizer)
        return loss.sum(axis=0).item()

    def separate(self, X):
        """Multichannel Wiener filter.  ref: ssspy/bss/mnmf.py:1174-1217."""
        N = self.n_sources
        Lamb = self._lamb()  # (N, F, T)
        D = self.spatial.transpose(1, 0, 2)  # (N, F, M)
        Qi = np.linalg.inv(self.diagonalizer)  # (F, M, M)
        QiH = Qi.transpose(0, 2, 1).conj()
        QQ = Qi[:, :, :, None] * QiH[:, None, :, :]  # (F, M, M', M): Qi[a,m] * conj(Qi[b,m])
        LambD = Lamb[:, :, :, None] * D[:, :, None, :]  # (N, F, T, M)
        R_n = np.sum(LambD[:, :, :, None, :, None] * QQ[:, None, :, :, :], axis=4)  # (N,F,T,M,M)
        R = sp.to_psd(np.sum(R_n, axis=0), self.flooring)
        R = np.tile(R, (N, 1, 1, 1, 1))
        WH = sp.solve(R, R_n)
        W = WH.transpose(0, 1, 2, 4, 3).conj()
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
