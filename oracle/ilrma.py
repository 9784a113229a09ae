"""Oracle: Gauss-ILRMA (IP1 / ISS1 spatial update, MM source update).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``GaussILRMA`` of the reference for ``source_algorithm="MM"``, no
partitioning, ``spatial_algorithm in {"IP","IP1","ISS","ISS1"}``, power
normalisation and projection-back scale restoration — the configuration on
the north-star path.  The per-iteration order
is Appendix A of SURVEY.md: basis, activation, spatial, normalise.
"""

import numpy as np

from . import spatial as sp


class GaussILRMAOracle:
    """State + update rules.  ref: ssspy/bss/ilrma.py:582-1989 (GaussILRMA)."""

    def __init__(
        self,
        n_basis,
        spatial_algorithm="IP",
        domain=2,
        flooring=sp.DEFAULT_FLOOR,
        normalization=True,
        scale_restoration=True,
        record_loss=True,
        reference_id=0,
        rng=None,
    ):
        assert spatial_algorithm in ("IP", "IP1", "IP2", "ISS", "ISS1", "ISS2")
        self.pairs = None  # None = the reference's default selector for the algorithm
        self.n_basis = n_basis
        self.spatial_algorithm = spatial_algorithm
        self.domain = domain
        self.flooring = flooring
        self.normalization = normalization
        self.scale_restoration = scale_restoration
        self.record_loss = record_loss
        self.reference_id = reference_id
        self.rng = np.random.default_rng() if rng is None else rng
        self.loss = [] if record_loss else None

    @property
    def uses_filter(self):
        return self.spatial_algorithm in ("IP", "IP1", "IP2")

    # -- initialisation ----------------------------------------------------
    def reset(self, X, basis=None, activation=None, demix_filter=None):
        """ref: ssspy/bss/ilrma.py:151-199 (ILRMABase._reset), :201-270 (_init_nmf), :875-898."""
        self.input = X.copy()
        N, F, T = X.shape
        self.n_sources = self.n_channels = N
        self.n_bins, self.n_frames = F, T
        if demix_filter is None:
            W = np.tile(np.eye(N, dtype=np.complex128), (F, 1, 1))
        else:
            W = demix_filter.copy()
        self.demix_filter = W
        self.output = sp.separate(self.input, W)
        if basis is None:
            basis = sp.floor(self.rng.random((N, F, self.n_basis)), self.flooring)
        else:
            basis = basis.copy()
        if activation is None:
            activation = sp.floor(self.rng.random((N, self.n_basis, T)), self.flooring)
        else:
            activation = activation.copy()
        self.basis, self.activation = basis, activation
        if not self.uses_filter:
            self.demix_filter = None

    def _current_output(self):
        if self.demix_filter is None:
            return self.output
        return sp.separate(self.input, self.demix_filter)

    # -- one iteration -----------------------------------------------------
    def update_basis(self):
        """ref: ssspy/bss/ilrma.py:1051-1128 (update_basis_mm, no partitioning)."""
        p = self.domain
        Y2 = np.abs(self._current_output()) ** 2
        T, V = self.basis, self.activation
        TV = T @ V
        TVp2p = TV ** ((p + 2) / p)
        num = np.sum((V[:, None, :, :] / TVp2p[:, :, None, :]) * Y2[:, :, None, :], axis=3)
        den = np.sum(V[:, None, :, :] / TV[:, :, None, :], axis=3)
        self.basis = sp.floor(((num / den) ** (p / (p + 2))) * T, self.flooring)

    def update_activation(self):
        """ref: ssspy/bss/ilrma.py:1130-1204 (update_activation_mm, no partitioning)."""
        p = self.domain
        Y2 = np.abs(self._current_output()) ** 2
        T, V = self.basis, self.activation
        TV = T @ V
        TVp2p = TV ** ((p + 2) / p)
        num = np.sum((T[:, :, :, None] / TVp2p[:, :, None, :]) * Y2[:, :, None, :], axis=1)
        den = np.sum(T[:, :, :, None] / TV[:, :, None, :], axis=1)
        self.activation = sp.floor(((num / den) ** (p / (p + 2))) * V, self.flooring)

    def update_spatial(self):
        """ref: ssspy/bss/ilrma.py:1440-1507 (IP1), :1635-1696 (ISS1)."""
        p = self.domain
        varphi = 1 / ((self.basis @ self.activation) ** (2 / p))
        N = self.n_sources
        if self.spatial_algorithm == "IP2":
            # ref: ssspy/bss/ilrma.py:1509-1633 (default pair_selector: sequential, :796-798)
            U = sp.weighted_covariance(self.input, varphi)
            pairs = self.pairs if self.pairs is not None else sp.sequential_pairs(N)
            self.demix_filter = sp.update_by_ip2(self.demix_filter, U, self.flooring, pairs)
        elif self.spatial_algorithm == "ISS2":
            # ref: ssspy/bss/ilrma.py:1698-1792
            pairs = self.pairs if self.pairs is not None else sp.sequential_pairs(N)
            self.output = sp.update_by_iss2(self.output, varphi, self.flooring, pairs)
        elif self.uses_filter:
            U = sp.weighted_covariance(self.input, varphi)
            self.demix_filter = sp.update_by_ip1(self.demix_filter, U, self.flooring)
        else:
            self.output = sp.update_by_iss1(self.output, varphi, self.flooring)

    def normalize(self):
        """ref: ssspy/bss/ilrma.py:365-444 (normalize_by_power, no partitioning)."""
        p = self.domain
        Y = self._current_output()
        psi = sp.floor(np.sqrt(np.mean(np.abs(Y) ** 2, axis=(-2, -1))), self.flooring)
        self.basis = self.basis / (psi[:, None, None] ** p)
        if self.demix_filter is None:
            self.output = Y / psi[:, None, None]
        else:
            self.demix_filter = self.demix_filter / psi[None, :, None]

    def update_once(self):
        """ref: ssspy/bss/ilrma.py:900-922."""
        self.update_basis()
        self.update_activation()
        self.update_spatial()
        if self.normalization:
            self.normalize()

    def compute_loss(self):
        """ref: ssspy/bss/ilrma.py:1910-1967."""
        p = self.domain
        if self.demix_filter is None:
            Y = self.output
            W = sp.demix_from_output(Y, self.input)
        else:
            W = self.demix_filter
            Y = sp.separate(self.input, W)
        Y2 = np.abs(Y) ** 2
        TV = self.basis @ self.activation
        loss = Y2 / (TV ** (2 / p)) + (2 / p) * np.log(TV)
        loss = np.sum(loss.mean(axis=-1), axis=0) - 2 * sp.logdet(W)
        return loss.sum(axis=0).item()

    # -- driver --------------------------------------------------------------
    def restore_scale(self):
        """ref: ssspy/bss/ilrma.py:538-565, :1969-1979 (projection back)."""
        if self.demix_filter is None:
            self.output = sp.projection_back_output(self.output, self.input, self.reference_id)
        else:
            self.demix_filter = sp.projection_back_filter(self.demix_filter, self.reference_id)
            self.output = sp.separate(self.input, self.demix_filter)

    def run(self, X, n_iter=100, **init):
        """ref: ssspy/bss/ilrma.py:820-855 and ssspy/bss/base.py:48-77."""
        self.reset(X, **init)
        if self.record_loss:
            self.loss.append(self.compute_loss())
        for _ in range(n_iter):
            self.update_once()
            if self.record_loss:
                self.loss.append(self.compute_loss())
        if self.scale_restoration:
            self.restore_scale()
        if self.demix_filter is not None:
            self.output = sp.separate(self.input, self.demix_filter)
        return self.output
