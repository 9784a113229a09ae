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
        model=("gauss", None),
    ):
        # model: ("gauss", None) GaussILRMA | ("t", dof) TILRMA | ("ggd", beta) GGDILRMA
        assert model[0] in ("gauss", "t", "ggd")
        self.model = model
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
        T, V = self.basis, self.activation
        TV = T @ V
        numer, expo = self._mm_numerator(self._current_output(), TV)
        num = np.sum(V[:, None, :, :] * numer[:, :, None, :], axis=3)
        den = np.sum(V[:, None, :, :] / TV[:, :, None, :], axis=3)
        self.basis = sp.floor(((num / den) ** expo) * T, self.flooring)

    def _mm_numerator(self, Y, TV):
        """Per-(n,i,j) factor of the MM numerator and the exponent of the ratio.

        Gauss: |y|^2 / TV^((p+2)/p), p/(p+2)                 ref: ssspy/bss/ilrma.py:1116-1125
        t    : |y|^2 / (R~ TV), p/(p+2)                      ref: :2505-2518
        GGD  : (beta/2) |y|^beta / TV^((beta+p)/p), p/(beta+p)  ref: :3810-3821
        """
        p = self.domain
        kind, param = self.model
        Y2 = np.abs(Y) ** 2
        if kind == "gauss":
            return Y2 / TV ** ((p + 2) / p), p / (p + 2)
        if kind == "t":
            nu_nu2 = param / (param + 2)
            R_tilde = nu_nu2 * TV ** (2 / p) + (1 - nu_nu2) * Y2
            return Y2 / (R_tilde * TV), p / (p + 2)
        beta = param
        return (beta / 2) * np.abs(Y) ** beta / TV ** ((beta + p) / p), p / (beta + p)

    def _spatial_weight(self, Y):
        """varphi = 1 / R~.  ref: ssspy/bss/ilrma.py:1494-1498 (Gauss), :2915-2935 (t), :3987-4011 (GGD)."""
        p = self.domain
        kind, param = self.model
        TV = self.basis @ self.activation
        if kind == "gauss":
            return 1 / TV ** (2 / p)
        if kind == "t":
            nu_nu2 = param / (param + 2)
            return 1 / (nu_nu2 * TV ** (2 / p) + (1 - nu_nu2) * np.abs(Y) ** 2)
        beta = param
        Y2b = sp.floor(np.abs(Y) ** (2 - beta), self.flooring)
        return 1 / ((2 / beta) * Y2b * TV ** (beta / p))

    def update_activation(self):
        """ref: ssspy/bss/ilrma.py:1130-1204 (update_activation_mm, no partitioning)."""
        T, V = self.basis, self.activation
        TV = T @ V
        numer, expo = self._mm_numerator(self._current_output(), TV)
        num = np.sum(T[:, :, :, None] * numer[:, :, None, :], axis=1)
        den = np.sum(T[:, :, :, None] / TV[:, :, None, :], axis=1)
        self.activation = sp.floor(((num / den) ** expo) * V, self.flooring)

    def update_spatial(self):
        """ref: ssspy/bss/ilrma.py:1440-1507 (IP1), :1635-1696 (ISS1)."""
        varphi = self._spatial_weight(self._current_output())
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
        kind, param = self.model
        if kind == "gauss":
            loss = Y2 / (TV ** (2 / p)) + (2 / p) * np.log(TV)
        elif kind == "t":  # ref: ssspy/bss/ilrma.py:3301-3305
            loss = (1 + param / 2) * np.log(1 + (2 / param) * Y2 / (TV ** (2 / p))) + (2 / p) * np.log(TV)
        else:  # ref: ssspy/bss/ilrma.py:4377-4381
            loss = np.abs(Y) ** param / TV ** (param / p) + (2 / p) * np.log(TV)
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
