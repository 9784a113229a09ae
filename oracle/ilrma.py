"""Oracle: ILRMA (Gauss, Student-t and GGD source models).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``GaussILRMA`` / ``TILRMA`` / ``GGDILRMA`` of the reference (ssspy/bss/ilrma.py) in its own
expression structure: ``source_algorithm`` "MM" and "ME", with and without ``partitioning`` (latent
variables), every ``spatial_algorithm`` (IP1 / IP2 / ISS1 / ISS2 through ``oracle/spatial.py``, IPA
through ``oracle/ipa.py``), power and projection-back normalisation, projection-back and
minimal-distortion scale restoration.  The
per-iteration order is Appendix A of SURVEY.md: basis, activation, spatial, normalise.  Pinned by
tests/test_oracle_golden.py against fixtures the reference generated.
"""

import numpy as np

from . import spatial as sp
from .ipa import update_by_ipa


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
        source_algorithm="MM",
        partitioning=False,
    ):
        # model: ("gauss", None) GaussILRMA | ("t", dof) TILRMA | ("ggd", beta) GGDILRMA
        assert model[0] in ("gauss", "t", "ggd")
        self.model = model
        # "ME": same numerator / denominator as MM, exponent 1, domain 2 only, Gauss and t models
        # (ref: ssspy/bss/ilrma.py:1249-1401, :2659-2830)
        assert source_algorithm in ("MM", "ME")
        assert source_algorithm == "MM" or (domain == 2 and model[0] != "ggd")
        self.source_algorithm = source_algorithm
        # partitioning: shared basis (F, K) / activation (K, T) assigned to sources by the latent
        # variables Z (N, K), columns of Z sum to one (ref: ssspy/bss/ilrma.py:201-270, :297-327)
        self.partitioning = partitioning
        assert spatial_algorithm in ("IP", "IP1", "IP2", "ISS", "ISS1", "ISS2", "IPA")
        # IPA keyword arguments of the reference (defaults: ssspy/bss/ilrma.py:749, iva.py:1579)
        self.lqpqm_normalization, self.newton_iter = True, 1
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
    def reset(self, X, basis=None, activation=None, demix_filter=None, latent=None):
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
        lead = () if self.partitioning else (N,)
        if self.partitioning:
            # the reference draws latent first, then basis, then activation (ilrma.py:230-251)
            if latent is None:
                latent = self.rng.random((N, self.n_basis))
                latent = sp.floor(latent / latent.sum(axis=0), self.flooring)
            else:
                latent = latent.copy()
            self.latent = latent
        if basis is None:
            basis = sp.floor(self.rng.random(lead + (F, self.n_basis)), self.flooring)
        else:
            basis = basis.copy()
        if activation is None:
            activation = sp.floor(self.rng.random(lead + (self.n_basis, T)), self.flooring)
        else:
            activation = activation.copy()
        self.basis, self.activation = basis, activation
        if not self.uses_filter:
            self.demix_filter = None

    def _current_output(self):
        if self.demix_filter is None:
            return self.output
        return sp.separate(self.input, self.demix_filter)

    def _tv(self):
        """Source model R_nij.  ref: ssspy/bss/ilrma.py:297-327 (reconstruct_nmf)."""
        if self.partitioning:
            return np.einsum("nk,ik,kj->nij", self.latent, self.basis, self.activation)
        return self.basis @ self.activation

    def _expo(self, expo):
        return 1 if self.source_algorithm == "ME" else expo

    # -- one iteration -----------------------------------------------------
    def update_latent(self):
        """z_nk <- z_nk (sum_ij t v numer / sum_ij t v / R)^e, columns renormalised.

        ref: ssspy/bss/ilrma.py:1007-1049 (MM), :1206-1247 (ME), :2384-2432, :3698-3743.
        """
        Z, T, V = self.latent, self.basis, self.activation
        R = self._tv()
        numer, expo = self._mm_numerator(self._current_output(), R)
        num = np.einsum("ik,kj,nij->nk", T, V, numer)
        den = np.einsum("ik,kj,nij->nk", T, V, 1 / R)
        Z = ((num / den) ** self._expo(expo)) * Z
        self.latent = Z / Z.sum(axis=0)

    def update_basis(self):
        """ref: ssspy/bss/ilrma.py:1051-1128 (update_basis_mm)."""
        T, V = self.basis, self.activation
        TV = self._tv()
        numer, expo = self._mm_numerator(self._current_output(), TV)
        if self.partitioning:
            Z = self.latent
            num = np.einsum("nk,kj,nij->ik", Z, V, numer)
            den = np.einsum("nk,kj,nij->ik", Z, V, 1 / TV)
        else:
            num = np.sum(V[:, None, :, :] * numer[:, :, None, :], axis=3)
            den = np.sum(V[:, None, :, :] / TV[:, :, None, :], axis=3)
        self.basis = sp.floor(((num / den) ** self._expo(expo)) * T, self.flooring)

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
        TV = self._tv()
        if kind == "gauss":
            return 1 / TV ** (2 / p)
        if kind == "t":
            nu_nu2 = param / (param + 2)
            return 1 / (nu_nu2 * TV ** (2 / p) + (1 - nu_nu2) * np.abs(Y) ** 2)
        beta = param
        Y2b = sp.floor(np.abs(Y) ** (2 - beta), self.flooring)
        return 1 / ((2 / beta) * Y2b * TV ** (beta / p))

    def update_activation(self):
        """ref: ssspy/bss/ilrma.py:1130-1204 (update_activation_mm)."""
        T, V = self.basis, self.activation
        TV = self._tv()
        numer, expo = self._mm_numerator(self._current_output(), TV)
        if self.partitioning:
            Z = self.latent
            num = np.einsum("nk,ik,nij->kj", Z, T, numer)
            den = np.einsum("nk,ik,nij->kj", Z, T, 1 / TV)
        else:
            num = np.sum(T[:, :, :, None] * numer[:, :, None, :], axis=1)
            den = np.sum(T[:, :, :, None] / TV[:, :, None, :], axis=1)
        self.activation = sp.floor(((num / den) ** self._expo(expo)) * V, self.flooring)

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
        elif self.spatial_algorithm == "IPA":
            # ref: ssspy/bss/ilrma.py:1794-1908
            self.output = update_by_ipa(self.output, varphi, self.flooring,
                                        self.lqpqm_normalization, self.newton_iter)
        elif self.uses_filter:
            U = sp.weighted_covariance(self.input, varphi)
            self.demix_filter = sp.update_by_ip1(self.demix_filter, U, self.flooring)
        else:
            self.output = sp.update_by_iss1(self.output, varphi, self.flooring)

    def normalize_by_projection_back(self):
        """ref: ssspy/bss/ilrma.py:446-522."""
        p = self.domain
        if self.demix_filter is None:
            Yf = self.output.transpose(1, 0, 2)
            Xf = self.input.transpose(1, 0, 2)
            YH = Yf.transpose(0, 2, 1).conj()
            scale = ((Xf @ YH) @ np.linalg.inv(Yf @ YH))[..., self.reference_id, :]  # (F, N)
            self.output = (Yf * scale[..., None]).swapaxes(-3, -2)
        else:
            scale = np.linalg.inv(self.demix_filter)[:, self.reference_id, :]
            self.demix_filter = self.demix_filter * scale[:, :, None]
        self.basis = self.basis * (np.abs(scale.T) ** p)[:, :, None]

    def normalize(self):
        """ref: ssspy/bss/ilrma.py:365-444 (normalize_by_power)."""
        if self.normalization == "projection_back":
            return self.normalize_by_projection_back()
        p = self.domain
        Y = self._current_output()
        psi = sp.floor(np.sqrt(np.mean(np.abs(Y) ** 2, axis=(-2, -1))), self.flooring)
        if self.partitioning:
            Z_psi = self.latent / (psi[:, None] ** p)
            scale = Z_psi.sum(axis=0)
            self.basis = self.basis * scale[None, :]
            self.latent = Z_psi / scale
        else:
            self.basis = self.basis / (psi[:, None, None] ** p)
        if self.demix_filter is None:
            self.output = Y / psi[:, None, None]
        else:
            self.demix_filter = self.demix_filter / psi[None, :, None]

    def update_once(self):
        """ref: ssspy/bss/ilrma.py:900-922."""
        if self.partitioning:
            self.update_latent()
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
        TV = self._tv()
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
        """Projection back (default) or the minimal distortion principle.  ref: ssspy/bss/ilrma.py:538-579, :1969-1989."""
        mdp = self.scale_restoration == "minimal_distortion_principle"
        if self.demix_filter is None:
            fn = sp.minimal_distortion_output if mdp else sp.projection_back_output
            self.output = fn(self.output, self.input, self.reference_id)
        elif mdp:
            Y = sp.minimal_distortion_output(sp.separate(self.input, self.demix_filter),
                                             self.input, self.reference_id)
            self.output, self.demix_filter = Y, sp.demix_from_output(Y, self.input)
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
