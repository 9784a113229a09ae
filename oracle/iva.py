"""Oracle: auxiliary-function IVA (IP1 / ISS1) with Laplace and Gauss contrasts.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates ``AuxIVA`` / ``AuxLaplaceIVA`` / ``AuxGaussIVA`` of the reference for
``spatial_algorithm in {"IP","IP1","ISS","ISS1"}`` (SURVEY.md section 8 rows
a12-a15, Appendix C).  The contrast is named by a string instead of a pair of
Python closures: ``"laplace"`` (G = 2r, G' = 2) or ``"gauss"``
(G = F log(alpha) + r^2/alpha, G' = 2r/alpha, alpha refreshed every iteration); ``("power", p)``
is the user-closure case of the generic ``AuxIVA`` class, G_R(r) = r^p, G' = p r^(p-1).
"""

import numpy as np

from . import spatial as sp
from .ipa import update_by_ipa


class AuxIVAOracle:
    """ref: ssspy/bss/iva.py:1403-2214 (AuxIVA), :2976-3128 (Laplace), :3131-3473 (Gauss)."""

    def __init__(
        self,
        spatial_algorithm="IP",
        contrast="laplace",
        flooring=sp.DEFAULT_FLOOR,
        scale_restoration=True,
        record_loss=True,
        reference_id=0,
    ):
        assert spatial_algorithm in ("IP", "IP1", "IP2", "ISS", "ISS1", "ISS2", "IPA")
        # IPA keyword arguments of the reference (defaults: ssspy/bss/ilrma.py:749, iva.py:1579)
        self.lqpqm_normalization, self.newton_iter = True, 1
        assert contrast in ("laplace", "gauss") or (type(contrast) is tuple and contrast[0] == "power")
        self.pairs = None
        self.spatial_algorithm = spatial_algorithm
        self.contrast = contrast
        self.flooring = flooring
        self.scale_restoration = scale_restoration
        self.record_loss = record_loss
        self.reference_id = reference_id
        self.loss = [] if record_loss else None

    @property
    def uses_filter(self):
        return self.spatial_algorithm in ("IP", "IP1", "IP2")

    def reset(self, X, demix_filter=None):
        """ref: ssspy/bss/iva.py:138-169 (IVABase._reset), :1687-1697, :3304-3317."""
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
        if not self.uses_filter:
            self.demix_filter = None
        if self.contrast == "gauss":
            self.variance = np.ones((N, T))

    # contrast functions ------------------------------------------------------
    def contrast_fn(self, Y):
        """ref: ssspy/bss/iva.py:3093-3103 (Laplace), :3256-3271 (Gauss)."""
        r = np.linalg.norm(Y, axis=1)
        if self.contrast == "laplace":
            return 2 * r
        if type(self.contrast) is tuple:
            return r ** self.contrast[1]
        return self.n_bins * np.log(self.variance) + (r**2) / self.variance

    def d_contrast_fn(self, r, variance=None):
        """ref: ssspy/bss/iva.py:3105-3115 (Laplace), :3273-3289 (Gauss)."""
        if self.contrast == "laplace":
            return 2 * np.ones_like(r)
        if type(self.contrast) is tuple:
            return self.contrast[1] * r ** (self.contrast[1] - 1)
        return 2 * r / (self.variance if variance is None else variance)

    def _current_output(self):
        if self.demix_filter is None:
            return self.output
        return sp.separate(self.input, self.demix_filter)

    def update_once(self):
        """ref: ssspy/bss/iva.py:1699-1793 (IP1), :1917-1966 (ISS1), :3319-3337, :3465-3473."""
        Y = self._current_output()
        if self.contrast == "gauss":
            self.variance = np.mean(np.abs(Y) ** 2, axis=1)
        N = self.n_sources
        if self.spatial_algorithm == "IP2":
            # ref: ssspy/bss/iva.py:1795-1915 (AuxIVA), :3339-3463 (AuxGaussIVA): the weights are
            # recomputed for every pair from the current filters
            W = self.demix_filter.copy()
            pairs = self.pairs if self.pairs is not None else sp.sequential_pairs(N)
            for m, n in pairs:
                Y_mn = sp.separate(self.input, W[:, (m, n), :])
                norm = np.linalg.norm(Y_mn, axis=1)
                if self.contrast == "gauss":
                    dG = self.d_contrast_fn(norm, variance=self.variance[(m, n), :])
                else:
                    dG = self.d_contrast_fn(norm)
                U_mn = sp.weighted_covariance(self.input, dG / sp.floor(2 * norm, self.flooring))
                W[:, (m, n), :] = sp.update_by_ip2_one_pair(W, U_mn, (m, n), self.flooring)
            self.demix_filter = W
            return
        r = np.linalg.norm(Y, axis=1)  # (N, T)
        weight = self.d_contrast_fn(r) / sp.floor(2 * r, self.flooring)
        if self.spatial_algorithm == "ISS2":
            # ref: ssspy/bss/iva.py:1968-2066
            pairs = self.pairs if self.pairs is not None else sp.sequential_pairs(N)
            self.output = sp.update_by_iss2(Y, weight[:, None, :], self.flooring, pairs)
            return
        if self.spatial_algorithm == "IPA":
            # ref: ssspy/bss/iva.py:2068-2175
            self.output = update_by_ipa(Y, weight[:, None, :], self.flooring,
                                        self.lqpqm_normalization, self.newton_iter)
            return
        if self.uses_filter:
            U = sp.weighted_covariance(self.input, weight)
            self.demix_filter = sp.update_by_ip1(self.demix_filter, U, self.flooring)
        else:
            self.output = sp.update_by_iss1(Y, weight[:, None, :], self.flooring)

    def compute_loss(self):
        """ref: ssspy/bss/iva.py:200-222 (filter path), :2177-2192 (ISS path)."""
        if self.demix_filter is None:
            Y = self.output
            W = sp.demix_from_output(Y, self.input)
        else:
            W = self.demix_filter
            Y = sp.separate(self.input, W)
        G = self.contrast_fn(Y)
        loss = np.sum(np.mean(G, axis=1), axis=0) - 2 * np.sum(sp.logdet(W), axis=0)
        return loss.item()

    def restore_scale(self):
        """Projection back (default) or the minimal distortion principle.  ref: ssspy/bss/iva.py:238-281, :2194-2214."""
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
        """ref: ssspy/bss/iva.py:1637-1672 and ssspy/bss/base.py:48-77."""
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
