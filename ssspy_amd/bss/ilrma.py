"""Independent low-rank matrix analysis (ILRMA) on MI355X.

Drop-in separator classes for the reference's ``ssspy.bss.ilrma`` hot path
(ssspy/bss/ilrma.py): same constructor arguments, ``__call__`` / ``update_once`` /
``compute_loss`` protocol and state attributes, with every per-iteration computation done
by the HIP kernels of ``libssspy_amd.so``.  Built here: ``GaussILRMA``, ``TILRMA`` and
``GGDILRMA`` with ``spatial_algorithm in {"IP", "IP1", "IP2", "ISS", "ISS1", "ISS2", "IPA"}``,
``source_algorithm in {"MM", "ME"}``, with and without ``partitioning``, power and projection-back
normalisation, projection-back and minimal-distortion scale restoration.  What is not built
raises ``NotImplementedError`` naming the step (never a CPU fallback).

Extension over the reference: ``input`` may be 4-D ``(n_mixtures, n_channels, n_bins,
n_frames)``; the mixtures are independent and every attribute then carries the same
leading axis (``loss`` entries become arrays of length ``n_mixtures``).
"""

import functools
from typing import Callable, Iterable, List, Optional, Tuple, Union

import numpy as np

from .. import _device as dv
from .. import _lib, _ops, _routes
from ..special.flooring import identity, max_flooring
from ..utils.flooring import choose_flooring_fn, device_flooring, host_floor, require_device_floor
from ..utils.select_pair import resolve_pairs, sequential_pair_selector
from ._device_state import DeviceStateMixin, Synced
from .base import IterativeMethodBase

__all__ = ["GaussILRMA", "TILRMA", "GGDILRMA"]

spatial_algorithms = ["IP", "IP1", "IP2", "ISS", "ISS1", "ISS2", "IPA"]
source_algorithms = ["MM", "ME"]
EPS = 1e-10

_IP1 = ("IP", "IP1")
_ISS1 = ("ISS", "ISS1")
_IP2 = ("IP2",)
_ISS2 = ("ISS2",)
_IPA = ("IPA",)
_PROJECTION_BACK = ("projection_back",)
_MDP = ("minimal_distortion_principle",)


class ILRMABase(DeviceStateMixin, IterativeMethodBase):
    """State handling shared by the ILRMA variants (ref: ssspy/bss/ilrma.py:32-579)."""

    demix_filter = Synced(dv.c128)
    output = Synced(dv.c128)
    basis = Synced(dv.f64)
    activation = Synced(dv.f64)
    latent = Synced(dv.f64)

    def __init__(
        self,
        n_basis: int,
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        callbacks=None,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
    ) -> None:
        super().__init__(callbacks=callbacks, record_loss=record_loss)
        self.n_basis = n_basis
        self.partitioning = partitioning
        self.flooring_fn = identity if flooring_fn is None else flooring_fn
        self.input = None
        self.scale_restoration = scale_restoration
        if reference_id is None and scale_restoration:
            raise ValueError("Specify 'reference_id' if scale_restoration=True.")
        self.reference_id = reference_id
        self.rng = np.random.default_rng() if rng is None else rng

    # -- reset ------------------------------------------------------------------------
    def _reset(self, flooring_fn="self", **kwargs) -> None:
        """ref: ssspy/bss/ilrma.py:151-199."""
        assert self._has_input(), "Specify data!"
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        for key, value in kwargs.items():
            setattr(self, key, value)
        B, N, F, T = self._X.shape
        self.n_sources, self.n_channels = N, N
        self.n_bins, self.n_frames = F, T
        if not self._state_has("demix_filter"):
            W = np.tile(np.eye(N, dtype=np.complex128), self._lead() + (F, 1, 1))
            self.demix_filter = W
        elif not self._state_is_none("demix_filter"):
            # private copy: the injected array is never written to
            self.demix_filter = np.array(self.demix_filter, dtype=np.complex128, copy=True)
        if self._state_is_none("demix_filter"):
            # the reference calls separate(X, None) here, which fails as well
            raise ValueError("demix_filter=None cannot be given at reset.")
        self._state_set_dev("output", _ops.separate(self._X, self._state_dev("demix_filter")))
        self._init_nmf(flooring_fn=flooring_fn, rng=self.rng)
        self._floor = device_flooring(flooring_fn, allow_host=True)
        K = self.n_basis
        self._ws, self._ws_bytes = _ops.ilrma_workspace(B, N, F, T, K, self._X.device)
        self._U = None

    def _init_nmf(self, flooring_fn="self", rng=None) -> None:
        """ref: ssspy/bss/ilrma.py:201-270.  With partitioning the basis (F, K) and activation (K, T)
        are shared and ``latent`` (N, K), columns summing to one, assigns them to the sources."""
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        if rng is None:
            rng = np.random.default_rng()
        N, F, T, K = self.n_sources, self.n_bins, self.n_frames, self.n_basis
        src = () if self.partitioning else (N,)
        if self.partitioning:
            # draw order of the reference: latent, then basis, then activation (ilrma.py:230-251)
            if not self._state_has("latent"):
                Z = rng.random(self._lead() + (N, K))
                self.latent = flooring_fn(Z / Z.sum(axis=-2, keepdims=True))
            else:
                self.latent = np.array(self.latent, dtype=np.float64, copy=True)
        if not self._state_has("basis"):
            self.basis = flooring_fn(rng.random(self._lead() + src + (F, K)))
        else:
            self.basis = np.array(self.basis, dtype=np.float64, copy=True)
        if not self._state_has("activation"):
            self.activation = flooring_fn(rng.random(self._lead() + src + (K, T)))
        else:
            self.activation = np.array(self.activation, dtype=np.float64, copy=True)
        if self.partitioning:
            B = self._X.shape[0]
            self._Teff = dv.empty((B, N, F, K), dv.f64, self._X.device)
            self._Vrep = dv.empty((B, N, K, T), dv.f64, self._X.device)

    # -- operators --------------------------------------------------------------------
    def separate(self, input: np.ndarray, demix_filter: np.ndarray) -> np.ndarray:
        """y_ij = W_i x_ij (ref: ssspy/bss/ilrma.py:272-295); NumPy in, NumPy out."""
        batched = input.ndim == 4
        X = dv.to_device(input if batched else input[None], dtype=np.complex128)
        W = dv.to_device(demix_filter if batched else demix_filter[None], dtype=np.complex128)
        Y = dv.to_host(_ops.separate(X, W))
        return Y if batched else Y[0]

    def reconstruct_nmf(self, basis, activation, latent=None) -> np.ndarray:
        """R = T V (ref: ssspy/bss/ilrma.py:297-327); host-side convenience, not on the hot path."""
        if latent is not None:
            return np.einsum("...nk,...ik,...kj->...nij", latent, basis, activation)
        return basis @ activation

    def _nmf_pair(self):
        """Device (basis, activation) in the per-source layout every kernel takes: the state itself,
        or with partitioning the expansion (z_nk t_ik, v_kj) rebuilt from the current parameters."""
        if not self.partitioning:
            return self._state_dev("basis"), self._state_dev("activation")
        _ops.ilrma_partition_expand(self._state_dev("basis"), self._state_dev("activation"),
                                    self._state_dev("latent"), self._Teff, self._Vrep)
        return self._Teff, self._Vrep

    def _resolve_floor(self, flooring_fn):
        if type(flooring_fn) is str and flooring_fn == "self":
            return self._floor
        return device_flooring(choose_flooring_fn(flooring_fn, method=self), allow_host=True)

    def _uses_filter(self) -> bool:
        return not self._state_is_none("demix_filter")

    def _implied_filter(self):
        return None

    # -- scale restoration ------------------------------------------------------------------
    def restore_scale(self) -> None:
        """ref: ssspy/bss/ilrma.py:538-555."""
        scale_restoration = self.scale_restoration
        assert scale_restoration, "Set self.scale_restoration=True."
        if type(scale_restoration) is bool:
            scale_restoration = _PROJECTION_BACK[0]
        if scale_restoration in _PROJECTION_BACK:
            self.apply_projection_back()
        elif scale_restoration in _MDP:
            self.apply_minimal_distortion_principle()
        else:
            raise ValueError("{} is not supported for scale restoration.".format(scale_restoration))

    def apply_projection_back(self) -> None:
        """ref: ssspy/bss/ilrma.py:557-565, :1969-1979; algorithm/projection_back.py:87-121."""
        assert self.scale_restoration, "Set self.scale_restoration=True."
        info = self._info_tensor()
        if self._uses_filter():
            W = self._state_dev("demix_filter")
            _ops.projection_back_filter(W, self.reference_id, info)
            self._state_touch("demix_filter")
            self._state_set_dev("output", _ops.separate(self._X, W))
        elif self._implied_filter() is not None and self.reference_id is not None:
            # the same scales from the filters the output state implies: one pass instead of four
            W = self._implied_filter().clone()
            _ops.projection_back_filter(W, self.reference_id, info)
            self._state_set_dev("output", _ops.separate(self._X, W))
            self._implied = (W, self._state_rev("output"))
        else:
            Y = self._state_dev("output")
            XY = _ops.cross_covariance(self._X, Y)
            YY = _ops.cross_covariance(Y, Y)
            G = _ops.projection_back_scale(XY, YY, self.reference_id, info)
            _ops.separate(Y, G, out=Y)
            self._state_touch("output")

    def apply_minimal_distortion_principle(self) -> None:
        """Per (bin, source) scale z = <y, x_ref> / <y, y>, output conj(z) y; with a filter state the
        filter is re-fitted as Y X^H (X X^H)^-1 like the reference.
        ref: ssspy/bss/ilrma.py:567-579, :1981-1989; algorithm/minimal_distortion_principle.py:6-43."""
        assert self.scale_restoration, "Set self.scale_restoration=True."
        filt = self._uses_filter()
        if self.reference_id is None:
            # reachable only by clearing the attribute after construction; as in the reference the
            # estimate gains a leading channel axis (minimal_distortion_principle.py:34-35) and a
            # filter state cannot take that shape
            if filt:
                raise ValueError("reference_id=None needs the output state (ISS / IPA), not filters.")
            from ..algorithm import minimal_distortion_principle as _mdp

            Y, X = dv.to_host(self._state_dev("output")), dv.to_host(self._X)
            out = np.stack([_mdp(y, reference=x, reference_id=None) for y, x in zip(Y, X)])
            self.output = out if self._batched else out[0]
            return
        if filt:
            Y = _ops.separate(self._X, self._state_dev("demix_filter"))
        else:
            Y = self._state_dev("output")
        G = _ops.mdp_scale(_ops.cross_covariance(Y, self._X), _ops.cross_covariance(Y, Y),
                           self.reference_id)
        _ops.separate(Y, G, out=Y)
        if filt:
            W = _ops.demix_from_covariance(_ops.cross_covariance(Y, self._X), self._C(),
                                           self._info_tensor())
            self._state_set_dev("demix_filter", W)
            self._state_set_dev("output", Y)
        else:
            self._state_touch("output")

    def _host_loss(self, data, logdet_sum):
        """loss = data term - 2 sum_i log|det W_i|, combined on the host (B numbers each)."""
        self._check_device_errors()
        values = dv.to_host(data) - 2.0 * dv.to_host(logdet_sum)
        return values.copy() if self._batched else values[0].item()


class _MMILRMA(ILRMABase):
    """Everything the three source models share: the iteration, the MM updates, the spatial
    updates, normalisation, loss.  The model enters only as ``_model`` = (SSSPY_SOURCE_*, param),
    which the kernels branch on (include/ssspy_amd.h)."""

    _base_model = _ops.GAUSS

    @property
    def _model(self):
        kind, param = self._base_model
        if self.source_algorithm == "ME":
            kind |= _lib.SOURCE_ME
        return (kind, param)
    _name = "ILRMA"

    def _configure(self, spatial_algorithm, source_algorithm, domain, partitioning, normalization,
                   pair_selector) -> None:
        self.spatial_algorithm = spatial_algorithm
        self.source_algorithm = source_algorithm
        self.domain = domain
        self.normalization = normalization
        if pair_selector is None:
            if spatial_algorithm in ["IP2", "ISS2"]:
                self.pair_selector = sequential_pair_selector
        else:
            self.pair_selector = pair_selector
        device_flooring(self.flooring_fn, allow_host=True)  # (translation errors, if any, surface before any upload)

    def __call__(
        self, input: np.ndarray, n_iter: int = 100, initial_call: bool = True, **kwargs
    ) -> np.ndarray:
        """Separate a frequency-domain multichannel mixture.

        Args:
            input: ``(n_channels, n_bins, n_frames)`` complex (or 4-D batch of such).
            n_iter: number of ``update_once`` rounds.
            initial_call: run loss/callbacks once before iterating.
            kwargs: attributes to inject before initialisation (``basis``, ``activation``,
                ``demix_filter``), as in the reference.

        Returns:
            Separated spectrograms, same shape as ``input``.
        """
        self._bind_input(input)
        self._reset(flooring_fn=self.flooring_fn, **kwargs)
        if not (self._iterate_with_deferred_loss(int(n_iter), initial_call)
                or self._iterate_with_resident_terms(int(n_iter), initial_call)):
            IterativeMethodBase.__call__(self, n_iter=n_iter, initial_call=initial_call)
        if self.scale_restoration:
            self.restore_scale()
        if self._uses_filter():
            self._state_set_dev("output", _ops.separate(self._X, self._state_dev("demix_filter")))
        return self._final_output()

    def __repr__(self) -> str:
        s = "{}(n_basis={}{}, spatial_algorithm={}, source_algorithm={}, domain={}".format(
            type(self).__name__, self.n_basis, self._repr_model(), self.spatial_algorithm,
            self.source_algorithm, self.domain
        )
        s += ", partitioning={}, normalization={}, scale_restoration={}, record_loss={}".format(
            self.partitioning, self.normalization, self.scale_restoration, self.record_loss
        )
        if self.scale_restoration:
            s += ", reference_id={}".format(self.reference_id)
        return s + ")"

    def _repr_model(self) -> str:
        return ""

    def _reset(self, flooring_fn="self", **kwargs) -> None:
        """ref: ssspy/bss/ilrma.py:875-898."""
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        super()._reset(flooring_fn=flooring_fn, **kwargs)
        self._logdet_cache = None
        self._implied = None
        self._amp_reset()
        if (self.spatial_algorithm in ["ISS", "ISS1", "ISS2", "IPA"] and self._X.shape[1] <= 4
                and self._base_model[0] == _lib.SOURCE_GAUSS):
            # the filters the output state implies (output = W x), kept next to it: see
            # _update_spatial_model_implied().  Up to 4 sources, where the passes over (X, W) are
            # the tuned IP1 ones (8 sources, 16 mixtures: ISS2 4.0 against 2.9 ms on Y); Gauss model
            # only: W U W^H rounds like eps |W|^2 |U| where the direct sum rounds like eps |y|^2, and
            # the t / GGD weights 1 / |y|^(2 - beta) feed that back (the GGD ISS2 golden: 4e-10 on Y,
            # 4e-7 through the filters after 10 iterations -- both started from 1e-15 at iteration 2).
            # How far the product can round is measured by every launch that forms it and the route
            # is left where that passes its bound (_amp_exceeded; rounds 5's fence of 16 frames per
            # source is gone: the draw it was fitted to -- 11 frames for 4 sources, 1e-7 of the
            # oracle after 8 ISS2 iterations through the filters -- now leaves after the second)
            self._implied = (self._state_dev("demix_filter").clone(), self._state_rev("output"))
        if self.spatial_algorithm in ["ISS", "ISS1", "ISS2", "IPA"] and not self.record_loss:
            self.demix_filter = None  # (nothing reads the log-determinant: no tracker)
        elif self.spatial_algorithm in ["ISS", "ISS1", "ISS2", "IPA"]:
            # sum_i log|det W_i| of the filters the ISS state stops carrying, as (tensor, revision of
            # `output` it describes); the fused sweep and the power normalisation move it along, so
            # compute_loss() need not rebuild W from Y X^H (see AuxIVA._reset)
            self._logdet_cache = (_ops.sum_logdet(self._state_dev("demix_filter")),
                                  self._state_rev("output"))
            self.demix_filter = None

    def _tracked_logdet(self):
        """The tracked sum_i log|det W_i| if it describes the current output, else None."""
        cache = getattr(self, "_logdet_cache", None)
        if cache is not None and cache[1] == self._state_rev("output"):
            return cache[0]
        return None

    def _restamp_logdet(self, tracked) -> None:
        self._logdet_cache = None if tracked is None else (tracked, self._state_rev("output"))

    # -- the loop with the loss as a by-product ----------------------------------------------
    def _fused_ip1(self) -> bool:
        """True when update_once() is the single fused C-ABI call (stock methods, IP1 on filters)."""
        return (self.spatial_algorithm in _IP1 and self._uses_filter() and self._is_stock()
                and self._power_normalization_or_off() and not self.partitioning)

    def _iterate_with_deferred_loss(self, n_iter: int, initial_call: bool) -> bool:
        """``record_loss=True`` without a fourth pass over the mixture per iteration.

        The loss after iteration t is the negative log-likelihood of the state the basis pass of
        iteration t + 1 reads, and that pass forms |y|^2 and R anyway: the fused update leaves the
        data term (and the log-determinants, taken before IP1 rewrites the filters) as by-products,
        so only the loss after the LAST iteration needs the dedicated pass.  The list is assembled
        at the end from one download -- which is why this path is taken only when nothing can look
        at ``self.loss`` in between: no callbacks, and ``update_once`` / ``compute_loss`` not
        overridden.  Otherwise (returns False) the reference's loop order runs unchanged.
        ref: ssspy/bss/base.py:68-77, ssspy/bss/ilrma.py:1910-1967."""
        cls = type(self)
        if not (self.record_loss and not self.callbacks and n_iter > 0 and self._fused_ip1()
                and host_floor(self._floor) is None
                and cls.update_once is _MMILRMA.update_once
                and cls.compute_loss is _MMILRMA.compute_loss):
            return False
        B, N, F, T = self._X.shape
        dev = self._X.device
        if not _ops.ilrma_deferred_loss_supported(N, F, T, self.n_basis, float(self.domain), self._model):
            return False  # shapes / models outside the tuned kernels have no such by-product
        if self._U is None:
            self._U = dv.empty((B, F, N, N, N), dv.c128, dev)
        data = dv.zeros((n_iter + 1, B), dv.f64, dev)
        logdet = dv.zeros((n_iter + 1, B), dv.f64, dev)
        C = self._C() if self.normalization else None
        W, Tb, Vb = (self._state_dev(k) for k in ("demix_filter", "basis", "activation"))
        args = (self._X, C, W, Tb, Vb, self._U, float(self.domain), bool(self.normalization),
                self._floor, self._ws, self._ws_bytes, self._info_tensor())
        # the data terms of all iterations as raw per-wave shares in ONE zeroed array, folded once at
        # the end (round 5: a memset, a counter memset and a fold launch per iteration before)
        nslots = _ops.ilrma_deferred_loss_slots(B, N, F, T, self.n_basis, float(self.domain),
                                                self._model)
        stride = (n_iter + 1) * B
        slots = None
        if nslots and stride < 2 ** 31 and nslots * stride * 8 <= (1 << 28):  # (<= 256 MB)
            slots = dv.zeros((nslots, stride), dv.f64, dev)
        flat = slots.reshape(-1) if slots is not None else None
        # the log-determinants the same way (round 6): one share per mixture (the finished sum) or,
        # for a handful of mixtures, one per 16-bin tile of the IP1 kernel, folded at the end
        nld = _ops.ilrma_deferred_logdet_slots(B, N, F, T, self.n_basis, float(self.domain),
                                               self._model) if slots is not None else 1
        ld = dv.zeros((nld, stride), dv.f64, dev) if nld > 1 else logdet
        ld_flat = ld.reshape(-1)
        for t in range(n_iter):
            if t == 0 and not initial_call:
                # the reference records nothing before the first iteration in this case
                _ops.ilrma_ip1_update(*args, model=self._model)
            elif slots is not None:
                _ops.ilrma_ip1_update_loss_slots(*args, flat[t * B:], stride, ld_flat[t * B:],
                                                 model=self._model)
            elif not _ops.ilrma_ip1_update_deferred_loss(*args, data[t], logdet[t],
                                                         model=self._model):
                raise RuntimeError("deferred loss unavailable although reported as supported")
        if slots is not None:
            _ops.fold_scalar_slots(slots, stride, nslots, data.reshape(-1))
            if nld > 1:
                _ops.fold_scalar_slots(ld, stride, nld, logdet.reshape(-1))
        for name in ("demix_filter", "basis", "activation"):
            self._state_touch(name)
        _ops.ilrma_loss_data(self._X, W, Tb, Vb, float(self.domain), out=data[n_iter],
                             model=self._model)
        _ops.sum_logdet(W, out=logdet[n_iter])
        self._check_device_errors()
        values = dv.to_host(data) - 2.0 * dv.to_host(logdet)
        if not initial_call:
            values = values[1:]
        self.loss.extend(v.copy() if self._batched else v[0].item() for v in values)
        return True

    def _iterate_with_resident_terms(self, n_iter: int, initial_call: bool) -> bool:
        """``record_loss=True`` for every iteration the deferred form above does not serve (ISS /
        ISS2 / IP2 / IPA, heavy-tailed models, ...) without a host round trip per iteration
        (round 6): ``compute_loss()`` downloads two numbers per mixture and the wait for them
        drains the queue after every iteration.  Here the terms stay in HBM and one download at
        the end assembles the list.  Only with the library's own ``update_once`` /
        ``compute_loss`` and no callbacks; otherwise (returns False) the reference's loop runs
        unchanged.  ref: ssspy/bss/base.py:68-77, ssspy/bss/ilrma.py:1910-1967."""
        cls = type(self)
        if not (self.record_loss and not self.callbacks and n_iter > 0
                and cls.update_once is _MMILRMA.update_once
                and cls.compute_loss is _MMILRMA.compute_loss):
            return False
        B, dev = self._X.shape[0], self._X.device
        data = dv.zeros((n_iter + 1, B), dv.f64, dev)
        logdet = dv.zeros((n_iter + 1, B), dv.f64, dev)
        for t in range(n_iter + 1):
            if t > 0 or initial_call:
                self._loss_terms(data[t], logdet[t])
            if t < n_iter:
                self.update_once()
        self._check_device_errors()
        values = dv.to_host(data) - 2.0 * dv.to_host(logdet)
        if not initial_call:
            values = values[1:]
        self.loss.extend(v.copy() if self._batched else v[0].item() for v in values)
        return True

    # -- one iteration -----------------------------------------------------------------------
    def _is_stock(self) -> bool:
        cls = type(self)
        return all(
            getattr(cls, name) is getattr(_MMILRMA, name)
            for name in ("update_source_model", "update_spatial_model", "normalize",
                         "update_basis_mm", "update_activation_mm", "update_spatial_model_ip1",
                         "normalize_by_power", "update_source_model_me", "update_basis_me",
                         "update_activation_me", "update_source_model_mm")
        )

    def update_once(self, flooring_fn="self") -> None:
        """Source model (basis, activation), spatial model, normalisation.

        ref: ssspy/bss/ilrma.py:900-922.  With the stock methods and the IP1 path the whole
        iteration is one C-ABI call (five kernel launches on the current stream).
        """
        floor = self._resolve_floor(flooring_fn)
        if self._fused_ip1() and host_floor(floor) is None:
            B, N, F, T = self._X.shape
            if self._U is None:
                self._U = dv.empty((B, F, N, N, N), dv.c128, self._X.device)
            _ops.ilrma_ip1_update(
                self._X, self._C() if self.normalization else None,
                self._state_dev("demix_filter"), self._state_dev("basis"),
                self._state_dev("activation"), self._U, float(self.domain),
                bool(self.normalization), floor, self._ws, self._ws_bytes, self._info_tensor(),
                model=self._model,
            )
            for name in ("demix_filter", "basis", "activation"):
                self._state_touch(name)
            return
        self.update_source_model(flooring_fn=flooring_fn)
        if self._folded_output_normalization(floor):
            if self._implied_filter() is not None:
                self._update_spatial_model_implied(flooring_fn)
            else:
                self._update_spatial_model_folded(flooring_fn)
            return
        self.update_spatial_model(flooring_fn=flooring_fn)
        if self.normalization:
            self.normalize(flooring_fn=flooring_fn)

    # -- ISS2 / IPA with the power normalisation folded into the update matrix (round 5) -------
    def _folded_output_normalization(self, floor) -> bool:
        cls = type(self)
        algos = _ISS2 + _IPA
        # ISS1 on the per-bin statistics with the folded normalisation against the register-resident
        # sweep + weight pass + scale pass: faster up to 4 sources (the tuned covariance pass), see
        # DESIGN 4 item 40 (_routes "iss1_statistics" forces either for the tests).
        iss1 = _routes.get("iss1_statistics")
        # (a handful of mixtures: the one fused sweep launch wins, 112 against 125 us for one
        #  mixture of configs[1]; 32 mixtures 1.80 -> 1.29 ms, 128: 6.82 -> 4.65 ms)
        if (iss1 is not False and self._base_model[0] == _lib.SOURCE_GAUSS
                and (iss1 is True or (self._X.shape[1] <= 4
                                      and self._X.shape[0] * self._X.shape[2] >= 4096))):
            algos = algos + _ISS1
        return (self.spatial_algorithm in algos and bool(self.normalization)
                and self._power_normalization_or_off() and not self.partitioning
                and not self._uses_filter() and host_floor(floor) is None and self._is_stock()
                and cls.update_spatial_model_iss1 is _MMILRMA.update_spatial_model_iss1
                and cls.update_spatial_model_iss2 is _MMILRMA.update_spatial_model_iss2
                and cls.update_spatial_model_ipa is _MMILRMA.update_spatial_model_ipa
                and _routes.get("folded_norm"))

    def _output_covariance(self, Y):
        """C_i = (1/T) sum_j y y^H of the separated spectrogram, (B, F, N, N): formed once and moved
        along by the folded updates (C <- G C G^H); rebuilt when anything else rewrote Y."""
        cache = getattr(self, "_ycov", None)
        if cache is not None and cache[1] == self._state_rev("output"):
            return cache[0]
        B, N, F, T = Y.shape
        return _ops.weighted_covariance(Y).reshape(B, F, N, N)

    def _update_spatial_model_folded(self, flooring_fn) -> None:
        """update_spatial_model() + normalize() of the ISS2 / IPA iterations in three passes over Y.
        The reference updates y <- G y, measures psi_n^2 = mean |y_n|^2 in a second pass and divides
        in a third (ilrma.py:1698-1908, :412-444).  The power of the UPDATED spectrogram is
        mean_i g_n^H C_i g_n with the covariance C_i of the current one, so psi is known before the
        update runs: rows of G / psi_n, basis / psi_n^p, one y <- G y, and C <- G C G^H for the next
        iteration.  The tracked log-determinant moves by sum_i log|det G_i|."""
        floor = self._resolve_floor(flooring_fn)
        if self.spatial_algorithm in _IPA:
            require_device_floor(floor, "IPA")
        Y = self._state_dev("output")
        B, N, F, T = Y.shape
        Vc = self._output_statistics(Y, flooring_fn)
        if Vc is None:
            if self._ggd_host_floor(flooring_fn):
                varphi = self._ggd_weights_host_floor(Y, floor)
            else:
                varphi = _ops.ilrma_iss_weight(*self._nmf_pair(), float(self.domain), Y=Y,
                                               model=self._model, flooring=floor)
            Vc = _ops.weighted_covariance(Y, varphi, _lib.WEIGHT_BIN_FRAME, N)
        if self.spatial_algorithm in _ISS1:
            G = _ops.iss1_transform(Vc, floor)
        elif self.spatial_algorithm in _ISS2:
            G = _ops.iss2_transform(Vc, resolve_pairs(getattr(self, "pair_selector", None), N),
                                    floor, self._info_tensor())
        else:
            G = _ops.ipa_sweep(Vc, self.lqpqm_normalization, self.newton_iter, floor,
                               self._info_tensor(), newton_ws=self._newton_words(Y.device),
                               not_converged=self._newton_counter())
        C = self._output_covariance(Y)
        _ops.ilrma_normalize_filter(G, C, self._state_dev("basis"), float(self.domain), floor,
                                    self._ws, self._ws_bytes)
        spare = getattr(self, "_ycov_spare", None)
        if spare is None or spare.shape != C.shape or spare.data_ptr() == C.data_ptr():
            spare = dv.empty(tuple(C.shape), dv.c128, Y.device)
        _ops.covariance_congruence(C, G, spare)
        tracked = self._tracked_logdet()
        if tracked is not None:
            tracked.add_(_ops.sum_logdet(G))
        _ops.separate(Y, G, out=Y)
        self._state_touch("output")
        self._state_touch("basis")
        self._ycov, self._ycov_spare = (spare, self._state_rev("output")), C
        self._restamp_logdet(tracked)

    # -- ISS / ISS2 / IPA read through the filters their updates imply (round 5) ---------------------
    def _implied_filter(self):
        """W with output = W x while nothing else rewrote ``output`` since, else None."""
        kept = getattr(self, "_implied", None)
        if (kept is None or kept[1] != self._state_rev("output")
                or not _routes.get("implied_filter")):
            return None
        return kept[0]

    def _fill_output_from_implied_filter(self) -> None:
        W = self._implied[0]
        _ops.separate(self._X, W, out=self._state()["output"]["dev"])

    def _leave_implied_route(self) -> None:
        """Form Y = W x now and go on with the iterations that rewrite it (the reference's)."""
        W = self._implied[0]
        self._state_dev("output")  # (runs the deferred fill)
        if getattr(self, "_logdet_cache", None) is not None:  # (the on-Y updates move it along)
            self._logdet_cache = (_ops.sum_logdet(W), self._state_rev("output"))
        self._implied = None

    def _update_spatial_model_implied(self, flooring_fn) -> None:
        """update_spatial_model() + normalize() of the ISS / ISS2 / IPA iterations without touching Y.
        The reference keeps only the separated spectrogram and rewrites it, y <- G y
        (ilrma.py:1635-1908), which makes an iteration four passes over (N, F, T): basis, activation,
        statistics, rewrite.  With output = W x the statistics are mean phi y y^H = W U W^H with the
        weighted covariances U of the MIXTURE (the IP iterations' pass), the NMF passes read |W x|^2
        like they do for IP1, the update is W <- G W on (F, N, N), and the power normalisation is the
        filter form's g^H C_x g: three passes, Y formed when somebody reads ``output``
        (_state_defer).  Same arithmetic up to the order of the N-term sums."""
        floor = self._resolve_floor(flooring_fn)
        if self.spatial_algorithm in _IPA:
            require_device_floor(floor, "IPA")
        if self._amp_exceeded():
            self._leave_implied_route()
            return self._update_spatial_model_folded(flooring_fn)
        W = self._implied_filter()
        B, N, F, T = self._X.shape
        dev = self._X.device
        if self._U is None:
            self._U = dv.empty((B, F, N, N, N), dv.c128, dev)
        if getattr(self, "_Vc", None) is None or tuple(self._Vc.shape) != (B, F, N, N, N):
            self._Vc = dv.empty((B, F, N, N, N), dv.c128, dev)
        _ops.ilrma_weighted_covariance(self._X, *self._nmf_pair(), float(self.domain), self._ws,
                                       self._ws_bytes, out=self._U, W=W, model=self._model,
                                       flooring=floor)
        tracked = self._amp_tracked(self._C())
        Vc = _ops.covariance_congruence(self._U, W, self._Vc, tracked=tracked)
        self._amp_launched(tracked)
        if self.spatial_algorithm in _ISS1:
            G = _ops.iss1_transform(Vc, floor)
        elif self.spatial_algorithm in _ISS2:
            G = _ops.iss2_transform(Vc, resolve_pairs(getattr(self, "pair_selector", None), N),
                                    floor, self._info_tensor())
        else:
            G = _ops.ipa_sweep(Vc, self.lqpqm_normalization, self.newton_iter, floor,
                               self._info_tensor(), newton_ws=self._newton_words(dev),
                               not_converged=self._newton_counter())
        spare = getattr(self, "_implied_spare", None)
        if spare is None or spare.shape != W.shape or spare.data_ptr() == W.data_ptr():
            spare = dv.empty(tuple(W.shape), dv.c128, dev)
        _ops.compose_filters(G, W, spare)
        _ops.ilrma_normalize_filter(spare, self._C(), self._state_dev("basis"), float(self.domain),
                                    floor, self._ws, self._ws_bytes)
        self._state_touch("basis")
        self._state_defer("output", self._fill_output_from_implied_filter)
        self._implied, self._implied_spare = (spare, self._state_rev("output")), W

    def _power_normalization_or_off(self) -> bool:
        return (not self.normalization) or type(self.normalization) is bool \
            or self.normalization == "power"

    def update_source_model(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:924-978."""
        if self.source_algorithm == "MM":
            self.update_source_model_mm(flooring_fn=flooring_fn)
        elif self.source_algorithm == "ME":
            self.update_source_model_me(flooring_fn=flooring_fn)
        else:
            raise ValueError(
                "{}-algorithm-based source model updates are not supported.".format(
                    self.source_algorithm
                )
            )

    def update_source_model_me(self, flooring_fn="self") -> None:
        """Same sums as MM with exponent 1 (ref: ssspy/bss/ilrma.py:980-1005, :1249-1401)."""
        if self.domain != 2:
            raise ValueError("Domain parameter is expected 2, but given {}.".format(self.domain))
        if self.partitioning:
            self.update_latent_me()
        self.update_basis_me(flooring_fn=flooring_fn)
        self.update_activation_me(flooring_fn=flooring_fn)

    def update_basis_me(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:1249-1325."""
        if self.source_algorithm != "ME":
            raise ValueError("update_basis_me needs source_algorithm='ME'.")
        self.update_basis_mm(flooring_fn=flooring_fn)

    def update_activation_me(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:1327-1401."""
        if self.source_algorithm != "ME":
            raise ValueError("update_activation_me needs source_algorithm='ME'.")
        self.update_activation_mm(flooring_fn=flooring_fn)

    def update_source_model_mm(self, flooring_fn="self") -> None:
        if self.partitioning:
            self.update_latent_mm()
        self.update_basis_mm(flooring_fn=flooring_fn)
        self.update_activation_mm(flooring_fn=flooring_fn)

    def _partition_update(self, steps, flooring_fn="self") -> None:
        # (with a flooring callable the kernels cannot run the step runs unfloored -- the resolved
        # floor is then (NONE, 0) -- and the callers floor the small array on the host afterwards)
        if self._base_model[0] != _lib.SOURCE_GAUSS:
            require_device_floor(self._resolve_floor(flooring_fn), "Partitioning with a heavy-tailed model")
        src, W = self._source_and_filter()
        _ops.ilrma_partition_update(src, W, self._state_dev("basis"), self._state_dev("activation"),
                                    self._state_dev("latent"), self._Teff, self._Vrep,
                                    float(self.domain), steps, self._resolve_floor(flooring_fn),
                                    self._ws, self._ws_bytes, model=self._model)

    def update_latent_mm(self) -> None:
        """ref: ssspy/bss/ilrma.py:1007-1049 (and :2384-2432, :3698-3743)."""
        self._partition_update(_lib.PARTITION_LATENT)
        self._state_touch("latent")

    def update_latent_me(self) -> None:
        """ref: ssspy/bss/ilrma.py:1206-1247 (and :2610-2657)."""
        if self.source_algorithm != "ME":
            raise ValueError("update_latent_me needs source_algorithm='ME'.")
        if self.domain != 2:
            raise ValueError("Domain parameter is expected 2, but given {}.".format(self.domain))
        self.update_latent_mm()

    def _source_and_filter(self):
        """(spectrogram tensor, filter tensor or None) whose |W x|^2 the MM updates use."""
        if self._uses_filter():
            return self._X, self._state_dev("demix_filter")
        W = self._implied_filter()
        if W is not None:
            return self._X, W
        return self._state_dev("output"), None

    def update_basis_mm(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:1051-1128."""
        if self.partitioning:
            self._partition_update(_lib.PARTITION_BASIS, flooring_fn)
            self._state_touch("basis")
            self._host_floor_state("basis", self._resolve_floor(flooring_fn))
            return
        src, W = self._source_and_filter()
        floor = self._resolve_floor(flooring_fn)
        _ops.ilrma_update_basis(src, W, self._state_dev("basis"), self._state_dev("activation"),
                                float(self.domain), floor, self._ws, self._ws_bytes,
                                model=self._model)
        self._state_touch("basis")
        self._host_floor_state("basis", floor)  # (an arbitrary callable: T = flooring_fn(T) on the host)

    def update_activation_mm(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:1130-1204."""
        if self.partitioning:
            self._partition_update(_lib.PARTITION_ACTIVATION, flooring_fn)
            self._state_touch("activation")
            self._host_floor_state("activation", self._resolve_floor(flooring_fn))
            return
        src, W = self._source_and_filter()
        floor = self._resolve_floor(flooring_fn)
        _ops.ilrma_update_activation(src, W, self._state_dev("basis"),
                                     self._state_dev("activation"), float(self.domain), floor,
                                     self._ws, self._ws_bytes, model=self._model)
        self._state_touch("activation")
        self._host_floor_state("activation", floor)

    def update_spatial_model(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:1403-1438."""
        if self.spatial_algorithm in _IP1:
            self.update_spatial_model_ip1(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _ISS1:
            self.update_spatial_model_iss1(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _IP2:
            self.update_spatial_model_ip2(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _ISS2:
            self.update_spatial_model_iss2(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _IPA:
            self.update_spatial_model_ipa(flooring_fn=flooring_fn)
        else:
            raise NotImplementedError("Not support {}.".format(self.spatial_algorithm))

    def update_spatial_model_ipa(self, flooring_fn="self") -> None:
        """Iterative projection with adjustment on per-bin statistics: per source, weighted
        covariance of the current output, LQPQM update matrix, Y <- G Y.
        ref: ssspy/bss/ilrma.py:1794-1908."""
        require_device_floor(self._resolve_floor(flooring_fn), "IPA")
        Y = self._state_dev("output")
        Vc = self._output_statistics(Y, flooring_fn)
        varphi = None
        if Vc is None:
            varphi = _ops.ilrma_iss_weight(*self._nmf_pair(), float(self.domain), Y=Y,
                                           model=self._model,
                                           flooring=self._resolve_floor(flooring_fn))
        _ops.update_by_ipa(Y, varphi, _lib.WEIGHT_BIN_FRAME, self.lqpqm_normalization,
                           self.newton_iter, self._resolve_floor(flooring_fn), self._info_tensor(),
                           not_converged=self._newton_counter(), Vc=Vc)
        self._state_touch("output")

    def _output_statistics(self, Y, flooring_fn):
        """U_n = mean_j varphi_nij y y^H of the separated spectrogram for the ISS2 / IPA steps in ONE
        pass: the covariance pass of the IP updates (weights formed from the NMF tiles on the fly)
        serves with Y in place of X -- instead of a weight pass (read |y|^2, write (N, F, T) weights)
        plus the generic weighted covariance (round 5: 0.78 -> 0.3 ms at 32 mixtures of
        configs[1]).  The Gauss weights 1 / (T V)^(2/p) do not depend on y and the pass takes no
        filter; the t / GGD weights are functions of |y|^2, which the pass forms as |w_n^H x|^2:
        up to 4 sources it is handed Y with identity filters (round 6), above that it forms the
        weights from the spectrogram it is given.  None: the caller forms weights (a host floor on
        the GGD weights; the t model's weights hold no floor)."""
        floor = self._resolve_floor(flooring_fn)
        gauss = self._base_model[0] == _lib.SOURCE_GAUSS
        if self._base_model[0] == _lib.SOURCE_GGD and host_floor(floor) is not None:
            return None  # (the callers refuse: GGD's weights floor |y|^(2 - beta) per element)
        B, N, F, T = Y.shape
        W = None
        if not gauss and N <= 4:
            W = getattr(self, "_eye_filter", None)
            if W is None or tuple(W.shape) != (B, F, N, N) or W.device != Y.device:
                W = dv.eye_filters(B, F, N, Y.device)
                self._eye_filter = W
        if getattr(self, "_Vc", None) is None or tuple(self._Vc.shape) != (B, F, N, N, N):
            self._Vc = dv.empty((B, F, N, N, N), dv.c128, Y.device)
        _ops.ilrma_weighted_covariance(Y, *self._nmf_pair(), float(self.domain), self._ws,
                                       self._ws_bytes, out=self._Vc, W=W, model=self._model,
                                       flooring=floor)
        return self._Vc

    def _ggd_weights_host_floor(self, Y, floor):
        """varphi (B, N, F, T) of the GGD model for a flooring callable the kernels do not know
        (round 6).  The reference floors q = |y|^(2 - beta) per element before it forms
        1 / ((2 / beta) q (T V)^(beta / p)) (ssspy/bss/ilrma.py:3993-4011, :4123-4141): q and the
        callable are evaluated on the host, one mixture -- the reference's (N, F, T) array -- at a
        time, and the weight kernel takes floor(q)^(2 / (2 - beta)) in place of |y|^2 with its own
        floor off: one spectrogram down and one real array up per update, the price of an opaque
        Python callable."""
        beta = float(self.beta)
        Yh = dv.to_host(Y)
        q = np.abs(Yh) ** (2.0 - beta)
        fq = np.stack([np.asarray(floor.host(qb), dtype=np.float64) for qb in q])
        if fq.shape != q.shape:
            raise ValueError("flooring_fn must return an array of the shape it was given")
        ypow = dv.to_device(fq ** (2.0 / (2.0 - beta)), dtype=np.float64, dev=Y.device)
        return _ops.ilrma_iss_weight(*self._nmf_pair(), float(self.domain), model=self._model,
                                     flooring=(_lib.FLOOR_NONE, 0.0), Ypow=ypow)

    def _ggd_host_floor(self, flooring_fn) -> bool:
        return (self._base_model[0] == _lib.SOURCE_GGD
                and host_floor(self._resolve_floor(flooring_fn)) is not None)

    def update_spatial_model_ip2(self, flooring_fn="self") -> None:
        """Weighted covariance + pairwise iterative projection.  ref: ssspy/bss/ilrma.py:1509-1633."""
        # (the t model's weights hold no floor, ssspy/bss/ilrma.py:2915-2935; GGD's floor
        #  |y|^(2 - beta) per element, :3987-4011: a callable goes through _ggd_weights_host_floor)
        B, N, F, T = self._X.shape
        if self._ggd_host_floor(flooring_fn):
            Y = _ops.separate(self._X, self._state_dev("demix_filter"))
            varphi = self._ggd_weights_host_floor(Y, self._resolve_floor(flooring_fn))
            self._U = _ops.weighted_covariance(self._X, varphi, _lib.WEIGHT_BIN_FRAME, N)
        else:
            if self._U is None:
                self._U = dv.empty((B, F, N, N, N), dv.c128, self._X.device)
            _ops.ilrma_weighted_covariance(self._X, *self._nmf_pair(), float(self.domain),
                                           self._ws, self._ws_bytes, out=self._U,
                                           W=self._state_dev("demix_filter"), model=self._model,
                                           flooring=self._resolve_floor(flooring_fn))
        _ops.update_by_ip2(self._state_dev("demix_filter"), self._U,
                           resolve_pairs(getattr(self, "pair_selector", None), N),
                           self._resolve_floor(flooring_fn), self._info_tensor())
        self._state_touch("demix_filter")

    def update_spatial_model_iss2(self, flooring_fn="self") -> None:
        """Pairwise iterative source steering on per-bin statistics.  ref: ilrma.py:1698-1792."""
        Y = self._state_dev("output")
        N = Y.shape[1]
        Vc = self._output_statistics(Y, flooring_fn)
        if Vc is None:
            if self._ggd_host_floor(flooring_fn):
                varphi = self._ggd_weights_host_floor(Y, self._resolve_floor(flooring_fn))
            else:
                varphi = _ops.ilrma_iss_weight(*self._nmf_pair(), float(self.domain), Y=Y,
                                               model=self._model,
                                               flooring=self._resolve_floor(flooring_fn))
            Vc = _ops.weighted_covariance(Y, varphi, _lib.WEIGHT_BIN_FRAME, N)
        G = _ops.iss2_transform(Vc, resolve_pairs(getattr(self, "pair_selector", None), N),
                                self._resolve_floor(flooring_fn), self._info_tensor())
        _ops.separate(Y, G, out=Y)
        self._state_touch("output")

    def update_spatial_model_ip1(self, flooring_fn="self") -> None:
        """Weighted covariance + iterative projection.  ref: ssspy/bss/ilrma.py:1440-1507."""
        B, N, F, T = self._X.shape
        if self._ggd_host_floor(flooring_fn):  # (its weights floor |y|^(2 - beta) per element)
            Y = _ops.separate(self._X, self._state_dev("demix_filter"))
            varphi = self._ggd_weights_host_floor(Y, self._resolve_floor(flooring_fn))
            self._U = _ops.weighted_covariance(self._X, varphi, _lib.WEIGHT_BIN_FRAME, N)
        else:
            if self._U is None:
                self._U = dv.empty((B, F, N, N, N), dv.c128, self._X.device)
            _ops.ilrma_weighted_covariance(self._X, *self._nmf_pair(), float(self.domain),
                                           self._ws, self._ws_bytes, out=self._U,
                                           W=self._state_dev("demix_filter"), model=self._model,
                                           flooring=self._resolve_floor(flooring_fn))
        _ops.update_by_ip1(self._state_dev("demix_filter"), self._U,
                           self._resolve_floor(flooring_fn), self._info_tensor())
        self._state_touch("demix_filter")

    def update_spatial_model_iss1(self, flooring_fn="self") -> None:
        """Iterative source steering on per-bin statistics.  ref: ssspy/bss/ilrma.py:1635-1696."""
        Y = self._state_dev("output")
        N = Y.shape[1]
        floor = self._resolve_floor(flooring_fn)
        if self._ggd_host_floor(flooring_fn):
            varphi = self._ggd_weights_host_floor(Y, floor)
        else:
            varphi = _ops.ilrma_iss_weight(*self._nmf_pair(), float(self.domain), Y=Y,
                                           model=self._model, flooring=floor)
        frame_power = None
        if host_floor(floor) is not None:
            tracked = None
            _ops.update_by_iss1_host_floor(Y, varphi, _lib.WEIGHT_BIN_FRAME, floor.host)
        elif Y.shape[-1] <= _ops.iss1_fused_max_frames(N):
            # the sweep also leaves sum_i |y_nij|^2 of the new Y: the power normalisation that
            # follows in update_once() reads it instead of making its own pass over Y
            frame_power = dv.empty((Y.shape[0], N, Y.shape[-1]), dv.f64, Y.device)
            tracked = self._tracked_logdet()
            _ops.iss1_fused(Y, varphi, _lib.WEIGHT_BIN_FRAME, floor, r2_next=frame_power,
                            logdet=tracked)
        else:
            tracked = None
            Vc = _ops.weighted_covariance(Y, varphi, _lib.WEIGHT_BIN_FRAME, N)
            G = _ops.iss1_transform(Vc, floor)
            _ops.separate(Y, G, out=Y)
        self._state_touch("output")
        self._iss_frame_power = (frame_power, self._state_rev("output"))
        self._restamp_logdet(tracked)

    def normalize(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:333-363."""
        normalization = self.normalization
        assert normalization, "Set normalization."
        if type(normalization) is bool:
            normalization = "power"
        if normalization == "power":
            self.normalize_by_power(flooring_fn=flooring_fn)
        elif normalization == "projection_back":
            self.normalize_by_projection_back()
        else:
            raise NotImplementedError("Normalization {} is not implemented.".format(normalization))

    def normalize_by_projection_back(self) -> None:
        """Projection back as the per-iteration normalisation; the basis takes |scale|^p.
        ref: ssspy/bss/ilrma.py:446-522."""
        if self.partitioning:
            raise NotImplementedError(
                "Projection-back-based normalization is not applicable with partitioning function."
            )
        reference_id = 0 if self.reference_id is None else self.reference_id
        info = self._info_tensor()
        if self._uses_filter():
            W = self._state_dev("demix_filter")
            G = dv.empty(tuple(W.shape), dv.c128, W.device)
            _ops.projection_back_filter(W, reference_id, info, scale_out=G)
            self._state_touch("demix_filter")
        else:
            Y = self._state_dev("output")
            G = _ops.projection_back_scale(_ops.cross_covariance(self._X, Y),
                                           _ops.cross_covariance(Y, Y), reference_id, info)
            _ops.separate(Y, G, out=Y)
            self._state_touch("output")
        _ops.ilrma_scale_basis(self._state_dev("basis"), G, float(self.domain))
        self._state_touch("basis")

    def normalize_by_power(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/ilrma.py:365-444 (no partitioning)."""
        floor = self._resolve_floor(flooring_fn)
        if host_floor(floor) is not None:
            return self._normalize_by_power_host_floor(floor)
        if self.partitioning:
            filt = self._uses_filter()
            _ops.ilrma_partition_normalize(
                self._state_dev("demix_filter") if filt else None, self._C() if filt else None,
                None if filt else self._state_dev("output"), self._state_dev("basis"),
                self._state_dev("latent"), float(self.domain), floor, self._ws, self._ws_bytes)
            self._state_touch("demix_filter" if filt else "output")
            self._state_touch("basis")
            self._state_touch("latent")
            return
        if self._uses_filter():
            _ops.ilrma_normalize_filter(self._state_dev("demix_filter"), self._C(),
                                        self._state_dev("basis"), float(self.domain), floor,
                                        self._ws, self._ws_bytes)
            self._state_touch("demix_filter")
        else:
            Y = self._state_dev("output")
            frame_power, rev = getattr(self, "_iss_frame_power", (None, None))
            if rev != self._state_rev("output"):
                frame_power = None  # Y was replaced or rewritten since the sweep
            tracked = self._tracked_logdet()
            _ops.ilrma_normalize_output(Y, self._state_dev("basis"), float(self.domain), floor,
                                        self._ws, self._ws_bytes, frame_power=frame_power,
                                        logdet=tracked)
            self._state_touch("output")
            self._restamp_logdet(tracked)
        self._state_touch("basis")

    def _normalize_by_power_host_floor(self, floor) -> None:
        """psi = flooring_fn(sqrt(mean |y|^2)) with an arbitrary callable: the frame powers come from
        one device pass, the N scales are floored on the host, W (or Y) and the basis are rescaled.
        ref: ssspy/bss/ilrma.py:412-444."""
        B, N, F, T = self._X.shape
        p = float(self.domain)
        filt = self._uses_filter()
        src = self._X if filt else self._state_dev("output")
        r2 = dv.to_host(_ops.iva_frame_power(src, self._state_dev("demix_filter") if filt else None))
        psi = np.stack([np.asarray(floor.host(np.sqrt(r.sum(axis=-1) / (F * T))), dtype=np.float64)
                        for r in r2])  # (B, N)
        lead = (slice(None),) if self._batched else (0,)
        psi_v = psi[lead]
        if self.partitioning:  # ref: ssspy/bss/ilrma.py:422-430 (small arrays: (N, K), (F, K))
            Z_psi = np.asarray(self.latent) / (psi_v[..., :, None] ** p)
            scale = np.sum(Z_psi, axis=-2)
            self.basis = np.asarray(self.basis) * scale[..., None, :]
            self.latent = Z_psi / scale[..., None, :]
        else:
            self.basis = np.asarray(self.basis) / (psi_v[..., :, None, None] ** p)
        if filt:
            self.demix_filter = np.asarray(self.demix_filter) / psi_v[..., None, :, None]
        else:
            G = np.zeros((B, F, N, N), dtype=np.complex128)
            G[:, :, np.arange(N), np.arange(N)] = 1.0 / psi[:, None, :]
            Y = self._state_dev("output")
            _ops.separate(Y, dv.to_device(G, dev=Y.device), out=Y)
            self._state_touch("output")
            self._restamp_logdet(None)

    def compute_loss(self) -> float:
        """Negative log-likelihood (ref: ssspy/bss/ilrma.py:1910-1967)."""
        return self._host_loss(*self._loss_terms())

    def _loss_terms(self, data_out=None, logdet_out=None):
        """(data term, sum_i log|det W_i|) of the current state on the device, each (B,)."""
        T, V = self._nmf_pair()
        if self._uses_filter():
            W = self._state_dev("demix_filter")
            data = _ops.ilrma_loss_data(self._X, W, T, V, float(self.domain), out=data_out,
                                        model=self._model)
        elif self._implied_filter() is not None:
            W = self._implied_filter()
            data = _ops.ilrma_loss_data(self._X, W, T, V, float(self.domain), out=data_out,
                                        model=self._model)
        else:
            Y = self._state_dev("output")
            data = _ops.ilrma_loss_data(Y, None, T, V, float(self.domain), out=data_out,
                                        model=self._model)
            tracked = self._tracked_logdet()
            if tracked is not None:  # moved along by the sweeps and the normalisation
                if logdet_out is None:
                    return data, tracked
                logdet_out.copy_(tracked)  # (the next sweep moves the tracked sum in place)
                return data, logdet_out
            W = _ops.demix_from_covariance(_ops.cross_covariance(Y, self._X), self._C(),
                                           self._info_tensor())
        return data, _ops.sum_logdet(W, out=logdet_out)


class GaussILRMA(_MMILRMA):
    """Gauss-ILRMA (ref: ssspy/bss/ilrma.py:582-1989).

    Args mirror the reference: ``n_basis``, ``spatial_algorithm`` ("IP"/"IP1"/"ISS"/"ISS1"
    on device), ``source_algorithm`` ("MM"), ``domain`` in (0, 2], ``partitioning`` (False),
    ``flooring_fn``, ``pair_selector``, ``callbacks``, ``normalization`` (True / "power" /
    False), ``scale_restoration`` (True / "projection_back" / False), ``record_loss``,
    ``reference_id``, ``rng``.
    """

    _ipa_default_kwargs = {"lqpqm_normalization": True, "newton_iter": 1}
    _default_kwargs = _ipa_default_kwargs

    def __init__(
        self,
        n_basis: int,
        spatial_algorithm: str = "IP",
        source_algorithm: str = "MM",
        domain: float = 2,
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        normalization: Optional[Union[bool, str]] = True,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
        **kwargs,
    ) -> None:
        super().__init__(
            n_basis=n_basis,
            partitioning=partitioning,
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
            rng=rng,
        )
        assert spatial_algorithm in spatial_algorithms, "Not support {}.".format(spatial_algorithm)
        assert source_algorithm in source_algorithms, "Not support {}.".format(source_algorithm)
        assert 0 < domain <= 2, "domain parameter should be chosen from [0, 2]."
        if source_algorithm == "ME":
            assert domain == 2, "domain parameter should be 2 when you specify ME algorithm."
        valid_keys = set(self._ipa_default_kwargs) if spatial_algorithm == "IPA" else set()
        invalid_keys = set(kwargs) - valid_keys
        assert invalid_keys == set(), "Invalid keywords {} are given.".format(invalid_keys)
        for key, value in kwargs.items():
            setattr(self, key, value)
        for key in valid_keys:
            if not hasattr(self, key):
                setattr(self, key, self._default_kwargs[key])
        self._configure(spatial_algorithm, source_algorithm, domain, partitioning, normalization,
                        pair_selector)


class TILRMA(_MMILRMA):
    """ILRMA on the Student-t distribution (ref: ssspy/bss/ilrma.py:1992-3334).

    Args as the reference: ``n_basis``, ``dof`` (degree of freedom nu > 0), then the
    ``GaussILRMA`` arguments.  ``spatial_algorithm="IPA"`` raises ``ValueError`` as upstream.
    """

    def __init__(
        self,
        n_basis: int,
        dof: float,
        spatial_algorithm: str = "IP",
        source_algorithm: str = "MM",
        domain: float = 2,
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        normalization: Optional[Union[bool, str]] = True,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
    ) -> None:
        super().__init__(
            n_basis=n_basis,
            partitioning=partitioning,
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
            rng=rng,
        )
        assert spatial_algorithm in spatial_algorithms, "Not support {}.".format(spatial_algorithms)
        assert source_algorithm in source_algorithms, "Not support {}.".format(source_algorithm)
        assert 0 < domain <= 2, "domain parameter should be chosen from [0, 2]."
        if spatial_algorithm == "IPA":
            raise ValueError("IPA is not supported for t-ILRMA.")
        if source_algorithm == "ME":
            assert domain == 2, "domain parameter should be 2 when you specify ME algorithm."
        assert dof > 0, "dof should be positive."
        self.dof = dof
        self._configure(spatial_algorithm, source_algorithm, domain, partitioning, normalization,
                        pair_selector)

    @property
    def _base_model(self):
        return (_lib.SOURCE_T, float(self.dof))

    def _repr_model(self) -> str:
        return ", dof={}".format(self.dof)


class GGDILRMA(_MMILRMA):
    """ILRMA on the generalised Gaussian distribution (ref: ssspy/bss/ilrma.py:3337-4410).

    Args as the reference: ``n_basis``, ``beta`` (shape parameter in (0, 2)), then the
    ``GaussILRMA`` arguments (``source_algorithm`` must be "MM").
    """

    def __init__(
        self,
        n_basis: int,
        beta: float,
        spatial_algorithm: str = "IP",
        source_algorithm: str = "MM",
        domain: float = 2,
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        normalization: Optional[Union[bool, str]] = True,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
    ) -> None:
        super().__init__(
            n_basis=n_basis,
            partitioning=partitioning,
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
            rng=rng,
        )
        assert 0 < beta < 2, "Shape parameter {} shoule be chosen from (0, 2).".format(beta)
        assert spatial_algorithm in spatial_algorithms, "Not support {}.".format(spatial_algorithms)
        assert source_algorithm == "MM", "Not support {}.".format(source_algorithm)
        assert 0 < domain <= 2, "domain parameter should be chosen from [0, 2]."
        if spatial_algorithm == "IPA":
            raise ValueError("IPA is not supported for GGD-ILRMA.")
        self.beta = beta
        self._configure(spatial_algorithm, source_algorithm, domain, partitioning, normalization,
                        pair_selector)

    @property
    def _base_model(self):
        return (_lib.SOURCE_GGD, float(self.beta))

    def _repr_model(self) -> str:
        return ", beta={}".format(self.beta)
