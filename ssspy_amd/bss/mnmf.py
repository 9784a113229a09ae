"""Multichannel NMF (FastGaussMNMF, GaussMNMF) on MI355X.

Drop-in separator for the reference's ``ssspy.bss.mnmf.FastGaussMNMF``
(ssspy/bss/mnmf.py:1076-1675 on top of FastMNMFBase :417-678 and MNMFBase :21-297): jointly
diagonalisable full-rank spatial model with per-bin diagonaliser ``Q`` (n_bins, n_channels,
n_channels), diagonal spatial ``D`` (n_bins, n_sources, n_channels), NMF ``basis`` /
``activation``; ``update_once`` = basis, activation, diagonaliser (IP1), spatial, power
normalisation; output by the multichannel Wiener filter.  ``diagonalizer_algorithm`` may be "IP" /
"IP1" or "IP2" (pairwise).  ``GaussMNMF`` (ssspy/bss/mnmf.py:681-1073) is the full-rank model:
``spatial`` (n_sources, n_bins, M, M) Hermitian PSD, updated by a matrix geometric mean.
``GaussMNMF(partitioning=True)`` keeps a shared basis / activation and latent variables.

``instant_covariance`` (the (n_bins, n_frames, M, M) PSD-projected outer products the reference
materialises at reset, mnmf.py:167-188) is never read by FastGaussMNMF's updates and is not
computed here.
"""

import functools
from typing import Callable, Iterable, List, Optional, Tuple, Union

import numpy as np

from .. import _device as dv
from .. import _lib, _ops
from ..special.flooring import identity, max_flooring
from ..utils.flooring import choose_flooring_fn, device_flooring, host_floor, require_device_floor
from ..utils.select_pair import resolve_pairs, sequential_pair_selector
from ._device_state import DeviceStateMixin, Synced
from .base import IterativeMethodBase

__all__ = ["FastGaussMNMF", "GaussMNMF"]

diagonalizer_algorithms = ["IP", "IP1", "IP2"]
EPS = 1e-10


class MNMFBase(DeviceStateMixin, IterativeMethodBase):
    """ref: ssspy/bss/mnmf.py:21-297."""

    output = Synced(dv.c128)
    basis = Synced(dv.f64)
    activation = Synced(dv.f64)
    latent = Synced(dv.f64)

    def __init__(
        self,
        n_basis: int,
        n_sources: Optional[int] = None,
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        callbacks=None,
        normalization: bool = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
    ) -> None:
        super().__init__(callbacks=callbacks, record_loss=record_loss)
        self.n_basis = n_basis
        self.n_sources = n_sources
        self.partitioning = partitioning
        self.flooring_fn = identity if flooring_fn is None else flooring_fn
        self.normalization = normalization
        self.input = None
        self.reference_id = reference_id
        self.rng = np.random.default_rng() if rng is None else rng

    def __call__(
        self, input: np.ndarray, n_iter: int = 100, initial_call: bool = True, **kwargs
    ) -> np.ndarray:
        """Separate a frequency-domain multichannel mixture (ref: ssspy/bss/mnmf.py:90-118)."""
        self._bind_input(input)
        # the reference fills `output` at reset (ssspy/bss/mnmf.py:540); only a callback can read it
        # before the Wiener filter of the final state below replaces it
        # (or a subclass's own update_once / compute_loss: then the reset-time filter is kept)
        cls = type(self)
        bases = [base for base in (FastGaussMNMF, GaussMNMF) if isinstance(self, base)]
        stock = bool(bases) and all(getattr(cls, name) is getattr(base, name)
                                    for base in bases for name in ("update_once", "compute_loss"))
        self._skip_reset_output = not self.callbacks and stock
        if self._skip_reset_output:
            self._state().pop("output", None)  # no stale estimate of an earlier call to be read
        try:
            self._reset(**kwargs)
        finally:
            self._skip_reset_output = False
        resident = getattr(self, "_iterate_with_resident_loss", None)
        if resident is None or not resident(int(n_iter), initial_call):
            IterativeMethodBase.__call__(self, n_iter=n_iter, initial_call=initial_call)
        self._separate_dev()
        return self._final_output()

    def _init_nmf(self, flooring_fn="self", rng=None) -> None:
        """ref: ssspy/bss/mnmf.py:190-259."""
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        if rng is None:
            rng = np.random.default_rng()
        N, F, T, K = self.n_sources, self.n_bins, self.n_frames, self.n_basis
        src = () if self.partitioning else (N,)
        if not self._state_has("basis"):
            self.basis = flooring_fn(rng.random(self._lead() + src + (F, K)))
        else:
            self.basis = np.array(self.basis, dtype=np.float64, copy=True)
        if not self._state_has("activation"):
            self.activation = flooring_fn(rng.random(self._lead() + src + (K, T)))
        else:
            self.activation = np.array(self.activation, dtype=np.float64, copy=True)
        if self.partitioning:
            if not self._state_has("latent"):
                Z = rng.random(self._lead() + (N, K))
                self.latent = flooring_fn(Z / Z.sum(axis=-2, keepdims=True))
            else:
                self.latent = np.array(self.latent, dtype=np.float64, copy=True)
            B = self._X.shape[0]
            self._Teff = dv.empty((B, N, F, K), dv.f64, self._X.device)
            self._Vrep = dv.empty((B, N, K, T), dv.f64, self._X.device)

    def reconstruct_nmf(self, basis, activation, latent=None) -> np.ndarray:
        """Lambda = T V (ref: ssspy/bss/mnmf.py:264-297); host-side convenience."""
        if latent is not None:
            return np.einsum("...nk,...ik,...kj->...nij", latent, basis, activation)
        return basis @ activation

    def _check_separate_input(self, input) -> None:
        """``separate(input)`` filters with the parameters of the bound call, whose activation has
        one column per frame and one set per mixture: another spectrogram must have that shape (the
        reference fails with a broadcasting error; the kernels would read out of bounds)."""
        want = tuple(self._X.shape) if input.ndim == 4 else tuple(self._X.shape[1:])
        if tuple(input.shape) != want or (input.ndim == 4) != self._batched:
            raise ValueError(
                "separate() expects a spectrogram of shape {} (the one the parameters were fitted "
                "on), got {}.".format(want if input.ndim == 4 or not self._batched
                                      else tuple(self._X.shape), tuple(input.shape)))

    def _nmf_pair(self):
        """Device (basis, activation) in the per-source layout the kernels take (the expansion
        (z_nk t_ik, v_kj) with partitioning)."""
        if not self.partitioning:
            return self._state_dev("basis"), self._state_dev("activation")
        _ops.ilrma_partition_expand(self._state_dev("basis"), self._state_dev("activation"),
                                    self._state_dev("latent"), self._Teff, self._Vrep)
        return self._Teff, self._Vrep


class FastMNMFBase(MNMFBase):
    """ref: ssspy/bss/mnmf.py:417-678."""

    diagonalizer = Synced(dv.c128)
    spatial = Synced(dv.f64)

    def _reset(self, flooring_fn="self", **kwargs) -> None:
        """ref: ssspy/bss/mnmf.py:499-540."""
        assert self._has_input(), "Specify data!"
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        for key, value in kwargs.items():
            setattr(self, key, value)
        B, M, F, T = self._X.shape
        N = M if self.n_sources is None else self.n_sources
        self.n_sources, self.n_channels = N, M
        self.n_bins, self.n_frames = F, T
        # (an arbitrary callable: the steps run one by one with the floor off, the callable on the
        #  small arrays on the host in between -- round 5, see _update_host_floor)
        self._floor = device_flooring(flooring_fn, allow_host=True, what="FastMNMF")
        self._init_nmf(flooring_fn=flooring_fn, rng=self.rng)
        self._init_diagonalizer(rng=self.rng)
        self._init_spatial(flooring_fn=flooring_fn, rng=self.rng)
        self._ws, self._ws_bytes = _ops.fastmnmf_workspace(B, N, M, F, T, self.n_basis,
                                                           self._X.device)
        # |Q x|^2 handed from the spatial pass to the next basis / activation passes (None for
        # shapes without it); valid while neither the diagonaliser nor the input moved under it
        self._handover = _ops.fastmnmf_handover(B, N, M, F, T, self.n_basis, self._X.device)
        self._handover_key = None
        if not getattr(self, "_skip_reset_output", False):
            self._separate_dev()

    def _init_diagonalizer(self, rng=None) -> None:
        """ref: ssspy/bss/mnmf.py:542-564."""
        M, F = self.n_channels, self.n_bins
        if not self._state_has("diagonalizer"):
            self.diagonalizer = np.tile(np.eye(M, dtype=np.complex128), self._lead() + (F, 1, 1))
        else:
            self.diagonalizer = np.array(self.diagonalizer, dtype=np.complex128, copy=True)

    def _init_spatial(self, flooring_fn="self", rng=None) -> None:
        """ref: ssspy/bss/mnmf.py:566-600."""
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        if rng is None:
            rng = np.random.default_rng()
        N, M, F = self.n_sources, self.n_channels, self.n_bins
        if not self._state_has("spatial"):
            self.spatial = flooring_fn(rng.random(self._lead() + (F, N, M)))
        else:
            # private device copy (the reference keeps the caller's array; it is never mutated here)
            self.spatial = np.array(self.spatial, dtype=np.float64, copy=True)

    def _resolve_floor(self, flooring_fn):
        if type(flooring_fn) is str and flooring_fn == "self":
            return self._floor
        return device_flooring(choose_flooring_fn(flooring_fn, method=self), allow_host=True,
                               what="FastMNMF")

    def _update_host_floor(self, steps, floor) -> None:
        """The steps of `steps` one by one for a flooring callable the kernels do not know: each pass
        over X runs with the floor off and the callable is applied on the host to the small array the
        reference applies it to -- basis (N, F, K) and activation (N, K, T) after their updates
        (mnmf.py:1358, 1415), the IP1 denominators (F,) per channel (_update_spatial_model.py:63-76),
        the normalisation scales psi (M,) (mnmf.py:673)."""
        for flag, name in ((_lib.MNMF_BASIS, "basis"), (_lib.MNMF_ACTIVATION, "activation")):
            if steps & flag:
                self._update(flag, identity)
                dev = self._state_dev(name)
                dev.copy_(dv.to_device(_ops._apply_host(floor.host, dev), dev=dev.device))
                self._state_touch(name)
        if steps & _lib.MNMF_DIAGONALIZER:
            U = self._diagonalizer_covariance()
            _ops.update_by_ip1(self._state_dev("diagonalizer"), U, floor, self._info_tensor())
            self._state_touch("diagonalizer")
        if steps & _lib.MNMF_SPATIAL:
            self._update(_lib.MNMF_SPATIAL, identity)
        if steps & _lib.MNMF_NORMALIZE:
            # psi_m^2 = mean_ij |q_im^H x_ij|^2 = mean_i q_im^H C_i q_im with the static covariance C:
            # (F, M, M) arrays on the host, psi floored by the callable, Q rows / psi, D / psi^2
            Q, D = self._state_dev("diagonalizer"), self._state_dev("spatial")
            Qh, Dh, Ch = dv.to_host(Q), dv.to_host(D), dv.to_host(self._C())
            for b in range(Qh.shape[0]):
                qx2 = np.real(np.einsum("fma,fab,fmb->fm", Qh[b], Ch[b], Qh[b].conj())).mean(axis=0)
                psi = np.asarray(floor.host(np.sqrt(np.maximum(qx2, 0.0))), dtype=np.float64)
                Qh[b] /= psi[None, :, None]
                Dh[b] /= psi ** 2
            Q.copy_(dv.to_device(Qh, dev=Q.device))
            D.copy_(dv.to_device(Dh, dev=D.device))
            self._state_touch("diagonalizer")
            self._state_touch("spatial")

    def _diagonalizer_covariance(self):
        """U_m = mean_j x x^H / R~_m (B, F, M, M, M): the weights of the diagonaliser updates."""
        if self.n_sources <= 4 and self.n_channels <= 4 and self.n_sources >= 2:
            return _ops.fastmnmf_diagonalizer_covariance(
                self._X, self._state_dev("spatial"), self._state_dev("basis"),
                self._state_dev("activation"), ws=self._ws, ws_bytes=self._ws_bytes)
        weights = _ops.fastmnmf_weights(
            self._X, self._state_dev("diagonalizer"), self._state_dev("spatial"),
            self._state_dev("basis"), self._state_dev("activation"))
        return _ops.weighted_covariance(self._X, weights, _lib.WEIGHT_BIN_FRAME, self.n_channels)

    def _update(self, steps, flooring_fn="self", logdet=None, logdet_stride=0) -> None:
        """``logdet`` (the record_loss loop): the call also leaves sum_i log|det Q_i| of the
        diagonalisers it starts from, as shares (``_ops.fastmnmf_update_logdet``)."""
        floor = self._resolve_floor(flooring_fn)
        if host_floor(floor) is not None:
            assert logdet is None
            return self._update_host_floor(steps, floor)
        need_c = bool(steps & _lib.MNMF_NORMALIZE)
        args = (
            self._X, self._C() if need_c else None, self._state_dev("diagonalizer"),
            self._state_dev("spatial"), self._state_dev("basis"), self._state_dev("activation"),
            steps, floor, self._ws, self._ws_bytes, self._info_tensor(),
        )
        handover = getattr(self, "_handover", None)
        key = (self._state_rev("diagonalizer"), self._X.data_ptr())
        if logdet is not None:
            valid = _ops.fastmnmf_update_logdet(*args, handover, self._handover_key == key, logdet,
                                                logdet_stride)
        elif handover is None:
            _ops.fastmnmf_update(*args)
            valid = False
        else:
            valid = _ops.fastmnmf_update_handover(*args, handover, self._handover_key == key)
        for name in ("diagonalizer", "spatial", "basis", "activation"):
            self._state_touch(name)
        self._handover_key = (
            (self._state_rev("diagonalizer"), self._X.data_ptr()) if valid else None)

    def normalize(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:602-630."""
        normalization = self.normalization
        assert normalization, "Set normalization."
        if type(normalization) is bool:
            normalization = "power"
        if normalization == "power":
            self.normalize_by_power(flooring_fn=flooring_fn)
        else:
            raise NotImplementedError("Normalization {} is not implemented.".format(normalization))

    def normalize_by_power(self, flooring_fn="self") -> None:
        """psi_m from mean |q_m^H x|^2; Q rows / psi, D / psi^2 (ref: ssspy/bss/mnmf.py:632-678)."""
        self._update(_lib.MNMF_NORMALIZE, flooring_fn)


class FastGaussMNMF(FastMNMFBase):
    """FastMNMF on a Gaussian distribution (ref: ssspy/bss/mnmf.py:1076-1675)."""

    def __init__(
        self,
        n_basis: int,
        n_sources: Optional[int] = None,
        diagonalizer_algorithm: str = "IP",
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        normalization: bool = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
    ) -> None:
        super().__init__(
            n_basis,
            n_sources=n_sources,
            partitioning=partitioning,
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            normalization=normalization,
            record_loss=record_loss,
            reference_id=reference_id,
            rng=rng,
        )
        assert diagonalizer_algorithm in diagonalizer_algorithms, "Not support {}.".format(
            diagonalizer_algorithm
        )
        assert not partitioning, "partitioning function is not supported."
        self.diagonalizer_algorithm = diagonalizer_algorithm
        if pair_selector is None:
            if diagonalizer_algorithm == "IP2":
                self.pair_selector = sequential_pair_selector
        else:
            self.pair_selector = pair_selector
        device_flooring(self.flooring_fn, allow_host=True, what="FastMNMF")

    def __repr__(self) -> str:
        s = "FastGaussMNMF(n_basis={}".format(self.n_basis)
        if self.n_sources is not None:
            s += ", n_sources={}".format(self.n_sources)
        if hasattr(self, "n_channels"):
            s += ", n_channels={}".format(self.n_channels)
        s += ", diagonalizer_algorithm={}, partitioning={}, record_loss={}, reference_id={})".format(
            self.diagonalizer_algorithm, self.partitioning, self.record_loss, self.reference_id
        )
        return s

    def _wiener(self, X):
        args = (X, self._state_dev("diagonalizer"), self._state_dev("spatial"),
                self._state_dev("basis"), self._state_dev("activation"), self.reference_id)
        if host_floor(self._floor) is not None:
            return _ops.fastmnmf_separate_host_floor(*args, self._floor.host, self._ws,
                                                     self._ws_bytes, self._info_tensor())
        return _ops.fastmnmf_separate(*args, self._floor, self._ws, self._ws_bytes,
                                      self._info_tensor())

    def _separate_dev(self) -> None:
        self._state_set_dev("output", self._wiener(self._X))

    def separate(self, input: np.ndarray) -> np.ndarray:
        """Multichannel Wiener filter with the current parameters (ref: mnmf.py:1174-1217)."""
        batched = input.ndim == 4
        self._check_separate_input(input)
        X = dv.to_device(input if batched else input[None], dtype=np.complex128)
        Y = self._wiener(X)
        self._check_device_errors()
        out = dv.to_host(Y)
        return out if batched else out[0]

    def _handover_valid(self) -> bool:
        return (getattr(self, "_handover", None) is not None and self._handover_key ==
                (self._state_rev("diagonalizer"), self._X.data_ptr()))

    def _loss_terms(self, data_out=None, logdet_out=None):
        """(data term, sum of log|det Q|) on the device; the data term from the |Q x|^2 hand-over
        while it is valid (half the bytes of the pass over X)."""
        Q = self._state_dev("diagonalizer")
        D, Tb, Vb = (self._state_dev(k) for k in ("spatial", "basis", "activation"))
        if self._handover_valid():
            data = _ops.fastmnmf_loss_data_handover(D, Tb, Vb, self._handover, self.n_channels,
                                                    self.n_frames, out=data_out)
        else:
            data = _ops.fastmnmf_loss_data(self._X, Q, D, Tb, Vb, out=data_out)
        if data_out is not None and logdet_out is None:
            return data, None  # (the resident loop: the log-determinants come from the updates)
        return data, _ops.sum_logdet(Q, out=logdet_out)

    def compute_loss(self) -> float:
        """ref: ssspy/bss/mnmf.py:1219-1261."""
        data, logdet = self._loss_terms()
        self._check_device_errors()
        values = dv.to_host(data) - 2.0 * dv.to_host(logdet)
        return values.copy() if self._batched else values[0].item()

    def _stock_ip1_iteration(self) -> bool:
        """update_once() is the one fused C-ABI call (stock step methods, IP1, power normalisation)."""
        cls = type(self)
        stock = all(
            getattr(cls, name) is getattr(FastGaussMNMF, name)
            for name in ("update_basis", "update_activation", "update_diagonalizer",
                         "update_spatial", "normalize", "normalize_by_power")
        )
        return stock and self.diagonalizer_algorithm in ["IP", "IP1"] and (
            not self.normalization or type(self.normalization) is bool
            or self.normalization == "power")

    def _iterate_with_resident_loss(self, n_iter: int, initial_call: bool) -> bool:
        """``record_loss=True`` (the reference's default) without a host round trip per iteration:
        the loss terms of every iteration stay in HBM and the list is assembled from one download
        at the end -- taken only when nothing can look at ``self.loss`` in between (no callbacks,
        stock ``update_once`` / ``compute_loss``); otherwise (returns False) the reference's loop
        runs unchanged.  ref: ssspy/bss/base.py:68-77, ssspy/bss/mnmf.py:1219-1261."""
        cls = type(self)
        if not (self.record_loss and not self.callbacks and n_iter > 0
                and self._stock_ip1_iteration()
                and cls.update_once is FastGaussMNMF.update_once
                and cls.compute_loss is FastGaussMNMF.compute_loss):
            return False
        B, dev = self._X.shape[0], self._X.device
        data = dv.zeros((n_iter + 1, B), dv.f64, dev)
        logdet = dv.zeros((n_iter + 1, B), dv.f64, dev)
        if host_floor(self._floor) is not None:
            if initial_call:
                self._loss_terms(data[0], logdet[0])
            for t in range(n_iter):
                self.update_once()
                self._loss_terms(data[t + 1], logdet[t + 1])
        else:
            # Round 6: sum_i log|det Q_i| of the state iteration t + 1 starts from is a by-product of
            # that iteration (the latency form of IP1 reads the diagonalisers anyway and leaves one
            # share per 16-bin tile; elsewhere the call stores the finished sum) -- folded once at
            # the end; only the last state needs sum_logdet.  The data term stays a pass per state.
            stride = (n_iter + 1) * B
            nld = _ops.fastmnmf_deferred_logdet_slots(B, self.n_sources, self.n_channels,
                                                      self.n_bins, self.n_frames, self.n_basis)
            ld = dv.zeros((nld, stride), dv.f64, dev) if nld > 1 else logdet
            ld_flat = ld.reshape(-1)
            steps = _lib.MNMF_ALL if self.normalization else _lib.MNMF_ALL & ~_lib.MNMF_NORMALIZE
            # the data term from the |Q x|^2 hand-over leaves its per-wave shares raw in one array
            # for the whole run (no memsets, no fold launch per loss)
            nds = _ops.fastmnmf_loss_handover_slots(B, self.n_sources, self.n_channels, self.n_bins,
                                                    self.n_frames, self.n_basis)
            ds = dv.zeros((nds, stride), dv.f64, dev) if nds and nds * stride * 8 <= (1 << 28) else None
            ds_flat = ds.reshape(-1) if ds is not None else None

            def data_term(t):
                if ds is not None and self._handover_valid():
                    _ops.fastmnmf_loss_data_handover_slots(
                        self._state_dev("spatial"), self._state_dev("basis"),
                        self._state_dev("activation"), self._handover, self.n_channels,
                        self.n_frames, ds_flat[t * B:], stride)
                else:
                    self._loss_terms(data[t], None)

            if initial_call:
                data_term(0)
            for t in range(n_iter):
                self._update(steps, "self", logdet=ld_flat[t * B:], logdet_stride=stride)
                data_term(t + 1)
            if nld > 1:
                _ops.fold_scalar_slots(ld, stride, nld, logdet.reshape(-1))
            _ops.sum_logdet(self._state_dev("diagonalizer"), out=logdet[n_iter])
            if ds is not None:  # (every state has its data term in `data` or in the shares)
                folded = dv.zeros((n_iter + 1, B), dv.f64, dev)
                _ops.fold_scalar_slots(ds, stride, nds, folded.reshape(-1))
                data = (data, folded)
        self._check_device_errors()
        if isinstance(data, tuple):
            data = dv.to_host(data[0]) + dv.to_host(data[1])
        else:
            data = dv.to_host(data)
        values = data - 2.0 * dv.to_host(logdet)
        if not initial_call:
            values = values[1:]
        self.loss.extend(v.copy() if self._batched else v[0].item() for v in values)
        return True

    def compute_logdet(self, diagonalizer: np.ndarray) -> np.ndarray:
        """log|det Q_i| per bin (ref: ssspy/bss/mnmf.py:1263-1276); host-side convenience."""
        return np.linalg.slogdet(diagonalizer)[1]

    def update_once(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:1278-1303; one C-ABI call for the whole iteration."""
        if self._stock_ip1_iteration():
            steps = _lib.MNMF_ALL if self.normalization else _lib.MNMF_ALL & ~_lib.MNMF_NORMALIZE
            self._update(steps, flooring_fn)
            return
        self.update_basis(flooring_fn=flooring_fn)
        self.update_activation(flooring_fn=flooring_fn)
        self.update_diagonalizer(flooring_fn=flooring_fn)
        self.update_spatial()
        if self.normalization:
            self.normalize(flooring_fn=flooring_fn)

    def update_basis(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:1305-1360."""
        self._update(_lib.MNMF_BASIS, flooring_fn)

    def update_activation(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:1362-1417."""
        self._update(_lib.MNMF_ACTIVATION, flooring_fn)

    def update_diagonalizer(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:1419-1447."""
        if self.diagonalizer_algorithm in ["IP", "IP1"]:
            self.update_diagonalizer_ip1(flooring_fn=flooring_fn)
        elif self.diagonalizer_algorithm in ["IP2"]:
            self.update_diagonalizer_ip2(flooring_fn=flooring_fn)
        else:
            raise NotImplementedError("Not support {}.".format(self.diagonalizer_algorithm))

    def update_diagonalizer_ip2(self, flooring_fn="self") -> None:
        """Pairwise iterative projection on Q.  ref: ssspy/bss/mnmf.py:1516-1633."""
        U = self._diagonalizer_covariance()
        _ops.update_by_ip2(self._state_dev("diagonalizer"), U,
                           resolve_pairs(getattr(self, "pair_selector", None), self.n_channels),
                           self._resolve_floor(flooring_fn), self._info_tensor())
        self._state_touch("diagonalizer")

    def update_diagonalizer_ip1(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:1449-1514."""
        self._update(_lib.MNMF_DIAGONALIZER, flooring_fn)

    def update_spatial(self) -> None:
        """ref: ssspy/bss/mnmf.py:1635-1675."""
        self._update(_lib.MNMF_SPATIAL, "self")


class MNMF(MNMFBase):
    """Full-rank spatial covariance MNMF state (ref: ssspy/bss/mnmf.py:300-414)."""

    spatial = Synced(dv.c128)

    def _reset(self, **kwargs) -> None:
        """ref: ssspy/bss/mnmf.py:139-165."""
        assert self._has_input(), "Specify data!"
        for key, value in kwargs.items():
            setattr(self, key, value)
        B, M, F, T = self._X.shape
        N = M if self.n_sources is None else self.n_sources
        self.n_sources, self.n_channels = N, M
        self.n_bins, self.n_frames = F, T
        self._floor = device_flooring(self.flooring_fn, what="GaussMNMF")
        require_device_floor(self._floor, "GaussMNMF")
        self._init_nmf(rng=self.rng)
        self._ws, self._ws_bytes = _ops.gmnmf_workspace(B, N, M, F, T, self.n_basis, self._X.device)
        if not getattr(self, "_skip_reset_output", False):
            self._separate_dev()

    def _init_nmf(self, rng=None) -> None:
        """NMF parameters, then H = I / M per source and bin.  ref: ssspy/bss/mnmf.py:327-353."""
        super()._init_nmf(rng=rng)
        N, M, F = self.n_sources, self.n_channels, self.n_bins
        if not self._state_has("spatial"):
            eye = np.eye(M, dtype=np.complex128) / M
            self.spatial = np.tile(eye, self._lead() + (N, F, 1, 1))
        else:
            self.spatial = np.array(self.spatial, dtype=np.complex128, copy=True)

    @property
    def instant_covariance(self) -> np.ndarray:
        """to_psd(x x^H), (n_bins, n_frames, M, M) (ref: ssspy/bss/mnmf.py:167-188).  The device
        kernels apply it in closed form; this host-side view exists for callbacks that read it."""
        X = self.input
        lead = X.ndim - 3
        XX = X[..., :, None, :, :] * X[..., None, :, :, :].conj()
        XX = np.moveaxis(XX, (lead, lead + 1), (-2, -1))
        kind, eps = self._floor
        lamb, P = np.linalg.eigh((XX + XX.swapaxes(-2, -1).conj()) / 2)
        lamb = np.maximum(lamb, eps) if kind == _lib.FLOOR_MAX else (
            lamb + eps if kind == _lib.FLOOR_ADD else lamb)
        out = (P * lamb[..., None, :]) @ P.swapaxes(-2, -1).conj()
        return (out + out.swapaxes(-2, -1).conj()) / 2

    def reconstruct_mnmf(self, basis, activation, spatial, latent=None) -> np.ndarray:
        """R_ij = sum_n lambda_nij H_ni (ref: ssspy/bss/mnmf.py:355-389); host-side convenience."""
        Lamb = self.reconstruct_nmf(basis, activation, latent=latent)
        return np.sum(Lamb[..., :, :, :, None, None] * spatial[..., :, :, None, :, :], axis=-5)

    def normalize(self, axis1=-2, axis2=-1) -> None:
        """Unit trace of H, scale moved into the basis (ref: ssspy/bss/mnmf.py:391-414)."""
        self._update(_lib.GMNMF_NORMALIZE)


class GaussMNMF(MNMF):
    """MNMF on a Gaussian distribution (ref: ssspy/bss/mnmf.py:681-1073).

    Args as the reference: ``n_basis``, ``n_sources`` (default: number of channels),
    ``partitioning`` (False on the device path), ``flooring_fn``, ``callbacks``,
    ``normalization``, ``record_loss``, ``reference_id``, ``rng``.  n_channels in [2, 8] (tuned for up to 4;
    above that the per-lane M x M kernels run out of registers and are an order of magnitude slower).
    """

    def __init__(
        self,
        n_basis: int,
        n_sources: Optional[int] = None,
        partitioning: bool = False,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        normalization: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        rng: Optional[np.random.Generator] = None,
    ) -> None:
        super().__init__(
            n_basis,
            n_sources=n_sources,
            partitioning=partitioning,
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            normalization=normalization,
            record_loss=record_loss,
            reference_id=reference_id,
            rng=rng,
        )
        device_flooring(self.flooring_fn, what="GaussMNMF")

    def __repr__(self) -> str:
        s = "GaussMNMF(n_basis={}".format(self.n_basis)
        if self.n_sources is not None:
            s += ", n_sources={}".format(self.n_sources)
        if hasattr(self, "n_channels"):
            s += ", n_channels={}".format(self.n_channels)
        s += ", partitioning={}, normalization={}, record_loss={}, reference_id={})".format(
            self.partitioning, self.normalization, self.record_loss, self.reference_id
        )
        return s

    def _resolve_floor(self, flooring_fn):
        if type(flooring_fn) is str and flooring_fn == "self":
            return self._floor
        return device_flooring(choose_flooring_fn(flooring_fn, method=self), what="GaussMNMF")

    def _update(self, steps, flooring_fn="self") -> None:
        latent = self._state_dev("latent") if self.partitioning else None
        _ops.gmnmf_update(self._X, self._state_dev("basis"), self._state_dev("activation"),
                          self._state_dev("spatial"), steps, self._resolve_floor(flooring_fn),
                          self._ws, self._ws_bytes, latent=latent)
        for name in ("basis", "activation", "spatial") + (("latent",) if self.partitioning else ()):
            self._state_touch(name)

    def _separate_dev(self) -> None:
        Y = _ops.gmnmf_separate(self._X, *self._nmf_pair(), self._state_dev("spatial"),
                                self.reference_id, self._floor)
        self._state_set_dev("output", Y)

    def separate(self, input: np.ndarray) -> np.ndarray:
        """Multichannel Wiener filter with the current parameters (ref: mnmf.py:729-763)."""
        batched = input.ndim == 4
        self._check_separate_input(input)
        X = dv.to_device(input if batched else input[None], dtype=np.complex128)
        Y = _ops.gmnmf_separate(X, *self._nmf_pair(), self._state_dev("spatial"),
                                self.reference_id, self._floor)
        self._check_device_errors()
        out = dv.to_host(Y)
        return out if batched else out[0]

    def compute_loss(self) -> float:
        """mean_j [tr(R^-1 XX) + log det R] summed over bins (ref: ssspy/bss/mnmf.py:765-804)."""
        data = _ops.gmnmf_loss(self._X, *self._nmf_pair(), self._state_dev("spatial"), self._floor)
        self._check_device_errors()
        values = dv.to_host(data)
        return values.copy() if self._batched else values[0].item()

    def compute_logdet(self, reconstructed: np.ndarray) -> np.ndarray:
        """log det R (ref: ssspy/bss/mnmf.py:791-804); host-side convenience."""
        return np.linalg.slogdet(reconstructed)[1]

    def update_once(self, flooring_fn="self") -> None:
        """basis, activation, spatial, unit-trace normalisation (ref: ssspy/bss/mnmf.py:806-834);
        one C-ABI call when the step methods are the stock ones."""
        cls = type(self)
        stock = all(
            getattr(cls, name) is getattr(GaussMNMF, name)
            for name in ("update_basis", "update_activation", "update_spatial", "normalize",
                         "update_latent")
        )
        if stock:
            steps = _lib.GMNMF_ALL if self.normalization else _lib.GMNMF_ALL & ~_lib.GMNMF_NORMALIZE
            if self.partitioning:
                steps |= _lib.GMNMF_LATENT
            self._update(steps, flooring_fn)
            return
        self.update_basis(flooring_fn=flooring_fn)
        self.update_activation(flooring_fn=flooring_fn)
        self.update_spatial(flooring_fn=flooring_fn)
        if self.normalization:
            self.normalize(axis1=-2, axis2=-1)
        if self.partitioning:
            self.update_latent(flooring_fn=flooring_fn)

    def update_basis(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:836-901."""
        self._update(_lib.GMNMF_BASIS, flooring_fn)

    def update_activation(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/mnmf.py:903-968."""
        self._update(_lib.GMNMF_ACTIVATION, flooring_fn)

    def update_spatial(self, flooring_fn="self") -> None:
        """H <- to_psd(P^-1 # H Q H) (ref: ssspy/bss/mnmf.py:970-1016)."""
        self._update(_lib.GMNMF_SPATIAL, flooring_fn)

    def update_latent(self, flooring_fn="self") -> None:
        """z_nk <- z_nk sqrt(num / den), columns renormalised (ref: ssspy/bss/mnmf.py:1018-1073)."""
        assert self.partitioning, "update_latent needs partitioning=True."
        self._update(_lib.GMNMF_LATENT, flooring_fn)
