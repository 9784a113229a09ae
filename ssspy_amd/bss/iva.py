"""Auxiliary-function independent vector analysis (AuxIVA) on MI355X.

Drop-in separator classes for the ``AuxIVA`` family of the reference's ``ssspy.bss.iva``
(ssspy/bss/iva.py:553-641, :1403-2214, :2976-3473): ``AuxIVA``, ``AuxLaplaceIVA`` and
``AuxGaussIVA`` with ``spatial_algorithm in {"IP", "IP1", "ISS", "ISS1"}``.  The contrast
functions of the reference are Python closures; the kernels implement the two the
reference ships (Laplace: G = 2r; time-varying Gauss: G = F log(alpha) + r^2/alpha).  A
user-supplied ``d_contrast_fn`` is evaluated on the host on the (n_sources, n_frames) frame norms
the device produces (a few KB per iteration; both passes over the spectrogram stay in the kernels);
a user-supplied ``contrast_fn`` takes the whole estimate and is evaluated on a host copy of it,
only when the loss is recorded.
The gradient / natural-gradient / Fast / PDS / ADMM IVA variants are out of scope
(SURVEY.md section 2, row 3).
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

__all__ = ["AuxIVA", "AuxLaplaceIVA", "AuxGaussIVA"]

spatial_algorithms = ["IP", "IP1", "IP2", "ISS", "ISS1", "ISS2", "IPA"]
EPS = 1e-10

_IP1 = ("IP", "IP1")
_ISS1 = ("ISS", "ISS1")
_IP2 = ("IP2",)
_ISS2 = ("ISS2",)
_IPA = ("IPA",)
_PROJECTION_BACK = ("projection_back",)
_MDP = ("minimal_distortion_principle",)


class IVABase(DeviceStateMixin, IterativeMethodBase):
    """ref: ssspy/bss/iva.py:48-281."""

    demix_filter = Synced(dv.c128)
    output = Synced(dv.c128)

    def __init__(
        self,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        callbacks=None,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
    ) -> None:
        super().__init__(callbacks=callbacks, record_loss=record_loss)
        self.flooring_fn = identity if flooring_fn is None else flooring_fn
        self.input = None
        self.scale_restoration = scale_restoration
        if reference_id is None and scale_restoration:
            raise ValueError("Specify 'reference_id' if scale_restoration=True.")
        self.reference_id = reference_id

    def _reset(self, **kwargs) -> None:
        """ref: ssspy/bss/iva.py:138-169."""
        assert self._has_input(), "Specify data!"
        for key, value in kwargs.items():
            setattr(self, key, value)
        B, N, F, T = self._X.shape
        self.n_sources, self.n_channels = N, N
        self.n_bins, self.n_frames = F, T
        if not self._state_has("demix_filter"):
            self.demix_filter = np.tile(np.eye(N, dtype=np.complex128), self._lead() + (F, 1, 1))
        elif not self._state_is_none("demix_filter"):
            self.demix_filter = np.array(self.demix_filter, dtype=np.complex128, copy=True)
        if self._state_is_none("demix_filter"):
            raise ValueError("demix_filter=None cannot be given at reset.")
        self._state_set_dev("output", _ops.separate(self._X, self._state_dev("demix_filter")))
        self._floor = device_flooring(self.flooring_fn, allow_host=True)

    def separate(self, input: np.ndarray, demix_filter: np.ndarray) -> np.ndarray:
        """y_ij = W_i x_ij (ref: ssspy/bss/iva.py:171-194); NumPy in, NumPy out."""
        batched = input.ndim == 4
        X = dv.to_device(input if batched else input[None], dtype=np.complex128)
        W = dv.to_device(demix_filter if batched else demix_filter[None], dtype=np.complex128)
        Y = dv.to_host(_ops.separate(X, W))
        return Y if batched else Y[0]

    def _uses_filter(self) -> bool:
        return not self._state_is_none("demix_filter")

    def _implied_filter(self):
        """W with output = W x while nothing else rewrote ``output`` since, else None (the ISS2 / IPA
        iterations of AuxIVA keep it, see AuxIVA._update_once_implied)."""
        kept = getattr(self, "_implied", None)
        if (kept is None or kept[1] != self._state_rev("output")
                or not _routes.get("implied_filter")):
            return None
        return kept[0]

    def _fill_output_from_implied_filter(self) -> None:
        _ops.separate(self._X, self._implied[0], out=self._state()["output"]["dev"])

    def _leave_implied_route(self) -> None:
        """Form Y = W x now and go on with the iterations that rewrite it (the reference's)."""
        W = self._implied[0]
        self._state_dev("output")  # (runs the deferred fill)
        if getattr(self, "_logdet_cache", None) is not None:
            self._logdet_cache = (_ops.sum_logdet(W), self._state_rev("output"))
        self._implied = None
        self._r2_cache = None

    def _resolve_floor(self, flooring_fn):
        if type(flooring_fn) is str and flooring_fn == "self":
            return self._floor
        return device_flooring(choose_flooring_fn(flooring_fn, method=self), allow_host=True)

    def _host_loss(self, data, logdet_sum):
        self._check_device_errors()
        values = dv.to_host(data) - 2.0 * dv.to_host(logdet_sum)
        return values.copy() if self._batched else values[0].item()

    def restore_scale(self) -> None:
        """ref: ssspy/bss/iva.py:238-257."""
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
        """ref: ssspy/bss/iva.py:259-267, :2194-2204; algorithm/projection_back.py:87-121."""
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
        ref: ssspy/bss/iva.py:269-281, :2206-2214; algorithm/minimal_distortion_principle.py:6-43."""
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


class AuxIVABase(IVABase):
    """ref: ssspy/bss/iva.py:553-641."""

    def __init__(
        self,
        contrast_fn: Callable[[np.ndarray], np.ndarray] = None,
        d_contrast_fn: Callable[[np.ndarray], np.ndarray] = None,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        callbacks=None,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
    ) -> None:
        super().__init__(
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
        )
        self.contrast_fn = contrast_fn
        self.d_contrast_fn = d_contrast_fn


def _device_contrast(contrast_fn, d_contrast_fn):
    """Which built-in contrast the pair of callables stands for (tag set by the subclasses), or
    None for user closures.  A closure cannot run inside a kernel, and it does not have to: the
    reference applies ``d_contrast_fn`` to the (n_sources, n_frames) frame norms only
    (ssspy/bss/iva.py:1787-1789), a few KB that the frame-power kernel produces -- the closure runs
    on the host on that array between the two passes over the spectrogram, which stay on the device."""
    a = getattr(contrast_fn, "_ssspy_amd_contrast", None)
    b = getattr(d_contrast_fn, "_ssspy_amd_contrast", None)
    if a is None or a != b:
        if d_contrast_fn is None:
            raise ValueError("Specify d_contrast_fn (and contrast_fn when record_loss=True).")
        return None
    return a


class AuxIVA(AuxIVABase):
    """Auxiliary-function-based IVA (ref: ssspy/bss/iva.py:1403-2214)."""

    _ipa_default_kwargs = {"lqpqm_normalization": True, "newton_iter": 1}
    _default_kwargs = _ipa_default_kwargs

    def __init__(
        self,
        spatial_algorithm: str = "IP",
        contrast_fn: Callable[[np.ndarray], np.ndarray] = None,
        d_contrast_fn: Callable[[np.ndarray], np.ndarray] = None,
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        **kwargs,
    ) -> None:
        super().__init__(
            contrast_fn=contrast_fn,
            d_contrast_fn=d_contrast_fn,
            flooring_fn=flooring_fn,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
        )
        assert spatial_algorithm in spatial_algorithms, "Not support {}.".format(spatial_algorithm)
        self.spatial_algorithm = spatial_algorithm
        if pair_selector is None:
            if spatial_algorithm in ["IP2", "ISS2"]:
                self.pair_selector = sequential_pair_selector
        else:
            self.pair_selector = pair_selector
        valid_keys = set(self._ipa_default_kwargs) if spatial_algorithm == "IPA" else set()
        invalid_keys = set(kwargs) - valid_keys
        assert invalid_keys == set(), "Invalid keywords {} are given.".format(invalid_keys)
        for key, value in kwargs.items():
            setattr(self, key, value)
        for key in valid_keys:
            if not hasattr(self, key):
                setattr(self, key, self._default_kwargs[key])
        device_flooring(self.flooring_fn, allow_host=True)

    def __call__(
        self, input: np.ndarray, n_iter: int = 100, initial_call: bool = True, **kwargs
    ) -> np.ndarray:
        """Separate a frequency-domain multichannel mixture (ref: ssspy/bss/iva.py:1637-1672)."""
        self._contrast = _device_contrast(self.contrast_fn, self.d_contrast_fn)
        self._bind_input(input)
        self._reset(**kwargs)
        if not self._iterate_with_resident_loss(int(n_iter), initial_call):
            IterativeMethodBase.__call__(self, n_iter=n_iter, initial_call=initial_call)
        if self.scale_restoration:
            self.restore_scale()
        if self._uses_filter():
            self._state_set_dev("output", _ops.separate(self._X, self._state_dev("demix_filter")))
        return self._final_output()

    def _iterate_with_resident_loss(self, n_iter: int, initial_call: bool) -> bool:
        """IP1 with ``record_loss=True`` (the reference's default) at the cost of ``record_loss=False``.

        ``compute_loss()`` needs the frame powers of the current estimate -- a pass over the mixture
        -- and so does the next ``update_once()``: here the loss of the state after iteration t is
        taken from the frame powers iteration t + 1 forms anyway, all loss terms stay in HBM, and
        the list is assembled from one download at the end.  Only when nothing can look at
        ``self.loss`` in between (no callbacks, stock methods, a contrast that runs on the device
        and keeps no variance state); otherwise (returns False) the reference's loop runs unchanged.
        ref: ssspy/bss/base.py:68-77, ssspy/bss/iva.py:200-222, :1736-1793."""
        cls = type(self)
        if not (self.record_loss and not self.callbacks and n_iter > 0
                and self.spatial_algorithm in _IP1 and self._contrast is not None
                and host_floor(self._floor) is None
                and self._variance_tensor() is None
                and cls.update_once is AuxIVA.update_once
                and cls.update_once_ip1 is AuxIVA.update_once_ip1
                and cls.compute_loss is AuxIVA.compute_loss):
            return self._iterate_with_resident_terms(n_iter, initial_call)
        B, dev, N = self._X.shape[0], self._X.device, self.n_sources
        data = dv.zeros((n_iter + 1, B), dv.f64, dev)
        logdet = dv.zeros((n_iter + 1, B), dv.f64, dev)
        W = self._state_dev("demix_filter")
        floor = self._resolve_floor("self")
        # (round 6) sum_i log|det W_i| of the state an update starts from is a by-product of that
        # update: one share per 16-bin tile from the latency form of IP1 for a handful of mixtures
        # (the one-block sum_logdet launch was 13 us of a 65 us iteration), else the finished sum;
        # folded once at the end, only the last state needs sum_logdet
        stride = (n_iter + 1) * B
        nld = _ops.update_by_ip1_logdet_slots(B, self.n_bins, N)
        ld = dv.zeros((nld, stride), dv.f64, dev) if nld > 1 else logdet
        ld_flat = ld.reshape(-1)
        for t in range(n_iter + 1):
            r2 = _ops.iva_frame_power(self._X, W)
            if t > 0 or initial_call:
                _ops.iva_loss_data(r2, None, self.n_bins, self._contrast, out=data[t])
            if t == n_iter:
                break
            weight = _ops.iva_weight(r2, self.n_bins, self._contrast, floor, variance=None)
            U = _ops.weighted_covariance(self._X, weight, _lib.WEIGHT_FRAME, N)
            _ops.update_by_ip1_logdet(W, U, floor, self._info_tensor(), ld_flat[t * B:], stride)
        if nld > 1:
            _ops.fold_scalar_slots(ld, stride, nld, logdet.reshape(-1))
        _ops.sum_logdet(W, out=logdet[n_iter])
        self._state_touch("demix_filter")
        self._check_device_errors()
        values = dv.to_host(data) - 2.0 * dv.to_host(logdet)
        if not initial_call:
            values = values[1:]
        self.loss.extend(v.copy() if self._batched else v[0].item() for v in values)
        return True

    def _iterate_with_resident_terms(self, n_iter: int, initial_call: bool) -> bool:
        """The same for every other spatial algorithm (round 6): ``compute_loss()`` downloads two
        numbers per mixture, and the wait for them drained the queue after every iteration -- one
        mixture of configs[2] (ISS, 8 sources) spent 0.9-1.2 ms of wall time per 0.2 ms iteration.
        Here the terms of every iteration stay in HBM (the fused ISS sweep's frame powers and
        tracked log-determinant serve as they come) and one download assembles the list.  Only
        with the library's own ``update_once`` / ``compute_loss`` and a contrast that runs on the
        device; otherwise (returns False) the reference's loop runs unchanged."""
        cls = type(self)
        if not (self.record_loss and not self.callbacks and n_iter > 0
                and self._contrast is not None
                and cls.update_once in (AuxIVA.update_once, AuxGaussIVA.update_once)
                and cls.compute_loss is AuxIVA.compute_loss):
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

    def __repr__(self) -> str:
        s = "AuxIVA(spatial_algorithm={}, scale_restoration={}, record_loss={}".format(
            self.spatial_algorithm, self.scale_restoration, self.record_loss
        )
        if self.scale_restoration:
            s += ", reference_id={}".format(self.reference_id)
        return s + ")"

    def _reset(self, **kwargs) -> None:
        """ref: ssspy/bss/iva.py:1687-1697."""
        super()._reset(**kwargs)
        # (also here, not only in __call__: update_once() after a manual _bind_input() / _reset() --
        #  the benchmarks do that, and the reference allows it -- must find the contrast code)
        self._contrast = _device_contrast(self.contrast_fn, self.d_contrast_fn)
        self._logdet_cache = None
        self._implied = None
        B, N, F, T = self._X.shape
        self._amp_reset()
        if self.spatial_algorithm in _ISS2 + _IPA and N <= 4:
            # the filters the output state implies (output = W x): the ISS2 / IPA iterations read the
            # mixture through them (_update_once_implied).  Up to 4 sources (the tuned covariance
            # pass).  W U W^H rounds like eps |W|^2 |U|, the direct sum over Y like eps |y|^2: next
            # to singular covariances lose digits -- every launch measures by how much and the
            # route is left past the bound (_amp_exceeded, see ilrma.py)
            self._implied = (self._state_dev("demix_filter").clone(), self._state_rev("output"))
        if self.spatial_algorithm in ["ISS", "ISS1", "ISS2", "IPA"] and not self.record_loss:
            self.demix_filter = None  # (nothing reads the log-determinant: no tracker)
        elif self.spatial_algorithm in ["ISS", "ISS1", "ISS2", "IPA"]:
            # sum_i log|det W_i| of the filters the ISS state stops carrying, as (tensor, revision of
            # `output` it describes): the fused sweep kernel moves it along (each sweep multiplies
            # det W_i by d_in^(-1/2)), so compute_loss() need not rebuild W from Y X^H
            self._logdet_cache = (_ops.sum_logdet(self._state_dev("demix_filter")),
                                  self._state_rev("output"))
            self.demix_filter = None
        # frame powers of the output (ISS state) as (tensor, revision of `output` they describe): any
        # later write to the output -- a kernel, scale restoration, an assignment by a callback --
        # changes the revision and retires the cache
        self._r2_cache = None

    def _variance_tensor(self):
        return None

    def _weights(self, flooring_fn):
        """Auxiliary weights varphi_nj = G'(r_nj) / floor(2 r_nj), (B, N, T)."""
        return self._weights_from_power(self._frame_power(), self._contrast, flooring_fn)

    def _weights_from_power(self, r2, contrast, flooring_fn):
        floor = self._resolve_floor(flooring_fn)
        if contrast is None or host_floor(floor) is not None:
            # a user closure, or a flooring callable the kernels cannot run: on the host, on the
            # (n_sources, n_frames) norms
            if self._variance_tensor() is not None:
                return self._host_weights_gauss(r2, contrast, flooring_fn)
            return self._host_weights(r2, flooring_fn)
        return _ops.iva_weight(r2, self.n_bins, contrast, self._resolve_floor(flooring_fn),
                               variance=self._variance_tensor())

    def _host_weights(self, r2, flooring_fn):
        """User closure on the frame norms: r (n_sources, n_frames) per mixture comes down, the
        weights go back up (ref: ssspy/bss/iva.py:1787-1789, :1962-1964)."""
        if type(flooring_fn) is str and flooring_fn == "self":
            flooring_fn = self.flooring_fn
        flooring_fn = choose_flooring_fn(flooring_fn, method=self)
        self._check_device_errors()
        r = np.sqrt(dv.to_host(r2))  # (B, N, T)
        weight = np.stack([np.asarray(self.d_contrast_fn(rb) / flooring_fn(2 * rb), dtype=np.float64)
                           for rb in r])
        if weight.shape != r.shape:
            raise ValueError("d_contrast_fn must map (n_sources, n_frames) to the same shape.")
        return dv.to_device(weight, dtype=np.float64, dev=r2.device)

    def _host_weights_gauss(self, r2, contrast, flooring_fn):
        """AuxGaussIVA with a flooring callable the kernels cannot run: the variance refresh
        alpha = r^2 / n_bins (ssspy/bss/iva.py:3465-3473) unless the pair loop of IP2 keeps it fixed,
        then varphi = (2 r / alpha) / flooring_fn(2 r) on the (n_sources, n_frames) norms
        (:3273-3288, :1787-1789) -- host arithmetic on B N T numbers, as for user closures."""
        flooring_fn = choose_flooring_fn(self.flooring_fn if (type(flooring_fn) is str and
                                                              flooring_fn == "self") else flooring_fn,
                                         method=self)
        self._check_device_errors()
        r2h = dv.to_host(r2)  # (B, N, T)
        var_dev = self._variance_tensor()
        if contrast == _lib.CONTRAST_GAUSS_FIXED:
            var = dv.to_host(var_dev)
        else:
            var = r2h / float(self.n_bins)
            var_dev.copy_(dv.to_device(var, dtype=np.float64, dev=r2.device))
        r = np.sqrt(r2h)
        weight = np.stack([np.asarray((2 * rb / vb) / flooring_fn(2 * rb), dtype=np.float64)
                           for rb, vb in zip(r, var)])
        return dv.to_device(weight, dtype=np.float64, dev=r2.device)

    def _frame_power(self):
        """r_nj^2 = sum_i |y_nij|^2 of the current estimate, (B, N, T)."""
        if self._uses_filter():
            return _ops.iva_frame_power(self._X, self._state_dev("demix_filter"))
        cache = self._r2_cache
        if cache is None or cache[1] != self._state_rev("output"):
            W = self._implied_filter()
            if W is not None:
                r2 = _ops.iva_frame_power(self._X, W)
            else:
                r2 = _ops.iva_frame_power(self._state_dev("output"), None)
            cache = (r2, self._state_rev("output"))
            self._r2_cache = cache
        return cache[0]

    def update_once(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/iva.py:1699-1734."""
        if self.spatial_algorithm in _IP1:
            self.update_once_ip1(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _ISS1:
            self.update_once_iss1(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _IP2:
            self.update_once_ip2(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _ISS2:
            self.update_once_iss2(flooring_fn=flooring_fn)
        elif self.spatial_algorithm in _IPA:
            self.update_once_ipa(flooring_fn=flooring_fn)
        else:
            raise NotImplementedError("Not support {}.".format(self.spatial_algorithm))

    def update_once_ipa(self, flooring_fn="self") -> None:
        """Iterative projection with adjustment.  ref: ssspy/bss/iva.py:2068-2175."""
        require_device_floor(self._resolve_floor(flooring_fn), "IPA")
        if self._update_once_implied(flooring_fn):
            return
        Y = self._state_dev("output")
        weight = self._weights(flooring_fn)
        r2 = _ops.update_by_ipa(Y, weight, _lib.WEIGHT_FRAME, self.lqpqm_normalization,
                                self.newton_iter, self._resolve_floor(flooring_fn),
                                self._info_tensor(), not_converged=self._newton_counter(),
                                frame_power=True)
        self._state_touch("output")
        self._r2_cache = None if r2 is None else (r2, self._state_rev("output"))

    def _pair_weight_contrast(self):
        """Contrast code for the per-pair weights of IP2 (the Gauss model keeps its variance)."""
        return self._contrast

    def update_once_ip2(self, flooring_fn="self") -> None:
        """Pairwise iterative projection; the auxiliary weights are recomputed for every pair from
        the current filters.  ref: ssspy/bss/iva.py:1795-1915."""
        N = self.n_sources
        floor = self._resolve_floor(flooring_fn)
        W = self._state_dev("demix_filter")
        for m, n in resolve_pairs(getattr(self, "pair_selector", None), N):
            r2 = _ops.iva_frame_power(self._X, W)
            weight = self._weights_from_power(r2, self._pair_weight_contrast(), flooring_fn)
            B, _, F, _ = self._X.shape
            if N <= 4 and B * ((F + 31) // 32) >= 512:
                # (batches: the tuned frame-weight covariance forms all N sets in 241 us where the
                #  generic kernel takes 325 for the pair's two -- 32 mixtures of configs[1])
                U = _ops.weighted_covariance(self._X, weight, _lib.WEIGHT_FRAME, N)
                _ops.update_by_ip2(W, U, [(m, n)], floor, self._info_tensor())
                continue
            w_pair = weight[:, [m, n], :].contiguous()  # gather of two rows (data movement only)
            U_pair = _ops.weighted_covariance(self._X, w_pair, _lib.WEIGHT_FRAME, 2)
            _ops.update_by_ip2(W, U_pair, [(m, n)], floor, self._info_tensor(), pair_only=True)
        self._state_touch("demix_filter")

    def update_once_iss2(self, flooring_fn="self") -> None:
        """Pairwise iterative source steering.  ref: ssspy/bss/iva.py:1968-2066."""
        N = self.n_sources
        if self._update_once_implied(flooring_fn):
            return
        Y = self._state_dev("output")
        floor = self._resolve_floor(flooring_fn)
        weight = self._weights(flooring_fn)
        Vc = _ops.weighted_covariance(Y, weight, _lib.WEIGHT_FRAME, N)
        G = _ops.iss2_transform(Vc, resolve_pairs(getattr(self, "pair_selector", None), N), floor,
                                self._info_tensor())
        self._separate_output(Y, G)

    def _update_once_implied(self, flooring_fn) -> bool:
        """ISS2 / IPA iteration without touching Y (round 5).  The reference keeps only the separated
        spectrogram and rewrites it (iva.py:1968-2175): weights from its frame powers, statistics
        mean phi y y^H, Y <- G Y -- three spectrogram-sized transfers per iteration even with the
        frame powers taken in the rewrite.  With output = W x the frame powers are |W x|^2 (the IP1
        pass), the statistics W U W^H with the weighted covariances U of the MIXTURE, the update
        W <- G W on (F, N, N): two read-only passes, Y formed when ``output`` is read
        (_state_defer).  False: not applicable, the caller runs the literal form."""
        W = self._implied_filter()
        floor = self._resolve_floor(flooring_fn)
        if W is None or host_floor(floor) is not None or self._contrast is None:
            return False
        if self._amp_exceeded():
            self._leave_implied_route()
            return False
        B, N, F, T = self._X.shape
        dev = self._X.device
        weight = self._weights(flooring_fn)  # (frame powers through _frame_power(): |W x|^2)
        U = _ops.weighted_covariance(self._X, weight, _lib.WEIGHT_FRAME, N)
        Vc = getattr(self, "_Vc_implied", None)
        if Vc is None or tuple(Vc.shape) != tuple(U.shape) or Vc.data_ptr() == U.data_ptr():
            Vc = self._Vc_implied = dv.empty(tuple(U.shape), dv.c128, dev)
        tracked = self._amp_tracked(self._C())
        _ops.covariance_congruence(U, W, Vc, tracked=tracked)
        self._amp_launched(tracked)
        if self.spatial_algorithm in _ISS2:
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
        self._state_defer("output", self._fill_output_from_implied_filter)
        self._implied, self._implied_spare = (spare, self._state_rev("output")), W
        self._r2_cache = None
        return True

    def update_once_ip1(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/iva.py:1736-1793."""
        N = self.n_sources
        weight = self._weights(flooring_fn)
        U = _ops.weighted_covariance(self._X, weight, _lib.WEIGHT_FRAME, N)
        _ops.update_by_ip1(self._state_dev("demix_filter"), U, self._resolve_floor(flooring_fn),
                           self._info_tensor())
        self._state_touch("demix_filter")

    def update_once_iss1(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/iva.py:1917-1966 and _update_spatial_model.py:146-194."""
        N = self.n_sources
        Y = self._state_dev("output")
        weight = self._weights(flooring_fn)
        floor = self._resolve_floor(flooring_fn)
        if host_floor(floor) is not None:
            _ops.update_by_iss1_host_floor(Y, weight, _lib.WEIGHT_FRAME, floor.host)
            self._state_touch("output")
            self._logdet_cache = None
        elif self.n_frames <= _ops.iss1_fused_max_frames(N):
            # one read + one write of Y; the kernel also leaves the next iteration's frame powers
            r2_next = dv.empty(tuple(weight.shape), dv.f64, Y.device)
            tracked = self._tracked_logdet()
            _ops.iss1_fused(Y, weight, _lib.WEIGHT_FRAME, floor, r2_next, logdet=tracked)
            self._state_touch("output")
            self._r2_cache = (r2_next, self._state_rev("output"))
            self._logdet_cache = None if tracked is None else (tracked, self._state_rev("output"))
        else:
            Vc = _ops.weighted_covariance(Y, weight, _lib.WEIGHT_FRAME, N)
            G = _ops.iss1_transform(Vc, floor)
            self._separate_output(Y, G)

    def _separate_output(self, Y, G) -> None:
        """Y <- G Y in place; the walk also leaves sum_i |y|^2 of the new Y, which the next
        iteration's weights would otherwise fetch with a pass of their own (round 5)."""
        r2 = _ops.separate_frame_power(Y, G)
        if r2 is None:
            _ops.separate(Y, G, out=Y)
        self._state_touch("output")
        self._r2_cache = None if r2 is None else (r2, self._state_rev("output"))

    def _tracked_logdet(self):
        """The tracked sum_i log|det W_i| if it describes the current output, else None."""
        cache = getattr(self, "_logdet_cache", None)
        if cache is not None and cache[1] == self._state_rev("output"):
            return cache[0]
        return None

    def _logdet_sum(self):
        """sum_i log|det W_i| (B,) on the device, and the filters if they had to be formed."""
        if self._uses_filter():
            W = self._state_dev("demix_filter")
            return _ops.sum_logdet(W), W
        W = self._implied_filter()
        if W is not None:
            return _ops.sum_logdet(W), W
        tracked = self._tracked_logdet()
        if tracked is not None:
            return tracked, None
        W = _ops.demix_from_covariance(_ops.cross_covariance(self._state_dev("output"), self._X),
                                       self._C(), self._info_tensor())
        return _ops.sum_logdet(W), W

    def compute_loss(self) -> float:
        """ref: ssspy/bss/iva.py:200-222 (filter state), :2177-2192 (ISS state)."""
        if self._contrast is None:
            logdet, W = self._logdet_sum()
            return self._host_contrast_loss(W, logdet)
        return self._host_loss(*self._loss_terms())

    def _loss_terms(self, data_out=None, logdet_out=None):
        """(contrast term, sum_i log|det W_i|) of the current state on the device, each (B,)."""
        logdet, _ = self._logdet_sum()
        data = _ops.iva_loss_data(self._frame_power(), self._variance_tensor(), self.n_bins,
                                  self._contrast, out=data_out)
        if logdet_out is not None:
            logdet_out.copy_(logdet)  # (the tracked sum is moved in place by the next sweep)
            logdet = logdet_out
        return data, logdet

    def _host_contrast_loss(self, W, logdet_dev):
        """Loss with a user ``contrast_fn``: the closure takes the whole separated spectrogram
        (ssspy/bss/iva.py:216, :2181), so the estimate crosses PCIe once per recorded loss -- the
        price of an opaque Python callable, paid only with ``record_loss=True``."""
        if self.contrast_fn is None:
            raise ValueError("Specify contrast_fn to record the loss.")
        if self._uses_filter():
            Y = dv.to_host(_ops.separate(self._X, W))
        else:
            Y = dv.to_host(self._state_dev("output"))
        self._check_device_errors()
        logdet = dv.to_host(logdet_dev)
        values = np.array([np.sum(np.mean(self.contrast_fn(Yb), axis=1), axis=0) for Yb in Y])
        values = values - 2.0 * logdet
        return values.copy() if self._batched else values[0].item()


class AuxLaplaceIVA(AuxIVA):
    """AuxIVA with the spherical Laplace source model (ref: ssspy/bss/iva.py:2976-3128)."""

    def __init__(
        self,
        spatial_algorithm: str = "IP",
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        **kwargs,
    ) -> None:
        def contrast_fn(y: np.ndarray) -> np.ndarray:
            """G(y) = 2 ||y||_2 over bins; y (n_sources, n_bins, n_frames)."""
            return 2 * np.linalg.norm(y, axis=1)

        def d_contrast_fn(y: np.ndarray) -> np.ndarray:
            """G'(r) = 2."""
            return 2 * np.ones_like(y)

        contrast_fn._ssspy_amd_contrast = _lib.CONTRAST_LAPLACE
        d_contrast_fn._ssspy_amd_contrast = _lib.CONTRAST_LAPLACE
        super().__init__(
            spatial_algorithm=spatial_algorithm,
            contrast_fn=contrast_fn,
            d_contrast_fn=d_contrast_fn,
            flooring_fn=flooring_fn,
            pair_selector=pair_selector,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
            **kwargs,
        )

    def __repr__(self) -> str:
        return "AuxLaplaceIVA" + super().__repr__()[len("AuxIVA"):]


class AuxGaussIVA(AuxIVA):
    """AuxIVA with the time-varying Gauss source model (ref: ssspy/bss/iva.py:3131-3473)."""

    variance = Synced(dv.f64)

    def __init__(
        self,
        spatial_algorithm: str = "IP",
        flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = functools.partial(
            max_flooring, eps=EPS
        ),
        pair_selector: Optional[Callable[[int], Iterable[Tuple[int, int]]]] = None,
        callbacks: Optional[Union[Callable, List[Callable]]] = None,
        scale_restoration: Union[bool, str] = True,
        record_loss: bool = True,
        reference_id: int = 0,
        **kwargs,
    ) -> None:
        def contrast_fn(y: np.ndarray) -> np.ndarray:
            """G(y) = n_bins log(alpha) + ||y||^2 / alpha."""
            norm = np.linalg.norm(y, axis=1)
            return self.n_bins * np.log(self.variance) + (norm**2) / self.variance

        def d_contrast_fn(y: np.ndarray, variance: np.ndarray = None) -> np.ndarray:
            """G'(r) = 2 r / alpha."""
            alpha = self.variance if variance is None else variance
            return 2 * y / alpha

        contrast_fn._ssspy_amd_contrast = _lib.CONTRAST_GAUSS
        d_contrast_fn._ssspy_amd_contrast = _lib.CONTRAST_GAUSS
        super().__init__(
            spatial_algorithm=spatial_algorithm,
            contrast_fn=contrast_fn,
            d_contrast_fn=d_contrast_fn,
            flooring_fn=flooring_fn,
            pair_selector=pair_selector,
            callbacks=callbacks,
            scale_restoration=scale_restoration,
            record_loss=record_loss,
            reference_id=reference_id,
            **kwargs,
        )

    def __repr__(self) -> str:
        return "AuxGaussIVA" + super().__repr__()[len("AuxIVA"):]

    def _reset(self, **kwargs) -> None:
        """ref: ssspy/bss/iva.py:3304-3317."""
        super()._reset(**kwargs)
        self.variance = np.ones(self._lead() + (self.n_sources, self.n_frames))

    def _variance_tensor(self):
        return self._state_dev("variance")

    def update_once(self, flooring_fn="self") -> None:
        """Refresh the variance, then the spatial update (ref: ssspy/bss/iva.py:3319-3337).

        The variance refresh alpha_nj = mean_i |y_nij|^2 (:3465-3473) is fused into the weight
        kernel (it is r^2 / n_bins of the same frame power).
        """
        super().update_once(flooring_fn=flooring_fn)
        self._state_touch("variance")

    def update_source_model(self) -> None:
        """alpha_nj = mean_i |y_nij|^2 (ref: ssspy/bss/iva.py:3465-3473)."""
        if host_floor(self._floor) is not None:  # (no floor acts on the variance itself)
            var = dv.to_host(self._frame_power()) / float(self.n_bins)
            self._variance_tensor().copy_(dv.to_device(var, dtype=np.float64))
        else:
            _ops.iva_weight(self._frame_power(), self.n_bins, _lib.CONTRAST_GAUSS, self._floor,
                            variance=self._variance_tensor())
        self._state_touch("variance")

    def _pair_weight_contrast(self):
        return _lib.CONTRAST_GAUSS_FIXED

    def update_once_ip2(self, flooring_fn="self") -> None:
        """ref: ssspy/bss/iva.py:3319-3337 (variance refresh) + :3339-3463 (pairs, fixed variance)."""
        self.update_source_model()
        super().update_once_ip2(flooring_fn=flooring_fn)
