"""Host-visible attributes backed by HBM buffers.

The reference keeps every piece of separator state (``demix_filter``, ``output``,
``basis`` ...) as a NumPy attribute that users, callbacks and tests read -- and inject
through ``__call__(**kwargs)``.  Here the truth lives on the device between iterations; a
``Synced`` attribute downloads lazily on read (one stream sync) and uploads lazily before
the next kernel that needs it.  Internally every buffer carries a leading batch axis B of
independent mixtures; when the separator was called with a single 3-D mixture that axis is
hidden from the host view so the shapes are exactly the reference's.
"""

import numpy as np
import torch

from .. import _device as dv
from .. import _lib


class HostSnapshot(np.ndarray):
    """Read-only host copy of a device-resident state variable.

    The device buffer stays the truth, so writing into the snapshot (``m.basis[0] = 1``) raises
    instead of being lost.  Augmented assignment on the attribute still works as in the
    reference: ``m.basis *= 2`` evaluates getter -> ``__imul__`` -> setter, and the in-place
    operators here compute out of place and hand a fresh plain array to the setter, which
    uploads it.  Every other result (ufuncs, slices of results) is a plain ``numpy.ndarray``."""

    def __array_wrap__(self, array, context=None, return_scalar=False):
        if return_scalar:
            return array[()]
        return np.asarray(array).view(np.ndarray)


def _snapshot_inplace(name, ufunc):
    """Augmented assignment: out of place on the read-only snapshot itself (the result goes to the
    attribute's setter, which uploads it), but TRUE in-place ndarray semantics on anything derived
    from it that the user owns -- ``c = m.basis.copy(); d = c; c += 1`` must change ``d`` too
    (round-3 advisor finding: copies, slices and pickles of a snapshot keep the subclass)."""
    base = getattr(np.ndarray, name)

    def op(self, other):
        if self.flags.writeable:
            return base(self, other)
        return ufunc(np.asarray(self).view(np.ndarray), other)
    return op


for _name, _ufunc in (("__iadd__", np.add), ("__isub__", np.subtract), ("__imul__", np.multiply),
                      ("__itruediv__", np.true_divide), ("__ipow__", np.power),
                      ("__ifloordiv__", np.floor_divide), ("__imatmul__", np.matmul)):
    setattr(HostSnapshot, _name, _snapshot_inplace(_name, _ufunc))


def _snapshot(host):
    view = host.view(HostSnapshot)
    view.flags.writeable = False
    return view


class Synced:
    """Descriptor: NumPy view of a device-resident state variable."""

    def __init__(self, dtype=None):
        self.dtype = dtype

    def __set_name__(self, owner, name):
        self.name = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return obj._state_get(self.name)

    def __set__(self, obj, value):
        obj._state_set_host(self.name, value, self.dtype)

    def __delete__(self, obj):
        obj._state().pop(self.name, None)


class DeviceStateMixin:
    """Book-keeping for Synced attributes (host cache <-> device buffer)."""

    _batched = False

    # -- input binding ------------------------------------------------------------------
    def _bind_input(self, input) -> None:
        """Keep a private copy of the mixture(s) and put them in HBM as (B, N, F, T) complex128.

        ``input`` is a NumPy array (the reference contract; copied, ssspy/bss/ilrma.py:840) or,
        as an extension, a complex128 tensor already resident on the HIP device (used as is:
        the kernels never write to it).
        """
        if input.ndim not in (3, 4):
            raise ValueError(
                "input must be (n_channels, n_bins, n_frames) or "
                "(n_mixtures, n_channels, n_bins, n_frames), got shape {}".format(
                    tuple(input.shape))
            )
        self._batched = input.ndim == 4
        if isinstance(input, torch.Tensor):
            if not input.is_cuda or input.dtype != torch.complex128:
                raise ValueError("a tensor input must be complex128 on the HIP device")
            self.input = input
            X4 = input if self._batched else input[None]
            self._X = X4.contiguous()
        else:
            # The private copy the reference makes (``self.input = input.copy()``) IS the HBM
            # buffer: a host copy as well cost 27 ms of a one-mixture configs[2] call (269 MB of
            # fresh pages) for an attribute nothing on the path reads.  ``self.input`` is formed
            # from the device copy on first access, in the dtype that came in.
            arr = np.asarray(input)
            X4 = arr if self._batched else arr[None]
            self._X = dv.to_device(X4, dtype=np.complex128)
            self.__dict__["_input_value"] = None
            self.__dict__["_input_dtype"] = arr.dtype
        self._static_cov = None

    @property
    def input(self):
        """The mixture(s) as given (ref: ssspy/bss/ilrma.py:840, a private copy).  After a call
        with a NumPy array this is a download of the private HBM copy, made once on first access."""
        value = self.__dict__.get("_input_value")
        if value is None and self.__dict__.get("_input_dtype") is not None:
            host = dv.to_host(self._X)
            host = host if self._batched else host[0]
            dtype = self.__dict__["_input_dtype"]
            if host.dtype != dtype:
                host = host.real.astype(dtype) if dtype.kind != "c" else host.astype(dtype)
            value = self.__dict__["_input_value"] = host
        return value

    @input.setter
    def input(self, value):
        self.__dict__["_input_value"] = value
        self.__dict__["_input_dtype"] = None

    def _has_input(self) -> bool:
        return (self.__dict__.get("_input_value") is not None
                or self.__dict__.get("_input_dtype") is not None)

    def call_on_device(self, input, n_iter: int = 100, initial_call: bool = True, **kwargs):
        """``__call__`` for callers that keep their spectrograms in HBM (extension over the
        reference): ``input`` is a complex128 tensor on the HIP device (or a NumPy array, uploaded as
        usual) and the separated spectrograms come back as a device tensor of the same shape -- no
        download, nothing synchronises.  Everything else (kwarg injection, loss list, callbacks) is
        ``__call__``'s.  Used by ``parallel.separate_pipelined`` to overlap the transfers of one
        sub-batch with the iterations of another."""
        self._return_device = True
        try:
            return self(input, n_iter=n_iter, initial_call=initial_call, **kwargs)
        finally:
            self._return_device = False

    def _lead(self):
        return (self._X.shape[0],) if self._batched else ()

    def _C(self):
        """Static covariance C_i = (1/T) sum_j x_ij x_ij^H, (B, F, N, N); once per call."""
        if self._static_cov is None:
            from .. import _ops

            B, N, F, T = self._X.shape
            self._static_cov = _ops.weighted_covariance(self._X).reshape(B, F, N, N)
        return self._static_cov

    def _state(self):
        return self.__dict__.setdefault("_dev_state", {})

    def _state_has(self, name):
        return name in self._state()

    def _state_get(self, name):
        st = self._state()
        if name not in st:
            raise AttributeError(
                "'{}' object has no attribute '{}'".format(type(self).__name__, name)
            )
        ent = st[name]
        if ent["none"]:
            return None
        self._state_settle(ent)
        if ent["host"] is None:
            self._check_device_errors()
            host = dv.to_host(ent["dev"])
            if not self._batched:
                host = host[0]
            # a snapshot of the device buffer, which stays the truth: writing into the snapshot
            # would be lost silently, so it is read-only -- state is changed by assigning the
            # attribute (``m.basis = new``), which uploads
            ent["host"], ent["host_rw"] = _snapshot(host), host
        return ent["host"]

    def _final_output(self):
        """What ``__call__`` returns: the host copy of ``output``.  The iteration is over, so the
        array is handed out writable like the reference's (which returns ``self.output`` itself)."""
        if getattr(self, "_return_device", False):  # call_on_device(): no download
            out_dev = self._state_dev("output")
            return out_dev if self._batched else out_dev[0]
        out = self.output
        ent = self._state().get("output", {})
        if ent.get("host") is out and ent.get("host_rw") is not None:
            return ent["host_rw"]  # the same memory as the read-only snapshot, writable
        return out

    def _state_set_host(self, name, value, dtype=None):
        if value is None:
            self._state()[name] = {"host": None, "dev": None, "none": True, "dtype": dtype,
                                   "rev": self._next_rev()}
        else:
            if isinstance(value, torch.Tensor):
                value = value.detach().cpu().numpy()
            self._state()[name] = {"host": value, "dev": None, "none": False, "dtype": dtype,
                                   "rev": self._next_rev()}

    def _state_set_dev(self, name, tensor):
        """Device buffer is now the truth (host cache dropped)."""
        st = self._state()
        dtype = st[name]["dtype"] if name in st else None
        st[name] = {"host": None, "dev": tensor, "none": False, "dtype": dtype,
                    "rev": self._next_rev()}

    def _state_touch(self, name):
        """A kernel rewrote the device buffer in place: drop the host cache."""
        ent = self._state()[name]
        ent["host"] = ent["host_rw"] = None
        ent["rev"] = self._next_rev()
        ent.pop("lazy", None)  # (whoever rewrote it read it through _state_dev, or replaced all of it)

    def _state_defer(self, name, fill):
        """The value changed but the device buffer was not rewritten: ``fill()`` brings it up to date
        and runs before the next read (host view or kernel operand), if there is one.  ILRMA's ISS /
        IPA iterations keep the filters their updates imply and read the mixture through them, so
        ``output`` is only formed when somebody looks at it."""
        ent = self._state()[name]
        ent["host"] = ent["host_rw"] = None
        ent["rev"] = self._next_rev()
        ent["lazy"] = fill

    @staticmethod
    def _state_settle(ent):
        fill = ent.pop("lazy", None)
        if fill is not None:
            fill()

    def _next_rev(self):
        self.__dict__["_state_serial"] = self.__dict__.get("_state_serial", 0) + 1
        return self.__dict__["_state_serial"]

    def _state_rev(self, name):
        """Serial number of the current contents of a state variable: changes whenever the value is
        replaced (host assignment, new device buffer) or rewritten in place (_state_touch).  Lets a
        by-product of one kernel (e.g. the frame powers the fused ISS sweep leaves behind) be reused
        by a later step only while the buffer it describes is still the same."""
        return self._state()[name].get("rev", 0)

    def _state_is_none(self, name):
        return self._state()[name]["none"]

    def _state_dev(self, name):
        """Device tensor of a state variable (uploading a fresh private copy if needed)."""
        ent = self._state()[name]
        if ent["none"]:
            return None
        self._state_settle(ent)
        if ent["dev"] is None:
            host = np.asarray(ent["host"])
            if not self._batched:
                host = host[None]
            np_dtype = None
            if ent["dtype"] is not None:
                np_dtype = np.complex128 if ent["dtype"] == dv.c128 else np.float64
            ent["dev"] = dv.to_device(host, dtype=np_dtype)
            # from here on the device copy is what the kernels see: hand out the host value
            # read-only so that an in-place edit cannot be lost silently (see _state_get)
            if isinstance(ent["host"], np.ndarray):
                ent["host"] = _snapshot(ent["host"])
        return ent["dev"]

    def _host_floor_state(self, name, floor) -> None:
        """``state = flooring_fn(state)`` on the host for a flooring callable the kernels cannot run
        (utils.flooring.DeviceFloor.host); one mixture at a time, so the callable sees the
        reference's shapes.  A no-op for the three built-in floors (applied inside the kernels)."""
        host = getattr(floor, "host", None)
        if host is None:
            return
        value = np.asarray(getattr(self, name))
        if self._batched:
            value = np.stack([np.asarray(host(v)) for v in value])
        else:
            value = np.asarray(host(value))
        setattr(self, name, value)

    # -- implied-filter route: its rounding bound, watched while it runs (round 6)
    # The ISS / ISS2 / IPA iterations that read the mixture through the filters their updates imply
    # form their statistics as W U W^H; ssspy_covariance_congruence_tracked leaves, per mixture, the
    # power-weighted mean square of kappa = (sum |w||u||w|) / (W U W^H)_rr of every launch in a pair
    # of device words: eps * kappa_rms estimates the relative error the product adds to the
    # spectrogram in that iteration, where the reference's sum over the samples adds eps.  Past the
    # limit the separator forms Y once and goes on with the reference's on-Y iteration.
    #
    # The look: behind every launch its words go to a page-locked mirror (asynchronous copy, event);
    # the NEXT iteration waits for that event before its own launch.  By then the device is past
    # the spot in a host-bound loop, and in a device-bound one it still has that iteration's source
    # model passes queued: no bubble either way, and the view is exactly one launch old.
    #
    # Calibration (benchmarks/implied_guard.py, tools/kappa_trace.py; profiles/r06_implied_guard.txt):
    # the route's distance from the oracle after 8-12 iterations is 0.05 .. 0.5 eps * max kappa_rms
    # where that exceeds the on-Y route's own 1e-13 .. 1e-9.  configs[1]: kappa_rms 1e3 .. 9e3 for
    # 60 iterations of ILRMA (then 1e5 .. 4e7 as the NMF variances of silent sources reach their
    # floor), 1e3 .. 6e3 for good with AuxIVA.  Degenerate draws (3 sources on 8 frames, 4 on 11):
    # 1e3 .. 1e4 for 5-7 iterations, then up to x400 per iteration to 1e12 (2e-6 of the oracle).
    # Random 4 x 4 mixing alone puts kappa_rms at 1e3 .. 2e4 for nine mixtures in ten and at 1.2e5
    # for one of the 128 of the bench batch (tools/kappa_batch.py), from the second iteration on and
    # for good -- with no measurable effect (6e-11 from the oracle after 8 iterations through the
    # filters against 3e-11 on Y).  The limit is 1e6: what the route adds per iteration stays below
    # 1e-10 of the spectrogram, and the launch that can slip through a one-launch-old view (x400)
    # adds 2e-9 .. 2e-8 once.  A batch leaves the route as a whole when one of its mixtures passes.
    # How often the host looks: after every launch while kappa moves (more than x1.5 between looks)
    # or is within two decades of the limit, else after every fourth -- the degenerate draws climb
    # by x1.1, 1.3, 2, 2, 2.5, 7, 14, 60, 300 per iteration, so the quiet phase ends well below it.
    _implied_amp_limit = 1.0e6
    _amp_every_launch = False  # (the calibration tools look after every launch)

    def _amp_reset(self) -> None:
        self.__dict__["_amp"] = None

    def _amp_slots(self):
        amp = self.__dict__.get("_amp")
        if amp is None:
            B = self._X.shape[0]
            host = torch.zeros((2, B, 2), dtype=dv.f64, pin_memory=True)
            amp = {"dev": dv.zeros((2, B, 2), dv.f64, self._X.device), "phase": 0, "worst": 0.0,
                   "host": host, "np": host.numpy(), "last": 0.0, "next_look": 1, "pending": None, "launches": 0,
                   "event": (torch.cuda.Event(), torch.cuda.Event())}
            self.__dict__["_amp"] = amp
        return amp

    def _amp_tracked(self, power):
        """The ``tracked`` argument of _ops.covariance_congruence for the next launch: None between
        two looks (the plain kernel: no power loads, no atomics)."""
        amp = self._amp_slots()
        if amp["launches"] + 1 < amp["next_look"]:
            return None
        return (power, amp["dev"], amp["phase"])

    def _amp_launched(self, tracked) -> None:
        """Behind a launch on the route; a tracked one sends its half of the ring to the mirror."""
        amp = self._amp_slots()
        amp["launches"] += 1
        if tracked is not None:
            half = amp["phase"] & 1
            amp["host"][half].copy_(amp["dev"][half], non_blocking=True)
            amp["event"][half].record()
            amp["pending"] = half
            amp["phase"] += 1

    def _amp_look(self, amp) -> None:
        half, amp["pending"] = amp["pending"], None
        amp["event"][half].synchronize()
        h = amp["np"][half]
        if h.shape[0] == 1:
            e2, p = float(h[0, 0]), float(h[0, 1])
            worst = (e2 / p) ** 0.5 if p > 0 and e2 >= 0 else (0.0 if p <= 0 else float("inf"))
        else:
            e2, p = h[:, 0], h[:, 1]
            with np.errstate(all="ignore"):
                k2 = (e2 / p)[p > 0]
            worst = 0.0 if not k2.size else (
                float("inf") if np.isnan(k2).any() else float(np.sqrt(k2.max())))
        if worst != worst:
            worst = float("inf")
        moving = worst > 1.5 * amp["last"] or worst * 100.0 > self._implied_amp_limit
        amp["next_look"] = amp["launches"] + (1 if moving or self._amp_every_launch else 4)
        amp["last"] = worst
        amp["worst"] = max(amp["worst"], worst)

    def _amp_kappa_rms(self) -> float:
        """Largest per-mixture kappa_rms among the launches looked at so far."""
        amp = self.__dict__.get("_amp")
        if amp is None:
            return 0.0
        if amp["pending"] is not None:
            self._amp_look(amp)
        return amp["worst"]

    def _amp_exceeded(self) -> bool:
        """Asked before an iteration on the route: did a launch looked at pass the limit?"""
        amp = self.__dict__.get("_amp")
        if amp is None:
            return False
        if amp["pending"] is not None:
            self._amp_look(amp)
        return not (amp["worst"] <= self._implied_amp_limit)

    def _implied_iterations(self) -> int:
        """Iterations the current call has run on the implied-filter route."""
        amp = self.__dict__.get("_amp")
        return 0 if amp is None else amp["launches"]

    # -- singular-matrix / non-convergence reporting
    def _info_tensor(self):
        """Device counters the kernels bump: [0] singular per-bin systems (the reference raises
        LinAlgError), [1] mixtures whose IPA Newton iteration had not converged (it warns)."""
        info = self.__dict__.get("_info")
        if info is None:
            info = dv.zeros((2,), dv.i32)
            self.__dict__["_info"] = info
        return info

    def _newton_counter(self):
        return self._info_tensor()[1:]

    def _newton_words(self, dev):
        """Vote words / arrival counters of the IPA sweeps (ssspy_ipa_sweep_newton_words), kept for
        the call: every sweep prepares them itself."""
        B, N = self._X.shape[0], self._X.shape[1]
        words = self.__dict__.get("_newton_ws")
        need = int(_lib.load().ssspy_ipa_sweep_newton_words(B, N))
        if words is None or words.numel() < need or words.device != dev:
            words = self.__dict__["_newton_ws"] = dv.empty((need,), dv.i64, dev)
        return words

    def _check_device_errors(self):
        from .. import _ops

        _ops.check_workspace_canaries()  # no-op unless SSSPY_AMD_WS_CANARY is set
        info = self.__dict__.get("_info")
        if info is not None:
            singular, not_converged = (int(v) for v in info.tolist())  # synchronises
            if self.__dict__.get("_newton_ws") is not None:
                timeouts = _lib.load().ssspy_debug_barrier_timeouts()
                if timeouts:
                    raise _lib.HipLibraryError(
                        "an IPA sweep gave up waiting for its mixture's workgroups {} time(s): the "
                        "results of this process are not to be trusted".format(timeouts))
            if singular or not_converged:
                info.zero_()
            if not_converged:
                import warnings

                warnings.warn(
                    "Newton-Raphson method did not converge in {} iterations.".format(
                        getattr(self, "newton_iter", "the given")), UserWarning)
            if singular:
                _lib.raise_if_singular(singular, type(self).__name__)
