"""Selection of the flooring function and its translation to the device vocabulary."""

import functools
import inspect

from .. import _lib
from ..special.flooring import EPS, identity


def choose_flooring_fn(flooring_fn="self", method=None):
    """Resolve ``"self"`` / ``None`` / callable exactly as the reference does.

    ref: ssspy/utils/flooring.py:8-24.
    """
    if flooring_fn is None:
        assert method is None, "method is given, but flooring function is not specified."
        flooring_fn = identity
    elif type(flooring_fn) is str and flooring_fn == "self":
        if method is None or not hasattr(method, "flooring_fn"):
            flooring_fn = identity
        else:
            flooring_fn = method.flooring_fn
    assert callable(flooring_fn), "flooring_fn should be callable."
    return flooring_fn


class DeviceFloor(tuple):
    """``(kind, eps)`` as the kernels take it, plus ``host``: the flooring callable itself when it
    is none of the reference's three and therefore cannot run inside a kernel.  The kernels are
    then launched without a floor (``FLOOR_NONE``) and the callable is applied on the host to the
    small arrays the reference applies it to -- basis, activation, per-bin denominators, the
    normalisation scales -- between the device passes (ssspy/bss/ilrma.py:1126, :1202, :420;
    _update_spatial_model.py:74, :188; iva.py:1788)."""

    def __new__(cls, kind, eps, host=None):
        obj = tuple.__new__(cls, (kind, eps))
        obj.host = host
        return obj

    def __reduce__(self):  # copy.deepcopy / pickle of a separator that keeps its resolved floor
        return (DeviceFloor, (self[0], self[1], self.host))


def _default_eps(fn):
    try:
        param = inspect.signature(fn).parameters.get("eps")
    except (TypeError, ValueError):
        return EPS
    if param is None or param.default is inspect.Parameter.empty:
        return EPS
    return float(param.default)


def device_flooring(flooring_fn, allow_host=False, what="this operation"):
    """Map a flooring callable to the ``(kind, eps)`` pair the HIP kernels take.

    Recognised: ``None`` / ``identity`` -> NONE, ``max_flooring`` -> MAX, ``add_flooring`` ->
    ADD, each possibly wrapped in ``functools.partial(..., eps=...)``; functions are matched
    by name so the reference's own ``ssspy.special.flooring`` functions work too.  Any other
    callable: with ``allow_host=True`` -- the caller has a host-evaluation path and reads
    ``.host`` -- it comes back as ``DeviceFloor(NONE, 0, host=callable)`` and is evaluated on the
    host on the small arrays it acts on (see DeviceFloor) while the passes over the spectrograms
    stay on the device; otherwise ``NotImplementedError`` (never a silently unfloored run).
    """
    if flooring_fn is None:
        return DeviceFloor(_lib.FLOOR_NONE, 0.0)
    fn, eps = flooring_fn, None
    while isinstance(fn, functools.partial):
        if fn.args:
            break
        if "eps" in fn.keywords and eps is None:
            eps = float(fn.keywords["eps"])
        extra = set(fn.keywords) - {"eps"}
        if extra:
            break
        fn = fn.func
    name = getattr(fn, "__name__", None)
    if not isinstance(fn, functools.partial):
        if name == "identity":
            return DeviceFloor(_lib.FLOOR_NONE, 0.0)
        if name == "max_flooring":
            return DeviceFloor(_lib.FLOOR_MAX, _default_eps(fn) if eps is None else eps)
        if name == "add_flooring":
            return DeviceFloor(_lib.FLOOR_ADD, _default_eps(fn) if eps is None else eps)
    floor = DeviceFloor(_lib.FLOOR_NONE, 0.0, host=flooring_fn)
    if not allow_host:
        require_device_floor(floor, what)
    return floor


def host_floor(floor):
    """The host callable of a resolved floor, or None when the kernels apply it themselves."""
    return getattr(floor, "host", None)


def require_device_floor(floor, what):
    """For the steps whose floors sit in the middle of a per-element computation on the
    spectrograms (no small array to take to the host)."""
    if host_floor(floor) is not None:
        raise NotImplementedError(
            "{} is built for identity / max_flooring / add_flooring only (the floor acts inside a "
            "pass over the spectrograms); got flooring_fn={!r}".format(what, floor.host))
