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


def _default_eps(fn):
    try:
        param = inspect.signature(fn).parameters.get("eps")
    except (TypeError, ValueError):
        return EPS
    if param is None or param.default is inspect.Parameter.empty:
        return EPS
    return float(param.default)


def device_flooring(flooring_fn):
    """Map a flooring callable to the ``(kind, eps)`` pair the HIP kernels take.

    Recognised: ``None`` / ``identity`` -> NONE, ``max_flooring`` -> MAX, ``add_flooring`` ->
    ADD, each possibly wrapped in ``functools.partial(..., eps=...)``; functions are matched
    by name so the reference's own ``ssspy.special.flooring`` functions work too.  Any other
    callable cannot run inside a kernel: NotImplementedError (there is no CPU fallback).
    """
    if flooring_fn is None:
        return (_lib.FLOOR_NONE, 0.0)
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
            return (_lib.FLOOR_NONE, 0.0)
        if name == "max_flooring":
            return (_lib.FLOOR_MAX, _default_eps(fn) if eps is None else eps)
        if name == "add_flooring":
            return (_lib.FLOOR_ADD, _default_eps(fn) if eps is None else eps)
    raise NotImplementedError(
        "flooring_fn={!r} is not one of identity / max_flooring / add_flooring "
        "(optionally functools.partial(..., eps=...)); arbitrary Python callables cannot be "
        "evaluated inside the HIP kernels".format(flooring_fn)
    )
