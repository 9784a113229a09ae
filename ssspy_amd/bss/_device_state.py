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
        if ent["host"] is None:
            self._check_device_errors()
            host = dv.to_host(ent["dev"])
            if not self._batched:
                host = host[0]
            ent["host"] = host
        return ent["host"]

    def _state_set_host(self, name, value, dtype=None):
        if value is None:
            self._state()[name] = {"host": None, "dev": None, "none": True, "dtype": dtype}
        else:
            if isinstance(value, torch.Tensor):
                value = value.detach().cpu().numpy()
            self._state()[name] = {"host": value, "dev": None, "none": False, "dtype": dtype}

    def _state_set_dev(self, name, tensor):
        """Device buffer is now the truth (host cache dropped)."""
        st = self._state()
        dtype = st[name]["dtype"] if name in st else None
        st[name] = {"host": None, "dev": tensor, "none": False, "dtype": dtype}

    def _state_touch(self, name):
        """A kernel rewrote the device buffer in place: drop the host cache."""
        self._state()[name]["host"] = None

    def _state_is_none(self, name):
        return self._state()[name]["none"]

    def _state_dev(self, name):
        """Device tensor of a state variable (uploading a fresh private copy if needed)."""
        ent = self._state()[name]
        if ent["none"]:
            return None
        if ent["dev"] is None:
            host = np.asarray(ent["host"])
            if not self._batched:
                host = host[None]
            np_dtype = None
            if ent["dtype"] is not None:
                np_dtype = np.complex128 if ent["dtype"] == dv.c128 else np.float64
            ent["dev"] = dv.to_device(host, dtype=np_dtype)
        return ent["dev"]

    # -- singular-matrix reporting ------------------------------------------------------
    def _info_tensor(self):
        info = self.__dict__.get("_info")
        if info is None:
            info = dv.zeros((1,), dv.i32)
            self.__dict__["_info"] = info
        return info

    def _check_device_errors(self):
        info = self.__dict__.get("_info")
        if info is not None:
            count = int(info.item())  # synchronises
            if count:
                info.zero_()
                _lib.raise_if_singular(count, type(self).__name__)
