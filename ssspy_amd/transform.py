"""STFT / ISTFT on the device, with the conventions of ``scipy.signal.stft`` / ``istft`` that the
reference's workflow uses either side of a separator (``window="hann"``, ``boundary="zeros"``,
``padded=True``, one-sided, ``scaling="spectrum"``; tests/package/bss/test_ilrma.py et al.).

    Z = stft(x, n_fft=2048, hop_length=512)          # (n_channels, n_fft // 2 + 1, n_frames)
    Y = GaussILRMA(n_basis=16)(Z)
    y = istft(Y, n_fft=2048, hop_length=512)[..., : x.shape[-1]]

Both take NumPy arrays or device tensors; with ``device_output=True`` the result stays in HBM, so
a separator can consume the spectrogram (``_bind_input`` accepts device tensors) and hand its
output to ``istft`` without the spectrogram crossing PCIe.  ``n_fft`` is a power of two <= 8192.
"""

from typing import Optional, Union

import numpy as np

from . import _device as dv
from . import _lib



def _L():
    return _lib.load()


def _window(window: Union[str, np.ndarray], n_fft: int) -> np.ndarray:
    if isinstance(window, str):
        if window not in ("hann", "hanning"):
            raise NotImplementedError("window {!r} (available: 'hann' or an array)".format(window))
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)  # periodic, as get_window
    w = np.asarray(window, dtype=np.float64)
    if w.shape != (n_fft,):
        raise ValueError("window must have n_fft = {} samples".format(n_fft))
    return w


def _stream():
    return dv.stream_handle()


def stft(x, n_fft: int, hop_length: Optional[int] = None, window="hann", device_output=False):
    """x (..., n_samples) real -> (..., n_fft // 2 + 1, n_frames) complex128."""
    hop = n_fft // 2 if hop_length is None else int(hop_length)
    w = _window(window, n_fft)
    xd = dv.to_device(x, dtype=np.float64) if isinstance(x, np.ndarray) else x
    lead, L = tuple(xd.shape[:-1]), int(xd.shape[-1])
    flat = xd.reshape(1, -1, L).contiguous()
    C = flat.shape[1]
    n_frames = int(_L().ssspy_stft_frames(L, n_fft, hop))
    Z = dv.empty((1, C, n_fft // 2 + 1, n_frames), dv.c128, flat.device)
    wd = dv.to_device(w, dtype=np.float64)
    _lib.check(
        _L().ssspy_stft(dv.ptr(flat), dv.ptr(Z), dv.ptr(wd), float(w.sum()), 1, C, L, n_fft,
                        hop, _stream()),
        "stft",
    )
    Z = Z.reshape(lead + (n_fft // 2 + 1, n_frames))
    return Z if device_output else dv.to_host(Z)


def istft(Z, n_fft: int, hop_length: Optional[int] = None, window="hann", device_output=False):
    """Z (..., n_fft // 2 + 1, n_frames) complex -> (..., n_samples) real (trim to the original
    length yourself, as with SciPy: the forward transform pads to a whole number of hops)."""
    hop = n_fft // 2 if hop_length is None else int(hop_length)
    w = _window(window, n_fft)
    Zd = dv.to_device(Z, dtype=np.complex128) if isinstance(Z, np.ndarray) else Z
    lead, F, n_frames = tuple(Zd.shape[:-2]), int(Zd.shape[-2]), int(Zd.shape[-1])
    if F != n_fft // 2 + 1:
        raise ValueError("expected {} bins for n_fft = {}, got {}".format(n_fft // 2 + 1, n_fft, F))
    flat = Zd.reshape(1, -1, F, n_frames).contiguous()
    C = flat.shape[1]
    L = int(_L().ssspy_istft_samples(n_frames, n_fft, hop))
    x = dv.empty((1, C, L), dv.f64, flat.device)
    seg = dv.empty((C * n_frames * n_fft,), dv.f64, flat.device)
    wd = dv.to_device(w, dtype=np.float64)
    _lib.check(
        _L().ssspy_istft(dv.ptr(flat), dv.ptr(x), dv.ptr(wd), float(w.sum()), dv.ptr(seg), 1, C,
                         n_frames, n_fft, hop, _stream()),
        "istft",
    )
    x = x.reshape(lead + (L,))
    return x if device_output else dv.to_host(x)
