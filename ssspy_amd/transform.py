"""STFT / ISTFT on the device, with the conventions of ``scipy.signal.stft`` / ``istft`` that the
reference's workflow uses either side of a separator (``window="hann"``, ``boundary="zeros"``,
``padded=True``, one-sided, ``scaling="spectrum"``; tests/package/bss/test_ilrma.py et al.).

    Z = stft(x, n_fft=2048, hop_length=512)          # (n_channels, n_fft // 2 + 1, n_frames)
    Y = GaussILRMA(n_basis=16)(Z)
    y = istft(Y, n_fft=2048, hop_length=512)[..., : x.shape[-1]]

Both take NumPy arrays or device tensors; with ``device_output=True`` the result stays in HBM, so
a separator can consume the spectrogram (``_bind_input`` accepts device tensors) and hand its
output to ``istft`` without the spectrogram crossing PCIe.  ``n_fft``: a power of two <= 65536 or any
length in [2, 32768] (Bluestein; up to 8192 points in LDS, longer transforms on a workspace in HBM);
``window``: an array, or what ``scipy.signal.get_window`` takes for
its periodic windows (a name, or a ``(name, parameter)`` tuple).
"""

import warnings
from typing import Optional, Union

import numpy as np

from . import _device as dv
from . import _lib



def _L():
    return _lib.load()


def _cosine_sum(a, n):
    k = 2.0 * np.pi * np.arange(n) / n  # periodic (fftbins=True): the (n + 1)-point symmetric window cut
    w = np.zeros(n)
    for i, c in enumerate(a):
        w += ((-1) ** i) * c * np.cos(i * k)
    return w


def get_window(window, n_fft: int) -> np.ndarray:
    """The periodic (``fftbins=True``) windows of ``scipy.signal.get_window`` that ``scipy.signal.stft``
    accepts by name: the (n_fft,) samples are a host-side table the kernels multiply by.
    ref: the reference's workflow passes ``window="hann"`` (tests/package/bss/test_ilrma.py:95-110)."""
    n = int(n_fft)
    args = ()
    if isinstance(window, tuple):
        window, args = window[0], tuple(window[1:])
    name = str(window).lower()
    m = n + 1  # the symmetric window of n + 1 points, last one dropped
    t = np.arange(n)
    if name in ("hann", "hanning", "han"):
        return _cosine_sum([0.5, 0.5], n)
    if name in ("hamming", "hamm", "ham"):
        return _cosine_sum([0.54, 0.46], n)
    if name in ("blackman", "black", "blk"):
        return _cosine_sum([0.42, 0.50, 0.08], n)
    if name in ("blackmanharris", "blackharr", "bkh"):
        return _cosine_sum([0.35875, 0.48829, 0.14128, 0.01168], n)
    if name in ("nuttall", "nutl", "nut"):
        return _cosine_sum([0.3635819, 0.4891775, 0.1365995, 0.0106411], n)
    if name in ("flattop", "flat", "flt"):
        return _cosine_sum([0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368], n)
    if name in ("general_hamming",):
        (alpha,) = args
        return _cosine_sum([alpha, 1.0 - alpha], n)
    if name in ("general_cosine",):
        (coeffs,) = args
        return _cosine_sum(list(coeffs), n)
    if name in ("boxcar", "box", "ones", "rect", "rectangular"):
        return np.ones(n)
    if name in ("bartlett", "bart", "brt"):
        return 1.0 - np.abs(2.0 * t / (m - 1) - 1.0)
    if name in ("triang", "triangle", "tri"):
        k = np.arange(1, (m + 1) // 2 + 1)
        if m % 2 == 0:
            half = (2 * k - 1.0) / m
            w = np.concatenate([half, half[::-1]])
        else:
            half = 2 * k / (m + 1.0)
            w = np.concatenate([half, half[-2::-1]])
        return w[:n]
    if name in ("cosine", "halfcosine"):
        return np.sin(np.pi / m * (t + 0.5))
    if name in ("bohman", "bman", "bmn"):
        x = np.abs(np.linspace(-1, 1, m)[1:-1])
        w = (1 - x) * np.cos(np.pi * x) + 1.0 / np.pi * np.sin(np.pi * x)
        return np.concatenate([[0.0], w, [0.0]])[:n]
    if name in ("parzen", "parz", "par"):
        k = np.arange(-(m - 1) / 2.0, (m - 1) / 2.0 + 0.5, 1.0)
        na = k[np.abs(k) <= (m - 1) / 4.0]
        nb = k[np.abs(k) > (m - 1) / 4.0]
        wa = 1 - 6 * (np.abs(na) / (m / 2.0)) ** 2 + 6 * (np.abs(na) / (m / 2.0)) ** 3
        wb = 2 * (1 - np.abs(nb) / (m / 2.0)) ** 3
        half = len(nb) // 2
        return np.concatenate([wb[:half], wa, wb[half:]])[:n]
    if name in ("kaiser", "ksr"):
        (beta,) = args
        k = np.arange(m)
        alpha = (m - 1) / 2.0
        return (np.i0(beta * np.sqrt(np.maximum(1 - ((k - alpha) / alpha) ** 2.0, 0.0)))
                / np.i0(beta))[:n]
    if name in ("gaussian", "gauss", "gss"):
        (std,) = args
        k = np.arange(m) - (m - 1.0) / 2.0
        return np.exp(-(k ** 2) / (2.0 * std * std))[:n]
    if name in ("tukey", "tuk"):
        alpha = args[0] if args else 0.5
        if alpha <= 0:
            return np.ones(n)
        if alpha >= 1.0:
            return _cosine_sum([0.5, 0.5], n)
        k = np.arange(m)
        width = int(np.floor(alpha * (m - 1) / 2.0))
        n1, n2, n3 = k[: width + 1], k[width + 1: m - width - 1], k[m - width - 1:]
        w1 = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * n1 / alpha / (m - 1))))
        w3 = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * n3 / alpha / (m - 1))))
        return np.concatenate([w1, np.ones(n2.shape), w3])[:n]
    if name in _SCIPY_ONLY_WINDOWS:
        raise NotImplementedError(
            "window {!r} is a scipy.signal window that is not built here; pass "
            "scipy.signal.get_window({!r}, n_fft) as an array instead.".format(name, window))
    raise ValueError("Unknown window type {!r}.".format(window))


# valid scipy.signal.get_window names that have no closed form here (see get_window)
_SCIPY_ONLY_WINDOWS = frozenset([
    "barthann", "brthan", "bth", "lanczos", "sinc", "taylor", "taylorwin", "exponential", "poisson",
    "chebwin", "cheb", "dpss", "general_gaussian", "general gaussian", "general gauss",
    "general_gauss", "ggs", "general_cosine", "general cosine", "general_hamming",
    "general hamming", "kaiser_bessel_derived", "kbd",
])


def _window(window, n_fft: int) -> np.ndarray:
    if isinstance(window, (str, tuple)):
        return np.ascontiguousarray(get_window(window, n_fft), dtype=np.float64)
    w = np.asarray(window, dtype=np.float64)
    if w.shape != (n_fft,):
        raise ValueError("window must have n_fft = {} samples".format(n_fft))
    return w


def _workspace(n_fft, dev):
    nbytes = int(_L().ssspy_stft_workspace_bytes(int(n_fft)))
    return dv.empty((max(nbytes, 16) // 16,), dv.c128, dev) if nbytes else None


def _stream():
    return dv.stream_handle()


def stft(x, n_fft: int, hop_length: Optional[int] = None, window="hann", device_output=False):
    """x (..., n_samples) real -> (..., n_fft // 2 + 1, n_frames) complex128."""
    hop = n_fft // 2 if hop_length is None else int(hop_length)
    xd = dv.to_device(x, dtype=np.float64) if isinstance(x, np.ndarray) else x
    lead, L = tuple(xd.shape[:-1]), int(xd.shape[-1])
    if n_fft > L:
        # a signal shorter than one window: scipy.signal.stft shrinks nperseg to the signal (with this
        # warning) and keeps noverlap, so the transform has L // 2 + 1 bins
        # (scipy/signal/_spectral_py.py, _triage_segments)
        warnings.warn("nperseg = {0:d} is greater than input length  = {1:d}, using nperseg = {1:d}"
                      .format(n_fft, L))
        noverlap = n_fft - hop
        if noverlap >= L:
            raise ValueError("noverlap must be less than nperseg.")
        n_fft, hop = L, L - noverlap
    w = _window(window, n_fft)
    flat = xd.reshape(1, -1, L).contiguous()
    C = flat.shape[1]
    n_frames = int(_L().ssspy_stft_frames(L, n_fft, hop))
    Z = dv.empty((1, C, n_fft // 2 + 1, n_frames), dv.c128, flat.device)
    wd = dv.to_device(w, dtype=np.float64)
    ws = _workspace(n_fft, flat.device)
    _lib.check(
        _L().ssspy_stft(dv.ptr(flat), dv.ptr(Z), dv.ptr(wd), float(w.sum()), 1, C, L, n_fft,
                        hop, dv.ptr(ws), _stream()),
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
    ws = _workspace(n_fft, flat.device)
    _lib.check(
        _L().ssspy_istft(dv.ptr(flat), dv.ptr(x), dv.ptr(wd), float(w.sum()), dv.ptr(seg), 1, C,
                         n_frames, n_fft, hop, dv.ptr(ws), _stream()),
        "istft",
    )
    x = x.reshape(lead + (L,))
    return x if device_output else dv.to_host(x)
