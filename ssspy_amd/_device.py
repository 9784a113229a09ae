"""Device-memory plumbing (PyTorch-ROCm tensors as HBM buffers, HIP streams).

PyTorch is used only as an allocator / stream provider / host<->device copier; every
arithmetic kernel on the hot path is in libssspy_amd.so.
"""

import threading
import weakref

import numpy as np
import torch


def device(index=None):
    if not torch.cuda.is_available():
        raise RuntimeError(
            "ssspy_amd needs a HIP device (torch.cuda.is_available() is False); "
            "the demixing hot path has no CPU fallback"
        )
    if index is None:
        index = torch.cuda.current_device()
    return torch.device("cuda", index)


# A copy from pageable memory is synchronous AND stream-ordered: the host waits for everything
# queued on the stream before it.  A runner that keeps several sub-batches in flight
# (parallel.separate_pipelined) would stall in the next separator's _reset -- in front of its random
# initialisation, 52 ms per 64 mixtures with the device idle -- so inside `staged_uploads()` the
# arrays go through page-locked staging buffers (torch's caching host allocator keeps a buffer
# alive until the copy that reads it has run) and the host never waits.  Off by default: for one
# call the extra host copy costs more than the wait it saves.
_staged = threading.local()  # (per thread: a runner's context must not change other threads' copies)
_STAGE_MIN_BYTES = 1 << 16


class staged_uploads:
    """Context manager: ``to_device`` copies of 64 KiB and more become asynchronous."""

    def __enter__(self):
        _staged.depth = getattr(_staged, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _staged.depth -= 1
        return False


def to_device(array, dtype=None, dev=None):
    """Copy a NumPy array into a fresh contiguous HBM buffer (never aliases the input)."""
    a = np.ascontiguousarray(array, dtype=dtype)
    if not a.flags.writeable:  # e.g. a view of an .npz member or a broadcast: torch wants writable
        a = a.copy()
    t = torch.from_numpy(a)
    if getattr(_staged, "depth", 0) and a.nbytes >= _STAGE_MIN_BYTES:
        pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        pinned.copy_(t)
        return pinned.to(dev or device(), non_blocking=True)
    return t.to(dev or device(), copy=True)


# A download into pageable memory is staged by the driver through its own bounce buffers: 7.4 GB/s
# for the 33.6 MB result of a configs[1] call (4.5 of its 15 ms).  Between 1 and 512 MB the copy
# goes into a page-locked block instead (25 GB/s) and the NumPy array handed out IS that block:
# torch's caching host allocator takes it back when the array dies and hands it to the next call,
# so only the first call pays for the page-locking.
# Page-locked memory cannot be swapped and the allocator rounds a block up to a power of two, so a
# caller who keeps every result (``outs = [m(X) for X in dataset]``) would pin up to twice the data
# (round-5 advisor finding).  The blocks alive in callers' hands are counted (rounded size, released
# by a finalizer when the array dies) and capped at _PINNED_OUTSTANDING_CAP; past the cap -- i.e.
# when results are being hoarded rather than consumed -- downloads are ordinary pageable arrays.
# (round 6: the upper bound went from 256 to 512 MB -- the 269 MB result of a one-mixture configs[2]
#  call took 17-120 ms into fresh pageable memory, the page faults of a new mapping, not the copy)
_PINNED_DOWNLOAD_BYTES = (1 << 20, 1 << 29)
_PINNED_OUTSTANDING_CAP = 1 << 30
_pinned_lock = threading.Lock()
_pinned_outstanding = [0]


def _pinned_release(nbytes):
    with _pinned_lock:
        _pinned_outstanding[0] -= nbytes


def pinned_outstanding_bytes():
    """Page-locked bytes (as the host allocator rounds them) held by arrays ``to_host`` handed out."""
    return _pinned_outstanding[0]


def to_host(tensor):
    t = tensor.detach()
    nbytes = t.numel() * t.element_size()
    if t.is_cuda and _PINNED_DOWNLOAD_BYTES[0] <= nbytes <= _PINNED_DOWNLOAD_BYTES[1]:
        rounded = 1 << (nbytes - 1).bit_length()
        with _pinned_lock:
            fits = _pinned_outstanding[0] + rounded <= _PINNED_OUTSTANDING_CAP
            if fits:
                _pinned_outstanding[0] += rounded
        if not fits:
            return t.cpu().numpy()
        try:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        except RuntimeError:  # (no page-locked memory left: the plain copy still works)
            _pinned_release(rounded)
            return t.cpu().numpy()
        host.copy_(t)
        out = host.numpy()
        # (views and snapshots of `out` keep it alive through .base; the block goes back to torch's
        #  host allocator when the last of them dies)
        weakref.finalize(out, _pinned_release, rounded)
        return out
    return t.cpu().numpy()


def empty(shape, dtype, dev=None):
    return torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=dev or device())


def zeros(shape, dtype, dev=None):
    return torch.zeros(tuple(int(s) for s in shape), dtype=dtype, device=dev or device())


def eye_filters(B, F, N, dev=None):
    """(B, F, N, N) identity matrices (plumbing: a torch fill, no arithmetic of the path)."""
    W = zeros((B, F, N, N), c128, dev)
    W.diagonal(dim1=-2, dim2=-1).fill_(1.0)
    return W


def ptr(tensor):
    if tensor is None:
        return None
    assert tensor.is_contiguous(), "device buffers must be C-contiguous"
    return tensor.data_ptr()


# torch.cuda.current_stream() builds a Stream object per call (~7 us): two C-ABI calls per iteration
# made that the largest host cost of a single-mixture AuxIVA-ISS iteration (25 us of issue time
# against 28 us in total).  The raw-handle query torch's own compiler backends use costs ~0.3 us.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle():
    """hipStream_t of torch's current stream as an integer for ctypes."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


c128 = torch.complex128
f64 = torch.float64
i32 = torch.int32
i64 = torch.int64
