"""Per-bin spatial update operators (NumPy in, NumPy out) running on the device.

The operator-level seam of the reference (ssspy/bss/_update_spatial_model.py): pure
functions on C-contiguous arrays.  Inputs may carry an extra leading batch axis.
"""

import functools
from typing import Callable, Optional

import numpy as np

from .. import _device as dv
from .. import _lib, _ops
from ..special.flooring import max_flooring
from ..utils.flooring import device_flooring
from ..utils.select_pair import resolve_pairs, sequential_pair_selector

EPS = 1e-10
_DEFAULT_FLOOR = functools.partial(max_flooring, eps=EPS)


def update_by_ip1(
    demix_filter: np.ndarray,
    weighted_covariance: np.ndarray,
    flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = _DEFAULT_FLOOR,
    overwrite: bool = True,
) -> np.ndarray:
    """Update demixing filters by iterative projection (ref: _update_spatial_model.py:17-78).

    Args:
        demix_filter: (n_bins, n_sources, n_channels) complex.
        weighted_covariance: (n_bins, n_sources, n_channels, n_channels) complex.
        flooring_fn: identity / max_flooring / add_flooring (optionally partial(..., eps=)).
        overwrite: write the result back into ``demix_filter`` (and return it), as the
            reference does.

    Raises:
        numpy.linalg.LinAlgError: if a bin's system ``W U_n`` is singular.
    """
    floor = device_flooring(flooring_fn, allow_host=True)  # _ops.update_by_ip1 floors on the host
    batched = demix_filter.ndim == 4
    W = dv.to_device(demix_filter if batched else demix_filter[None], dtype=np.complex128)
    U = dv.to_device(weighted_covariance if batched else weighted_covariance[None],
                     dtype=np.complex128)
    info = dv.zeros((1,), dv.i32)
    _ops.update_by_ip1(W, U, floor, info)
    _lib.raise_if_singular(int(info.item()), "update_by_ip1")
    out = dv.to_host(W)
    out = out if batched else out[0]
    if overwrite:
        demix_filter[...] = out
        return demix_filter
    return out


def update_by_iss1(
    separated: np.ndarray,
    weight: np.ndarray,
    flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = _DEFAULT_FLOOR,
) -> np.ndarray:
    """Update separated spectrograms by iterative source steering (ref: :146-194).

    Args:
        separated: (n_sources, n_bins, n_frames) complex.
        weight: (n_sources, n_bins, n_frames) or (n_sources, 1, n_frames) real.

    Returns:
        New array with the updated spectrograms.
    """
    floor = device_flooring(flooring_fn, allow_host=True)
    batched = separated.ndim == 4
    Y = dv.to_device(separated if batched else separated[None], dtype=np.complex128)
    wt = weight if batched else weight[None]
    B, N, F, T = Y.shape
    if wt.shape[2] == 1 and F != 1:
        w = dv.to_device(wt[:, :, 0, :], dtype=np.float64)
        kind = _lib.WEIGHT_FRAME
    else:
        w = dv.to_device(np.broadcast_to(wt, (B, N, F, T)), dtype=np.float64)
        kind = _lib.WEIGHT_BIN_FRAME
    if floor.host is not None:  # an arbitrary callable: the N x F denominators go to the host
        out = dv.to_host(_ops.update_by_iss1_host_floor(Y, w, kind, floor.host))
    elif T <= _ops.iss1_fused_max_frames(N):
        out = dv.to_host(_ops.iss1_fused(Y, w, kind, floor))  # Y is a private device copy
    else:
        Vc = _ops.weighted_covariance(Y, w, kind, N)
        G = _ops.iss1_transform(Vc, floor)
        out = dv.to_host(_ops.separate(Y, G))
    return out if batched else out[0]


def update_by_ip2(
    demix_filter: np.ndarray,
    weighted_covariance: np.ndarray,
    flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = _DEFAULT_FLOOR,
    pair_selector=None,
    overwrite: bool = True,
) -> np.ndarray:
    """Update demixing filters by pairwise iterative projection (ref: :81-143).

    The two rows of a pair come out with the arbitrary phase of a 2 x 2 eigenvector (as in the
    reference, where it is LAPACK's); scale restoration removes it.
    """
    floor = device_flooring(flooring_fn, what="update_by_ip2", allow_host=True)
    N = demix_filter.shape[-2]
    W = dv.to_device(demix_filter[None], dtype=np.complex128)
    U = dv.to_device(weighted_covariance[None], dtype=np.complex128)
    info = dv.zeros((1,), dv.i32)
    _ops.update_by_ip2(W, U, resolve_pairs(pair_selector, N), floor, info)
    _lib.raise_if_singular(int(info.item()), "update_by_ip2")
    out = dv.to_host(W)[0]
    if overwrite:
        demix_filter[...] = out
        return demix_filter
    return out


def update_by_iss2(
    separated: np.ndarray,
    weight: np.ndarray,
    flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = _DEFAULT_FLOOR,
    pair_selector=None,
) -> np.ndarray:
    """Update separated spectrograms by pairwise iterative source steering (ref: :197-314).

    Default pairs: (0,1), (2,3), ... as in the reference (sequential selector with step 2).
    """
    floor = device_flooring(flooring_fn, what="update_by_iss2", allow_host=True)
    Y = dv.to_device(separated[None], dtype=np.complex128)
    B, N, F, T = Y.shape
    if pair_selector is None:
        pair_selector = functools.partial(sequential_pair_selector, stop=N, step=2)
    wt = weight[None]
    if wt.shape[2] == 1 and F != 1:
        w = dv.to_device(wt[:, :, 0, :], dtype=np.float64)
        kind = _lib.WEIGHT_FRAME
    else:
        w = dv.to_device(np.broadcast_to(wt, (B, N, F, T)), dtype=np.float64)
        kind = _lib.WEIGHT_BIN_FRAME
    info = dv.zeros((1,), dv.i32)
    Vc = _ops.weighted_covariance(Y, w, kind, N)
    G = _ops.iss2_transform(Vc, resolve_pairs(pair_selector, N), floor, info)
    out = dv.to_host(_ops.separate(Y, G))[0]
    _lib.raise_if_singular(int(info.item()), "update_by_iss2")
    return out


def update_by_ipa(
    separated: np.ndarray,
    weight: np.ndarray,
    normalization: bool = True,
    flooring_fn: Optional[Callable[[np.ndarray], np.ndarray]] = _DEFAULT_FLOOR,
    max_iter: int = 1,
) -> np.ndarray:
    """Update separated spectrograms by iterative projection with adjustment (ref: :398-513).

    Args:
        separated: (n_sources, n_bins, n_frames) complex, n_sources in [2, 4].
        weight: (n_sources, n_bins, n_frames) or (n_sources, 1, n_frames) real.
        normalization: unit-trace normalisation of the LQPQM problem.
        max_iter: Newton steps of the LQPQM solver (every bin runs all of them).
    """
    floor = device_flooring(flooring_fn, what="update_by_ipa")
    Y = dv.to_device(separated[None], dtype=np.complex128)
    B, N, F, T = Y.shape
    wt = weight[None]
    if wt.shape[2] == 1 and F != 1:
        w = dv.to_device(wt[:, :, 0, :], dtype=np.float64)
        kind = _lib.WEIGHT_FRAME
    else:
        w = dv.to_device(np.broadcast_to(wt, (B, N, F, T)), dtype=np.float64)
        kind = _lib.WEIGHT_BIN_FRAME
    info = dv.zeros((2,), dv.i32)  # [singular systems, mixtures whose Newton loop did not converge]
    _ops.update_by_ipa(Y, w, kind, normalization, max_iter, floor, info, not_converged=info[1:])
    out = dv.to_host(Y)[0]
    singular, not_converged = (int(v) for v in info.tolist())
    if not_converged:
        import warnings

        warnings.warn("Newton-Raphson method did not converge in {} iterations.".format(max_iter),
                      UserWarning)
    _lib.raise_if_singular(singular, "update_by_ipa")
    return out
