"""Thin Python wrappers over the C ABI: torch HBM tensors in, torch HBM tensors out.

Shapes carry the leading batch axis B (see include/ssspy_amd.h).  Nothing here
synchronises with the device.
"""

from . import _device as dv
from . import _lib, _routes
from ._device import ptr


def _L():
    return _lib.load()


# ---- workspace allocation.  With SSSPY_AMD_WS_CANARY=1 (the GPU tests set it) every workspace gets
# a guard band behind it, filled with a pattern that check_workspace_canaries() verifies: a kernel
# that writes past the size the C ABI asked for is caught instead of corrupting a neighbour.
import os as _os

_CANARY_DOUBLES = 4096 if _os.environ.get("SSSPY_AMD_WS_CANARY") else 0
_CANARY_VALUE = -7.25e300
_canaries = []


def _workspace(nbytes, dev):
    n = (int(nbytes) + 7) // 8
    buf = dv.empty((n + _CANARY_DOUBLES,), dv.f64, dev)
    if not _CANARY_DOUBLES:
        return buf, int(nbytes)
    import weakref

    buf[n:].fill_(_CANARY_VALUE)
    view = buf[:n]
    view._guard = buf[n:]  # keeps the band addressable for as long as the workspace lives
    _canaries.append((weakref.ref(view), n))
    return view, int(nbytes)


def check_workspace_canaries():
    """Raise if anything wrote behind a live workspace (only active with SSSPY_AMD_WS_CANARY=1)."""
    alive = []
    for ref, n in _canaries:
        view = ref()
        if view is None:
            continue
        alive.append((ref, n))
        if not bool((view._guard == _CANARY_VALUE).all().item()):
            raise RuntimeError("a kernel wrote past the end of a {}-byte workspace".format(8 * n))
    _canaries[:] = alive


def _st():
    return dv.stream_handle()


# ---- transient scratch of single operators (partial sums folded inside the same C-ABI call): one
# growing buffer per (device, stream): calls on one stream run in order, so the buffer is never
# needed by two operators at once; separators driven on different streams of one device (threads,
# parallel.separate_pipelined callers) get their own (round-3 advisor finding)
_scratch_bufs = {}


def _scratch(nbytes, dev):
    nbytes = int(nbytes)
    if nbytes == 0:
        return None, 0
    key = (str(dev), _st())
    ent = _scratch_bufs.get(key)
    if ent is None or ent[1] < nbytes:
        ent = _workspace(nbytes, dev)
        _scratch_bufs[key] = ent
    return ent


def separate(X, W, out=None):
    B, N, F, T = X.shape
    if out is None:
        out = dv.empty((B, N, F, T), dv.c128, X.device)
    _lib.check(_L().ssspy_separate(ptr(X), ptr(W), ptr(out), B, N, F, T, _st()), "separate")
    return out


def weighted_covariance(A, weight=None, kind=_lib.WEIGHT_UNIT, n_sets=1, out=None):
    B, N, F, T = A.shape
    if out is None:
        out = dv.empty((B, F, n_sets, N, N), dv.c128, A.device)
    _lib.check(
        _L().ssspy_weighted_covariance(ptr(A), ptr(weight), kind, ptr(out), B, N, n_sets, F, T, _st()),
        "weighted_covariance",
    )
    return out


def covariance_congruence(C, G, out, tracked=None):
    """out = G C G^H per bin: C, out (B, F, N, N) or (B, F, S, N, N) with the S matrices of a bin
    sharing its G (B, F, N, N).  ``tracked`` = (power (B, F, N, N), slots (2, B, 2) f64, phase): the
    launch also leaves its power-weighted rounding amplification per mixture in slots[phase & 1]
    and clears the other half (2..4 sources, see the header)."""
    B, F, N = C.shape[0], C.shape[1], C.shape[-1]
    S = C.shape[2] if C.dim() == 5 else 1
    if tracked is not None:
        power, slots, phase = tracked
        assert tuple(slots.shape) == (2, B, 2) and tuple(power.shape) == (B, F, N, N)
        _lib.check(_L().ssspy_covariance_congruence_tracked(ptr(C), ptr(G), ptr(out), B, F, S, N,
                                                            ptr(power), ptr(slots), int(phase),
                                                            _st()),
                   "covariance_congruence")
        return out
    _lib.check(_L().ssspy_covariance_congruence_sets(ptr(C), ptr(G), ptr(out), B, F, S, N, _st()),
               "covariance_congruence")
    return out


def compose_filters(G, W, out):
    """out = G W per bin, (B, F, N, N)."""
    B, F, N = W.shape[0], W.shape[1], W.shape[-1]
    _lib.check(_L().ssspy_compose_filters(ptr(G), ptr(W), ptr(out), B, F, N, _st()),
               "compose_filters")
    return out


def cross_covariance(A, Bm, out=None):
    B, N, F, T = A.shape
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, A.device)
    _lib.check(_L().ssspy_cross_covariance(ptr(A), ptr(Bm), ptr(out), B, N, F, T, _st()),
               "cross_covariance")
    return out


def _apply_host(fn, dev_tensor):
    """fn on the host copy of a device tensor, one mixture (leading axis) at a time: the shapes the
    callable sees are the reference's."""
    import numpy as np

    host = dv.to_host(dev_tensor)
    out = np.stack([np.asarray(fn(h), dtype=host.dtype) for h in host])
    if out.shape != host.shape:
        raise ValueError("flooring_fn must return an array of the shape it was given")
    return out


def update_by_ip1(W, U, flooring, info=None):
    B, F, N, _ = W.shape
    host = getattr(flooring, "host", None)
    if host is not None:
        # an arbitrary flooring callable: one source at a time, the denominators d (B, F) come down,
        # are floored on the host and go back up (ssspy/bss/_update_spatial_model.py:63-76)
        denom = dv.empty((B, F), dv.f64, W.device)
        for n in range(N):
            _lib.check(_L().ssspy_ip1_source_solve(ptr(W), ptr(U), ptr(denom), n, B, F, N,
                                                   ptr(info), _st()), "ip1_source_solve")
            denom.copy_(dv.to_device(_apply_host(host, denom), dev=W.device))
            _lib.check(_L().ssspy_scale_filter_row(ptr(W), ptr(denom), n, B, F, N, _st()),
                       "scale_filter_row")
        return W
    _lib.check(
        _L().ssspy_update_by_ip1(ptr(W), ptr(U), B, F, N, flooring[0], flooring[1], ptr(info), _st()),
        "update_by_ip1",
    )
    return W


def update_by_ip1_logdet_slots(B, F, N):
    """Shares per mixture ``update_by_ip1_logdet`` leaves in ``logdet``."""
    return int(_L().ssspy_update_by_ip1_logdet_slots(B, F, N))


def update_by_ip1_logdet(W, U, flooring, info, logdet, logdet_stride):
    """update_by_ip1 that also leaves sum_i log|det W_i| of the filters as they come in, as shares at
    logdet[s * logdet_stride + b] (device floors only; the caller zeroes the array once per run)."""
    B, F, N, _ = W.shape
    _lib.check(
        _L().ssspy_update_by_ip1_logdet(ptr(W), ptr(U), B, F, N, flooring[0], flooring[1], ptr(info),
                                        ptr(logdet), int(logdet_stride), _st()),
        "update_by_ip1_logdet",
    )
    return W


def update_by_iss1_host_floor(Y, weight, kind, host):
    """ISS1 with an arbitrary flooring callable: per source one covariance pass and one pass that
    applies the steering step; the N x F denominators are floored on the host in between.
    ref: ssspy/bss/_update_spatial_model.py:178-192."""
    import numpy as np

    B, N, F, T = Y.shape
    Vc = None
    eye = np.eye(N, dtype=np.complex128)
    for n in range(N):
        Vc = weighted_covariance(Y, weight, kind, N, out=Vc)  # (B, F, m, a, b) = mean varphi_m y y^H
        V = dv.to_host(Vc)
        num = V[:, :, np.arange(N), np.arange(N), n]          # (B, F, m): mean varphi_m y_m conj(y_n)
        den = np.real(V[:, :, :, n, n])                       # (B, F, m): mean varphi_m |y_n|^2
        # the callable sees (n_sources, n_bins) as in the reference
        den = np.stack([np.asarray(host(d.T), dtype=np.float64).T for d in den])
        v = num / den
        v[:, :, n] = 1.0 - 1.0 / np.sqrt(den[:, :, n])
        G = np.broadcast_to(eye, (B, F, N, N)).copy()
        G[:, :, :, n] -= v                                     # y <- y - v y_n
        separate(Y, dv.to_device(G, dev=Y.device), out=Y)
    return Y


def iss1_transform(Vc, flooring, out=None):
    B, F, N = Vc.shape[0], Vc.shape[1], Vc.shape[-1]
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, Vc.device)
    _lib.check(
        _L().ssspy_iss1_transform(ptr(Vc), ptr(out), B, F, N, flooring[0], flooring[1], _st()),
        "iss1_transform",
    )
    return out


def _pair_array(pairs):
    import ctypes

    flat = [int(v) for pair in pairs for v in pair]
    return (ctypes.c_int * len(flat))(*flat), len(flat) // 2


def update_by_ip2(W, U, pairs, flooring, info=None, pair_only=False):
    """Pairwise iterative projection in place on W; U (B,F,N,N,N), or with pair_only the pair's own
    two covariances (B,F,2,N,N)."""
    B, F, N, _ = W.shape
    host = getattr(flooring, "host", None)
    if host is not None:
        # an arbitrary flooring callable: pair by pair, the two denominators (B, F) floored on the
        # host between the projection and the division (ref: _update_spatial_model.py:381-388)
        import numpy as np

        denom = dv.empty((B, F, 2), dv.f64, W.device)
        for pair in pairs:
            arr, _ = _pair_array([pair])
            _lib.check(_L().ssspy_update_by_ip2_deferred(ptr(W), ptr(U), int(bool(pair_only)), arr,
                                                         B, F, N, ptr(denom), ptr(info), _st()),
                       "update_by_ip2 (deferred)")
            d = dv.to_host(denom)
            for k, row in enumerate(pair):
                # (the callable sees (n_bins, 1), as in the reference)
                dk = np.stack([np.asarray(host(d[b, :, k][:, None]), dtype=np.float64).reshape(F)
                               for b in range(B)])
                _lib.check(_L().ssspy_scale_filter_row(ptr(W), ptr(dv.to_device(dk, dev=W.device)),
                                                       int(row), B, F, N, _st()), "scale_filter_row")
        return W
    arr, n_pairs = _pair_array(pairs)
    _lib.check(
        _L().ssspy_update_by_ip2(ptr(W), ptr(U), int(bool(pair_only)), arr, n_pairs, B, F, N,
                                 flooring[0], flooring[1], ptr(info), _st()),
        "update_by_ip2",
    )
    return W


def iss2_transform(Vc, pairs, flooring, info=None, out=None):
    B, F, N = Vc.shape[0], Vc.shape[1], Vc.shape[-1]
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, Vc.device)
    host = getattr(flooring, "host", None)
    if host is not None:
        # an arbitrary flooring callable: pair by pair on the accumulated transform, the two
        # denominators floored on the host (ref: _update_spatial_model.py:300-312)
        import numpy as np

        denom = dv.empty((B, F, 2), dv.f64, Vc.device)
        for p, pair in enumerate(pairs):
            arr, _ = _pair_array([pair])
            _lib.check(_L().ssspy_iss2_transform_deferred(ptr(Vc), ptr(out), arr, int(p > 0), B, F,
                                                          N, ptr(denom), ptr(info), _st()),
                       "iss2_transform (deferred)")
            d = dv.to_host(denom)
            # (the callable sees both members at once, (2, n_bins, 1), as in the reference:
            #  _update_spatial_model.py:300-312 floors the square root with keepdims)
            fl = np.stack([np.asarray(host(d[b].T[:, :, None]), dtype=np.float64).reshape(2, F)
                           for b in range(B)])
            for k, row in enumerate(pair):
                dk = np.ascontiguousarray(fl[:, k, :])
                _lib.check(_L().ssspy_scale_filter_row(ptr(out), ptr(dv.to_device(dk, dev=Vc.device)),
                                                       int(row), B, F, N, _st()), "scale_filter_row")
        return out
    arr, n_pairs = _pair_array(pairs)
    _lib.check(
        _L().ssspy_iss2_transform(ptr(Vc), ptr(out), arr, n_pairs, B, F, N, flooring[0],
                                  flooring[1], ptr(info), _st()),
        "iss2_transform",
    )
    return out


def ipa_sweep(Vc, normalization, max_iter, flooring, info=None, out=None, newton_ws=None,
              not_converged=None):
    """All N source steps of an IPA sweep on the per-bin statistics Vc (overwritten: V_m <- G V_m G^H
    after each step); returns the accumulated transform G (B, F, N, N)."""
    B, F, N = Vc.shape[0], Vc.shape[1], Vc.shape[-1]
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, Vc.device)
    _lib.check(
        _L().ssspy_ipa_sweep(ptr(Vc), ptr(out), B, F, N, int(bool(normalization)), int(max_iter),
                             flooring[0], flooring[1], ptr(info), ptr(newton_ws),
                             ptr(not_converged), _st()),
        "ipa_sweep",
    )
    return out


def update_by_ipa(Y, weight, kind, normalization, max_iter, flooring, info=None, not_converged=None,
                  Vc=None, frame_power=False):
    """One IPA sweep in place on the device spectrogram Y (B, N, F, T).  The weights are fixed over
    the sweep (ref: _update_spatial_model.py:436-445), so the covariances the reference recomputes
    from the updated spectrogram before every source step are G V G^H of the previous ones: one
    weighted covariance, the N steps on the per-bin statistics, one Y <- G Y (round 5; three passes
    over Y instead of 3 N).
    Vc: the weighted covariances of Y when the caller has formed them already (then `weight` is not
    read; overwritten)."""
    B, N = Y.shape[0], Y.shape[1]
    newton_ws = dv.empty((int(_L().ssspy_ipa_sweep_newton_words(B, N)),), dv.i64, Y.device)
    if Vc is None:
        Vc = weighted_covariance(Y, weight, kind, N)
    G = ipa_sweep(Vc, normalization, max_iter, flooring, info, newton_ws=newton_ws,
                  not_converged=not_converged)
    if frame_power:  # (AuxIVA: the next iteration's weights want sum_i |y|^2 of the new Y)
        r2 = separate_frame_power(Y, G)
        if r2 is not None:
            return r2
    separate(Y, G, out=Y)
    return None if frame_power else Y


def iss1_fused_max_frames(n_sources):
    return int(_L().ssspy_iss1_fused_max_frames(n_sources))


def iss1_fused(Y, weight, kind, flooring, r2_next=None, logdet=None):
    """In-place fused ISS1 on Y; optionally leaves the next iteration's frame powers (summed without
    atomics: blocks' partial sums in a scratch buffer, folded in order), and moves `logdet` (B,) =
    sum_i log|det W_i| of the implied demixing filters along with the sweeps."""
    B, N, F, T = Y.shape
    ws, ws_bytes = (None, 0)
    if r2_next is not None or logdet is not None:
        ws, ws_bytes = _scratch(_L().ssspy_iss1_fused_workspace_bytes(B, N, F, T), Y.device)
    if logdet is not None:
        _lib.check(
            _L().ssspy_iss1_fused_tracked(ptr(Y), ptr(weight), kind, ptr(r2_next), B, N, F, T,
                                          flooring[0], flooring[1], ptr(logdet), ptr(ws), ws_bytes,
                                          _st()),
            "iss1_fused_tracked",
        )
        return Y
    _lib.check(
        _L().ssspy_iss1_fused(ptr(Y), ptr(weight), kind, ptr(r2_next), B, N, F, T, flooring[0],
                              flooring[1], ptr(ws), ws_bytes, _st()),
        "iss1_fused",
    )
    return Y


def projection_back_filter(W, reference_id, info=None, scale_out=None):
    B, F, N, _ = W.shape
    _lib.check(
        _L().ssspy_projection_back_filter(ptr(W), ptr(scale_out), B, F, N, reference_id, ptr(info),
                                          _st()),
        "projection_back_filter",
    )
    return W


def mdp_scale(YX, YY, reference_id, out=None):
    B, F, N, _ = YX.shape
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, YX.device)
    _lib.check(_L().ssspy_mdp_scale(ptr(YX), ptr(YY), ptr(out), B, F, N, int(reference_id), _st()),
               "mdp_scale")
    return out


def ilrma_scale_basis(basis, G, domain):
    B, N, F, K = basis.shape
    _lib.check(_L().ssspy_ilrma_scale_basis(ptr(basis), ptr(G), B, N, F, K, domain, _st()),
               "ilrma_scale_basis")


def projection_back_scale(XY, YY, reference_id, info=None, out=None):
    B, F, N, _ = XY.shape
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, XY.device)
    _lib.check(
        _L().ssspy_projection_back_scale(ptr(XY), ptr(YY), ptr(out), B, F, N, reference_id,
                                         ptr(info), _st()),
        "projection_back_scale",
    )
    return out


def demix_from_covariance(YX, XX, info=None, out=None):
    B, F, N, _ = YX.shape
    if out is None:
        out = dv.empty((B, F, N, N), dv.c128, YX.device)
    _lib.check(
        _L().ssspy_demix_from_covariance(ptr(YX), ptr(XX), ptr(out), B, F, N, ptr(info), _st()),
        "demix_from_covariance",
    )
    return out


def sum_logdet(W, out=None):
    B, F, N, _ = W.shape
    if out is None:
        out = dv.empty((B,), dv.f64, W.device)
    _lib.check(_L().ssspy_sum_logdet(ptr(W), ptr(out), B, F, N, _st()), "sum_logdet")
    return out


# ----------------------------------------------------------------------------- ILRMA
def ilrma_workspace(B, N, F, T, K, dev):
    return _workspace(_L().ssspy_ilrma_workspace_bytes(B, N, F, T, K), dev)


GAUSS = (_lib.SOURCE_GAUSS, 0.0)  # source model as (SSSPY_SOURCE_*, model_param); see include/ssspy_amd.h


def ilrma_update_basis(X, W, basis, activation, domain, flooring, ws, ws_bytes, model=GAUSS):
    B, N, F, T = X.shape
    K = basis.shape[-1]
    _lib.check(
        _L().ssspy_ilrma_update_basis(ptr(X), ptr(W), ptr(basis), ptr(activation), B, N, F, T, K,
                                      domain, model[0], model[1], flooring[0], flooring[1], ptr(ws),
                                      ws_bytes, _st()),
        "ilrma_update_basis",
    )


def ilrma_update_activation(X, W, basis, activation, domain, flooring, ws, ws_bytes, model=GAUSS):
    B, N, F, T = X.shape
    K = basis.shape[-1]
    _lib.check(
        _L().ssspy_ilrma_update_activation(ptr(X), ptr(W), ptr(basis), ptr(activation), B, N, F, T,
                                           K, domain, model[0], model[1], flooring[0], flooring[1],
                                           ptr(ws), ws_bytes, _st()),
        "ilrma_update_activation",
    )


def ilrma_weighted_covariance(X, basis, activation, domain, ws, ws_bytes, out=None, W=None,
                              model=GAUSS, flooring=(0, 0.0)):
    B, N, F, T = X.shape
    K = basis.shape[-1]
    if out is None:
        out = dv.empty((B, F, N, N, N), dv.c128, X.device)
    _lib.check(
        _L().ssspy_ilrma_weighted_covariance(ptr(X), ptr(W), ptr(basis), ptr(activation), ptr(out),
                                             B, N, F, T, K, domain, model[0], model[1], flooring[0],
                                             flooring[1], ptr(ws), ws_bytes, _st()),
        "ilrma_weighted_covariance",
    )
    return out


def ilrma_normalize_filter(W, C, basis, domain, flooring, ws, ws_bytes):
    B, F, N, _ = W.shape
    K = basis.shape[-1]
    _lib.check(
        _L().ssspy_ilrma_normalize_filter(ptr(W), ptr(C), ptr(basis), B, N, F, K, domain,
                                          flooring[0], flooring[1], ptr(ws), ws_bytes, _st()),
        "ilrma_normalize_filter",
    )


def ilrma_normalize_output(Y, basis, domain, flooring, ws, ws_bytes, frame_power=None, logdet=None):
    """frame_power (B,N,T): sum over bins of |Y|^2 of the current Y when the caller already has it.
    logdet (B,): the tracked sum_i log|det W_i| of the ISS state, moved along with the scaling."""
    B, N, F, T = Y.shape
    K = basis.shape[-1]
    if logdet is not None:
        _lib.check(
            _L().ssspy_ilrma_normalize_output_tracked(
                ptr(Y), ptr(basis), ptr(frame_power), B, N, F, T, K, domain, flooring[0],
                flooring[1], ptr(ws), ws_bytes, ptr(logdet), _st()),
            "ilrma_normalize_output_tracked",
        )
        return
    _lib.check(
        _L().ssspy_ilrma_normalize_output(ptr(Y), ptr(basis), ptr(frame_power), B, N, F, T, K,
                                          domain, flooring[0], flooring[1], ptr(ws), ws_bytes,
                                          _st()),
        "ilrma_normalize_output",
    )


def ilrma_iss_weight(basis, activation, domain, out=None, Y=None, model=GAUSS, flooring=(0, 0.0),
                     Ypow=None):
    """Ypow: |y|^2 (B, N, F, T) float64 handed in instead of y (ssspy_ilrma_iss_weight_power)."""
    B, N, F, K = basis.shape
    T = activation.shape[-1]
    if out is None:
        out = dv.empty((B, N, F, T), dv.f64, basis.device)
    if Ypow is not None:
        _lib.check(
            _L().ssspy_ilrma_iss_weight_power(ptr(Ypow), ptr(basis), ptr(activation), ptr(out), B, N,
                                              F, T, K, domain, model[0], model[1], flooring[0],
                                              flooring[1], _st()),
            "ilrma_iss_weight_power",
        )
        return out
    _lib.check(
        _L().ssspy_ilrma_iss_weight(ptr(Y), ptr(basis), ptr(activation), ptr(out), B, N, F, T, K,
                                    domain, model[0], model[1], flooring[0], flooring[1], _st()),
        "ilrma_iss_weight",
    )
    return out


def ilrma_loss_data(X, W, basis, activation, domain, out=None, model=GAUSS):
    B, N, F, T = X.shape
    K = basis.shape[-1]
    if out is None:
        out = dv.empty((B,), dv.f64, X.device)
    ws, ws_bytes = _scratch(_L().ssspy_ilrma_loss_workspace_bytes(B, N, F, T, K, int(W is not None)),
                            X.device)
    _lib.check(
        _L().ssspy_ilrma_loss_data(ptr(X), ptr(W), ptr(basis), ptr(activation), ptr(out), B, N, F,
                                   T, K, domain, model[0], model[1], ptr(ws), ws_bytes, _st()),
        "ilrma_loss_data",
    )
    return out


def ilrma_ip1_update(X, C, W, basis, activation, U, domain, normalize, flooring, ws, ws_bytes,
                     info, model=GAUSS):
    B, N, F, T = X.shape
    K = basis.shape[-1]
    _lib.check(
        _L().ssspy_ilrma_ip1_update(ptr(X), ptr(C), ptr(W), ptr(basis), ptr(activation), ptr(U), B,
                                    N, F, T, K, domain, model[0], model[1], int(bool(normalize)),
                                    flooring[0], flooring[1], ptr(ws), ws_bytes, ptr(info), _st()),
        "ilrma_ip1_update",
    )


def ilrma_deferred_loss_supported(N, F, T, K, domain, model=GAUSS):
    return bool(_L().ssspy_ilrma_deferred_loss_supported(N, F, T, K, domain, model[0]))


def ilrma_ip1_update_deferred_loss(X, C, W, basis, activation, U, domain, normalize, flooring, ws,
                                   ws_bytes, info, loss_data, logdet, model=GAUSS):
    """The fused update_once() that also leaves the loss of the state at entry (data term, log-dets);
    returns False without touching the state when the shape has no such by-product."""
    B, N, F, T = X.shape
    K = basis.shape[-1]
    rc = _L().ssspy_ilrma_ip1_update_deferred_loss(
        ptr(X), ptr(C), ptr(W), ptr(basis), ptr(activation), ptr(U), B, N, F, T, K, domain, model[0],
        model[1], int(bool(normalize)), flooring[0], flooring[1], ptr(ws), ws_bytes, ptr(info),
        ptr(loss_data), ptr(logdet), _st())
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc, "ilrma_ip1_update_deferred_loss")
    return True


def ilrma_deferred_logdet_slots(B, N, F, T, K, domain, model=GAUSS):
    """Shares per mixture ``ilrma_ip1_update_loss_slots`` leaves in ``logdet`` (0: no by-product)."""
    return int(_L().ssspy_ilrma_deferred_logdet_slots(B, N, F, T, K, domain, model[0]))


def ilrma_deferred_loss_slots(B, N, F, T, K, domain, model=GAUSS):
    """Raw loss slots per mixture of ilrma_ip1_update_loss_slots (0: no by-product for this shape)."""
    return int(_L().ssspy_ilrma_deferred_loss_slots(B, N, F, T, K, domain, model[0]))


def ilrma_ip1_update_loss_slots(X, C, W, basis, activation, U, domain, normalize, flooring, ws,
                                ws_bytes, info, slots, slot_stride, logdet, model=GAUSS):
    """ilrma_ip1_update_deferred_loss with the data term left as raw slots (slot s of mixture b at
    slots[s * slot_stride + b]); the caller zeroes the array once and folds it once
    (fold_scalar_slots) for a whole run."""
    B, N, F, T = X.shape
    K = basis.shape[-1]
    _lib.check(_L().ssspy_ilrma_ip1_update_loss_slots(
        ptr(X), ptr(C), ptr(W), ptr(basis), ptr(activation), ptr(U), B, N, F, T, K, domain, model[0],
        model[1], int(bool(normalize)), flooring[0], flooring[1], ptr(ws), ws_bytes, ptr(info),
        ptr(slots), int(slot_stride), ptr(logdet), _st()), "ilrma_ip1_update_loss_slots")


def fold_scalar_slots(slots, total, nslots, out):
    """out[e] = sum_s slots[s * total + e] in slot order."""
    ws, ws_bytes = _scratch(_L().ssspy_fold_scalar_slots_workspace_bytes(int(total), int(nslots)),
                            slots.device)
    _lib.check(_L().ssspy_fold_scalar_slots(ptr(slots), int(total), int(nslots), ptr(out), ptr(ws),
                                            ws_bytes, _st()), "fold_scalar_slots")
    return out


def ilrma_partition_expand(basis, activation, latent, Teff, Vrep):
    B, N, F, K = Teff.shape
    T = Vrep.shape[-1]
    _lib.check(
        _L().ssspy_ilrma_partition_expand(ptr(basis), ptr(activation), ptr(latent), ptr(Teff),
                                          ptr(Vrep), B, N, F, T, K, _st()),
        "ilrma_partition_expand",
    )


def ilrma_partition_update(X, W, basis, activation, latent, Teff, Vrep, domain, steps, flooring, ws,
                           ws_bytes, model=GAUSS):
    B, N, F, T = X.shape
    K = basis.shape[-1]
    _lib.check(
        _L().ssspy_ilrma_partition_update(ptr(X), ptr(W), ptr(basis), ptr(activation), ptr(latent),
                                          ptr(Teff), ptr(Vrep), B, N, F, T, K, domain, model[0],
                                          model[1], steps, flooring[0], flooring[1], ptr(ws),
                                          ws_bytes, _st()),
        "ilrma_partition_update",
    )


def ilrma_partition_normalize(W, C, Y, basis, latent, domain, flooring, ws, ws_bytes):
    B, N, K = latent.shape
    F = basis.shape[-2]
    T = Y.shape[-1] if Y is not None else 1
    _lib.check(
        _L().ssspy_ilrma_partition_normalize(ptr(W), ptr(C), ptr(Y), ptr(basis), ptr(latent), B, N,
                                             F, T, K, domain, flooring[0], flooring[1], ptr(ws),
                                             ws_bytes, _st()),
        "ilrma_partition_normalize",
    )


# ------------------------------------------------------------------------------- IVA
def iva_frame_power(X, W, out=None):
    B, N, F, T = X.shape
    if out is None:
        out = dv.empty((B, N, T), dv.f64, X.device)
    ws, ws_bytes = _scratch(_L().ssspy_iva_frame_power_workspace_bytes(B, N, F, T), X.device)
    _lib.check(_L().ssspy_iva_frame_power(ptr(X), ptr(W), ptr(out), B, N, F, T, ptr(ws), ws_bytes,
                                          _st()), "iva_frame_power")
    return out


def separate_frame_power(Y, G, r2=None):
    """Y <- G Y in place and the frame powers sum_i |y|^2 (B, N, T) of the result in one walk; None
    when the shape has no such kernel (more than 8 sources: the caller makes the two passes)."""
    B, N, F, T = Y.shape
    if N > _lib.MAX_SOURCES:
        return None
    if r2 is None:
        r2 = dv.empty((B, N, T), dv.f64, Y.device)
    ws, ws_bytes = _scratch(_L().ssspy_iva_frame_power_workspace_bytes(B, N, F, T), Y.device)
    _lib.check(_L().ssspy_separate_frame_power(ptr(Y), ptr(G), ptr(Y), ptr(r2), B, N, F, T, ptr(ws),
                                               ws_bytes, _st()), "separate_frame_power")
    return r2


def iva_weight(r2, n_bins, contrast, flooring, weight=None, variance=None):
    B, N, T = r2.shape
    if weight is None:
        weight = dv.empty((B, N, T), dv.f64, r2.device)
    _lib.check(
        _L().ssspy_iva_weight(ptr(r2), ptr(weight), ptr(variance), B, N, n_bins, T, contrast,
                              flooring[0], flooring[1], _st()),
        "iva_weight",
    )
    return weight


def iva_loss_data(r2, variance, n_bins, contrast, out=None):
    B, N, T = r2.shape
    if out is None:
        out = dv.empty((B,), dv.f64, r2.device)
    _lib.check(
        _L().ssspy_iva_loss_data(ptr(r2), ptr(variance), ptr(out), B, N, n_bins, T, contrast, _st()),
        "iva_loss_data",
    )
    return out


# ----------------------------------------------------------------------------- FastMNMF
def fastmnmf_workspace(B, N, M, F, T, K, dev):
    return _workspace(_L().ssspy_fastmnmf_workspace_bytes(B, N, M, F, T, K), dev)


def fastmnmf_update(X, C, Q, D, basis, activation, steps, flooring, ws, ws_bytes, info):
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    _lib.check(
        _L().ssspy_fastmnmf_update(ptr(X), ptr(C), ptr(Q), ptr(D), ptr(basis), ptr(activation), B,
                                   N, M, F, T, K, steps, flooring[0], flooring[1], ptr(ws),
                                   ws_bytes, ptr(info), _st()),
        "fastmnmf_update",
    )


def fastmnmf_handover(B, N, M, F, T, K, dev):
    """The |Q x|^2 hand-over buffer of ssspy_fastmnmf_update_handover, or None for shapes without it."""
    n = _L().ssspy_fastmnmf_handover_doubles(B, N, M, F, T, K)
    if not n or not _routes.get("handover"):
        return None
    return _workspace(8 * n, dev)[0]


def fastmnmf_update_handover(X, C, Q, D, basis, activation, steps, flooring, ws, ws_bytes, info,
                             handover, valid):
    """`valid`: whether `handover` matches (Q, X); returns whether it does afterwards."""
    import ctypes

    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    flag = ctypes.c_int(1 if valid else 0)
    _lib.check(
        _L().ssspy_fastmnmf_update_handover(
            ptr(X), ptr(C), ptr(Q), ptr(D), ptr(basis), ptr(activation), B, N, M, F, T, K, steps,
            flooring[0], flooring[1], ptr(ws), ws_bytes, ptr(info), ptr(handover),
            ctypes.cast(ctypes.pointer(flag), ctypes.c_void_p), _st()),
        "fastmnmf_update_handover",
    )
    return bool(flag.value)


def fastmnmf_deferred_logdet_slots(B, N, M, F, T, K):
    """Shares per mixture ``fastmnmf_update_logdet`` leaves in ``logdet``."""
    return int(_L().ssspy_fastmnmf_deferred_logdet_slots(B, N, M, F, T, K))


def fastmnmf_update_logdet(X, C, Q, D, basis, activation, steps, flooring, ws, ws_bytes, info,
                           handover, valid, logdet, logdet_stride):
    """fastmnmf_update(_handover) that also leaves sum_i log|det Q_i| of the diagonalisers as they
    come in, as shares at logdet[s * logdet_stride + b] (the caller zeroes the array once per run
    and folds it once).  handover None: no hand-over buffer.  Returns the buffer's validity."""
    import ctypes

    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    flag = ctypes.c_int(1 if valid else 0)
    _lib.check(
        _L().ssspy_fastmnmf_update_handover_logdet(
            ptr(X), ptr(C), ptr(Q), ptr(D), ptr(basis), ptr(activation), B, N, M, F, T, K, steps,
            flooring[0], flooring[1], ptr(ws), ws_bytes, ptr(info), ptr(handover),
            ctypes.cast(ctypes.pointer(flag), ctypes.c_void_p) if handover is not None else None,
            ptr(logdet), int(logdet_stride), _st()),
        "fastmnmf_update_handover_logdet",
    )
    return bool(flag.value) if handover is not None else False


def fastmnmf_diagonalizer_covariance(X, D, basis, activation, out=None, ws=None, ws_bytes=0):
    """ws (the separator's fastmnmf_workspace): the tuned covariance pass instead of the generic one."""
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B, F, M, M, M), dv.c128, X.device)
    _lib.check(
        _L().ssspy_fastmnmf_diagonalizer_covariance(ptr(X), ptr(D), ptr(basis), ptr(activation),
                                                    ptr(out), B, N, M, F, T, K, ptr(ws),
                                                    int(ws_bytes), _st()),
        "fastmnmf_diagonalizer_covariance",
    )
    return out


def fastmnmf_weights(X, Q, D, basis, activation, out=None):
    """1 / R~ per (channel, bin, frame), (B, M, F, T): the weights of the diagonaliser covariance."""
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B, M, F, T), dv.f64, X.device)
    _lib.check(
        _L().ssspy_fastmnmf_weights(ptr(X), ptr(Q), ptr(D), ptr(basis), ptr(activation), ptr(out), B,
                                    N, M, F, T, K, _st()),
        "fastmnmf_weights",
    )
    return out


def fastmnmf_loss_data(X, Q, D, basis, activation, out=None):
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B,), dv.f64, X.device)
    ws, ws_bytes = _scratch(_L().ssspy_fastmnmf_loss_workspace_bytes(B, N, M, F, T), X.device)
    _lib.check(
        _L().ssspy_fastmnmf_loss_data(ptr(X), ptr(Q), ptr(D), ptr(basis), ptr(activation), ptr(out),
                                      B, N, M, F, T, K, ptr(ws), ws_bytes, _st()),
        "fastmnmf_loss_data",
    )
    return out


def fastmnmf_loss_data_handover(D, basis, activation, handover, n_channels, n_frames, out=None):
    """Data term of the loss from a valid |Q x|^2 hand-over (see fastmnmf_update_handover)."""
    B, N, F, K = basis.shape
    if out is None:
        out = dv.empty((B,), dv.f64, basis.device)
    ws, ws_bytes = _scratch(
        _L().ssspy_fastmnmf_loss_workspace_bytes(B, N, n_channels, F, n_frames), basis.device)
    _lib.check(
        _L().ssspy_fastmnmf_loss_data_handover(ptr(D), ptr(basis), ptr(activation), ptr(handover),
                                               ptr(out), B, N, n_channels, F, n_frames, K, ptr(ws),
                                               ws_bytes, _st()),
        "fastmnmf_loss_data_handover",
    )
    return out


def fastmnmf_loss_handover_slots(B, N, M, F, T, K):
    """Shares per mixture ``fastmnmf_loss_data_handover_slots`` writes (0: no hand-over)."""
    return int(_L().ssspy_fastmnmf_loss_handover_slots(B, N, M, F, T, K))


def fastmnmf_loss_data_handover_slots(D, basis, activation, handover, n_channels, n_frames, slots,
                                      slot_stride):
    """fastmnmf_loss_data_handover with the per-wave shares left raw (share s of mixture b at
    slots[s * slot_stride + b]); the caller zeroes the array once per run and folds it once."""
    B, N, F, K = basis.shape
    _lib.check(
        _L().ssspy_fastmnmf_loss_data_handover_slots(
            ptr(D), ptr(basis), ptr(activation), ptr(handover), ptr(slots), int(slot_stride), B, N,
            n_channels, F, n_frames, K, _st()),
        "fastmnmf_loss_data_handover_slots",
    )


def fastmnmf_separate(X, Q, D, basis, activation, reference_id, flooring, ws, ws_bytes, info,
                      out=None):
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B, N, F, T), dv.c128, X.device)
    _lib.check(
        _L().ssspy_fastmnmf_separate(ptr(X), ptr(Q), ptr(D), ptr(basis), ptr(activation), ptr(out),
                                     B, N, M, F, T, K, reference_id, flooring[0], flooring[1],
                                     ptr(ws), ws_bytes, ptr(info), _st()),
        "fastmnmf_separate",
    )
    return out


_HOST_FLOOR_EIG_BYTES = 1 << 28  # eigenvector scratch of fastmnmf_separate_host_floor


def fastmnmf_separate_host_floor(X, Q, D, basis, activation, reference_id, host_fn, ws, ws_bytes,
                                 info, out=None):
    """The Wiener filter with an arbitrary flooring callable on the eigenvalues of R_ij: stage 1
    leaves ascending eigenvalues (B, F, T, M) and eigenvectors in HBM, the callable runs on the host
    on each mixture's (F, T, M) array -- what to_psd hands it in the reference (special/psd.py:54-62)
    -- and stage 2 finishes from the floored values."""
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B, N, F, T), dv.c128, X.device)
    # a few mixtures at a time: the eigenvectors of every point are (F, T, M, M) complex128 per
    # mixture (34 MB at 4 channels of 513 x 256, 1 GB for 32 of them), the buffers are reused
    per = F * T * M * M * 16
    cb = max(1, min(B, _HOST_FLOOR_EIG_BYTES // per))
    lam = dv.empty((cb, F, T, M), dv.f64, X.device)
    P = dv.empty((cb, F, T, M, M), dv.c128, X.device)
    for b0 in range(0, B, cb):
        n = min(cb, B - b0)
        sl = slice(b0, b0 + n)
        for stage in (1, 2):
            _lib.check(
                _L().ssspy_fastmnmf_separate_eig(ptr(X[sl]), ptr(Q[sl]), ptr(D[sl]), ptr(basis[sl]),
                                                 ptr(activation[sl]), ptr(out[sl]), n, N, M, F, T, K,
                                                 reference_id, stage, ptr(lam), ptr(P), ptr(ws),
                                                 ws_bytes, ptr(info), _st()),
                "fastmnmf_separate_eig",
            )
            if stage == 1:
                lam[:n].copy_(dv.to_device(_apply_host(host_fn, lam[:n]), dev=X.device))
    return out


# ------------------------------------------------------------------------- GaussMNMF
def gmnmf_workspace(B, N, M, F, T, K, dev):
    return _workspace(_L().ssspy_gmnmf_workspace_bytes(B, N, M, F, T, K), dev)


def gmnmf_update(X, basis, activation, spatial, steps, flooring, ws, ws_bytes, latent=None):
    B, M, F, T = X.shape
    N, K = spatial.shape[1], basis.shape[-1]
    _lib.check(
        _L().ssspy_gmnmf_update(ptr(X), ptr(basis), ptr(activation), ptr(latent), ptr(spatial), B,
                                N, M, F, T, K, steps, flooring[0], flooring[1], ptr(ws), ws_bytes,
                                _st()),
        "gmnmf_update",
    )


def gmnmf_loss(X, basis, activation, spatial, flooring, out=None):
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B,), dv.f64, X.device)
    ws, ws_bytes = _scratch(_L().ssspy_gmnmf_loss_workspace_bytes(B, F, T), X.device)
    _lib.check(
        _L().ssspy_gmnmf_loss(ptr(X), ptr(basis), ptr(activation), ptr(spatial), ptr(out), B, N, M,
                              F, T, K, flooring[0], flooring[1], ptr(ws), ws_bytes, _st()),
        "gmnmf_loss",
    )
    return out


def gmnmf_separate(X, basis, activation, spatial, reference_id, flooring, out=None):
    B, M, F, T = X.shape
    N, K = basis.shape[1], basis.shape[-1]
    if out is None:
        out = dv.empty((B, N, F, T), dv.c128, X.device)
    _lib.check(
        _L().ssspy_gmnmf_separate(ptr(X), ptr(basis), ptr(activation), ptr(spatial), ptr(out), B,
                                  N, M, F, T, K, int(reference_id), flooring[0], flooring[1],
                                  _st()),
        "gmnmf_separate",
    )
    return out
