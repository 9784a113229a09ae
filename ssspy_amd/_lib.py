"""ctypes binding of libssspy_amd.so (the C ABI declared in include/ssspy_amd.h).

The product path has no CPU fallback: if the shared library is missing or an entry point
fails, an exception is raised.
"""

import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# SSSPY_AMD_LIB: another build of the same library (A / B runs of kernel experiments)
LIB_PATH = os.environ.get("SSSPY_AMD_LIB") or os.path.join(_PKG, "lib", "libssspy_amd.so")

OK, ERR_BADARG, ERR_HIP, ERR_UNSUPPORTED, ERR_INTERNAL = 0, 1, 2, 3, 4
FLOOR_NONE, FLOOR_MAX, FLOOR_ADD = 0, 1, 2
WEIGHT_UNIT, WEIGHT_FRAME, WEIGHT_BIN_FRAME = 0, 1, 2
SOURCE_GAUSS, SOURCE_T, SOURCE_GGD = 0, 1, 2
PARTITION_LATENT, PARTITION_BASIS, PARTITION_ACTIVATION = 1, 2, 4
SOURCE_ME = 0x100  # OR-ed into the model: source_algorithm="ME"
CONTRAST_LAPLACE, CONTRAST_GAUSS, CONTRAST_GAUSS_FIXED = 0, 1, 2
MAX_PAIRS = 32
# SSSPY_MAX_SOURCES (per-N kernels: IPA, both MNMF classes, the Hermitian operators),
# SSSPY_RT_MAX_SOURCES (run-time-N kernels: the shared operators, ILRMA and AuxIVA), SSSPY_MAX_BASIS
MAX_SOURCES, RT_MAX_SOURCES, MAX_BASIS = 8, 16, 65536
ABI_VERSION = 3  # SSSPY_ABI_VERSION of the include/ssspy_amd.h these prototypes mirror

_p, _i, _d, _z = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
_q = ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/ssspy_amd.h one to one
PROTOTYPES = {
    "ssspy_amd_version": (ctypes.c_char_p, []),
    "ssspy_abi_version": (_i, []),
    "ssspy_last_error": (ctypes.c_char_p, []),
    "ssspy_separate": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "ssspy_weighted_covariance": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "ssspy_cross_covariance": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "ssspy_update_by_ip1": (_i, [_p, _p, _i, _i, _i, _i, _d, _p, _p]),
    "ssspy_update_by_ip1_logdet_slots": (_i, [_i, _i, _i]),
    "ssspy_update_by_ip1_logdet": (_i, [_p, _p, _i, _i, _i, _i, _d, _p, _p, _q, _p]),
    "ssspy_ip1_source_solve": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "ssspy_scale_filter_row": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "ssspy_iss1_transform": (_i, [_p, _p, _i, _i, _i, _i, _d, _p]),
    "ssspy_update_by_ip2": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _d, _p, _p]),
    "ssspy_covariance_congruence": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "ssspy_covariance_congruence_sets": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "ssspy_covariance_congruence_tracked": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _i, _p]),
    "ssspy_compose_filters": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "ssspy_ipa_sweep": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _d, _p, _p, _p, _p]),
    "ssspy_ipa_sweep_newton_words": (_z, [_i, _i]),
    "ssspy_debug_barrier_timeouts": (_i, []),
    "ssspy_iss2_transform": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _p]),
    "ssspy_update_by_ip2_deferred": (_i, [_p, _p, _i, _p, _i, _i, _i, _p, _p, _p]),
    "ssspy_iss2_transform_deferred": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "ssspy_iss1_fused_max_frames": (_i, [_i]),
    "ssspy_iss1_fused_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "ssspy_iss1_fused": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _d, _p, _z, _p]),
    "ssspy_iss1_fused_tracked": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _d, _p, _p, _z, _p]),
    "ssspy_projection_back_filter": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "ssspy_mdp_scale": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "ssspy_ilrma_scale_basis": (_i, [_p, _p, _i, _i, _i, _i, _d, _p]),
    "ssspy_projection_back_scale": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "ssspy_demix_from_covariance": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "ssspy_sum_logdet": (_i, [_p, _p, _i, _i, _i, _p]),
    "ssspy_solve": (_i, [_p, _p, _p, _q, _i, _i, _p, _p]),
    "ssspy_inv2": (_i, [_p, _p, _q, _p]),
    "ssspy_eigh": (_i, [_p, _p, _p, _q, _i, _p]),
    "ssspy_to_psd": (_i, [_p, _p, _q, _i, _i, _d, _p]),
    "ssspy_herm_rebuild": (_i, [_p, _p, _p, _q, _i, _i, _p]),
    "ssspy_eigh2": (_i, [_p, _p, _p, _p, _q, _i, _p, _p]),
    "ssspy_ilrma_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "ssspy_ilrma_update_basis": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _i, _d, _p, _z,
                                      _p]),
    "ssspy_ilrma_update_activation": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _i, _d, _p,
                                           _z, _p]),
    "ssspy_ilrma_weighted_covariance": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _i,
                                             _d, _p, _z, _p]),
    "ssspy_ilrma_normalize_filter": (_i, [_p, _p, _p, _i, _i, _i, _i, _d, _i, _d, _p, _z, _p]),
    "ssspy_ilrma_normalize_output": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _p, _z, _p]),
    "ssspy_ilrma_normalize_output_tracked": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _p, _z,
                                                  _p, _p]),
    "ssspy_ilrma_iss_weight": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _i, _d, _p]),
    "ssspy_ilrma_iss_weight_power": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _i, _d, _p]),
    "ssspy_ilrma_loss_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "ssspy_ilrma_loss_data": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _p, _z, _p]),
    "ssspy_ilrma_ip1_update": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _i, _i, _d,
                                    _p, _z, _p, _p]),
    "ssspy_ilrma_deferred_loss_supported": (_i, [_i, _i, _i, _i, _d, _i]),
    "ssspy_ilrma_ip1_update_deferred_loss": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i,
                                                  _d, _i, _i, _d, _p, _z, _p, _p, _p, _p]),
    "ssspy_ilrma_deferred_loss_slots": (_i, [_i, _i, _i, _i, _i, _d, _i]),
    "ssspy_ilrma_deferred_logdet_slots": (_i, [_i, _i, _i, _i, _i, _d, _i]),
    "ssspy_ilrma_ip1_update_loss_slots": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i,
                                               _d, _i, _i, _d, _p, _z, _p, _p, _q, _p, _p]),
    "ssspy_fold_scalar_slots_workspace_bytes": (_z, [_q, _i]),
    "ssspy_fold_scalar_slots": (_i, [_p, _q, _i, _p, _p, _z, _p]),
    "ssspy_ilrma_partition_expand": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ssspy_ilrma_partition_update": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d,
                                          _i, _i, _d, _p, _z, _p]),
    "ssspy_ilrma_partition_normalize": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _d, _i, _d, _p,
                                             _z, _p]),
    "ssspy_iva_frame_power_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "ssspy_iva_frame_power": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "ssspy_separate_frame_power": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _z, _p]),
    "ssspy_iva_weight": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _d, _p]),
    "ssspy_iva_loss_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "ssspy_gmnmf_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "ssspy_gmnmf_update": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _d, _p, _z, _p]),
    "ssspy_gmnmf_loss_workspace_bytes": (_z, [_i, _i, _i]),
    "ssspy_gmnmf_loss": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _d, _p, _z, _p]),
    "ssspy_gmnmf_separate": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _d, _p]),
    "ssspy_eigh_general": (_i, [_p, _p, _p, _p, _q, _i, _i, _p, _p]),
    "ssspy_sqrtmh": (_i, [_p, _p, _q, _i, _i, _i, _d, _p]),
    "ssspy_gmeanmh": (_i, [_p, _p, _p, _q, _i, _i, _p]),
    "ssspy_lqpqm2": (_i, [_p, _p, _p, _p, _q, _i, _i, _i, _d, _p, _p, _p]),
    "ssspy_lqpqm2_masked": (_i, [_p, _p, _p, _p, _q, _i, _i, _i, _d, _p, _p, _p, _p]),
    "ssspy_stft_frames": (_i, [_q, _i, _i]),
    "ssspy_stft_workspace_bytes": (_z, [_i]),
    "ssspy_stft": (_i, [_p, _p, _p, _d, _i, _i, _q, _i, _i, _p, _p]),
    "ssspy_istft_samples": (_q, [_i, _i, _i]),
    "ssspy_istft": (_i, [_p, _p, _p, _d, _p, _i, _i, _i, _i, _i, _p, _p]),
    "ssspy_fastmnmf_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "ssspy_fastmnmf_update": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _d, _p,
                                   _z, _p, _p]),
    "ssspy_fastmnmf_handover_doubles": (_z, [_i, _i, _i, _i, _i, _i]),
    "ssspy_fastmnmf_update_handover": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _d,
                                            _p, _z, _p, _p, _p, _p]),
    "ssspy_fastmnmf_deferred_logdet_slots": (_i, [_i, _i, _i, _i, _i, _i]),
    "ssspy_fastmnmf_update_handover_logdet": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i,
                                                   _i, _d, _p, _z, _p, _p, _p, _p, _q, _p]),
    "ssspy_fastmnmf_loss_handover_slots": (_i, [_i, _i, _i, _i, _i, _i]),
    "ssspy_fastmnmf_loss_data_handover_slots": (_i, [_p, _p, _p, _p, _p, _q, _i, _i, _i, _i, _i, _i,
                                                     _p]),
    "ssspy_fastmnmf_loss_data_handover": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z,
                                               _p]),
    "ssspy_fastmnmf_loss_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "ssspy_fastmnmf_diagonalizer_covariance": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p,
                                                    _z, _p]),
    "ssspy_fastmnmf_loss_data": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _z, _p]),
    "ssspy_fastmnmf_weights": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "ssspy_fastmnmf_separate": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _d, _p,
                                     _z, _p, _p]),
    "ssspy_fastmnmf_separate_eig": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p,
                                         _p, _p, _z, _p, _p]),
}
GMNMF_BASIS, GMNMF_ACTIVATION, GMNMF_SPATIAL, GMNMF_NORMALIZE, GMNMF_ALL = 1, 2, 4, 8, 15
GMNMF_LATENT = 16
MNMF_BASIS, MNMF_ACTIVATION, MNMF_DIAGONALIZER, MNMF_SPATIAL, MNMF_NORMALIZE, MNMF_ALL = 1, 2, 4, 8, 16, 31

_lib = None


class HipLibraryError(RuntimeError):
    """libssspy_amd.so is missing, fails to load, or an entry point returned an error."""


def load():
    """Load the shared library (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "{} not found: build it with `python -m ssspy_amd._build` "
            "(there is no CPU fallback for the device path)".format(LIB_PATH)
        )
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=os.RTLD_NOW)  # unresolved symbols fail here, not at first call
    except OSError as e:  # e.g. libamdhip64 missing
        raise HipLibraryError("cannot load {}: {}".format(LIB_PATH, e)) from e
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ssspy_abi_version() != ABI_VERSION:
        raise HipLibraryError(
            "{} implements ABI version {}, this binding was written against {}: rebuild it with "
            "`python -m ssspy_amd._build`".format(LIB_PATH, lib.ssspy_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    """Translate a C-ABI status into the Python exception the reference would raise."""
    if rc == OK:
        return
    msg = load().ssspy_last_error().decode()
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError("{}: {}".format(what, msg))
    if rc == ERR_BADARG:
        raise ValueError("{}: {}".format(what, msg))
    raise HipLibraryError("{}: {} (status {})".format(what, msg, rc))


def raise_if_singular(count, what):
    """Device kernels count singular per-bin systems; the reference raises LinAlgError."""
    if count:
        raise np.linalg.LinAlgError("Singular matrix ({} bin(s) in {})".format(int(count), what))
