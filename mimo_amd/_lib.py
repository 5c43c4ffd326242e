"""ctypes binding of libmimo_hip.so (C ABI declared in include/mimo_hip.h).

The product path has no CPU fallback: if the shared library is missing or cannot be loaded
this module raises, and so does every engine call.  Build it with
``make -C mimo_amd/csrc`` (or ``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIMO_HIP_LIB") or os.path.join(_HERE, "libmimo_hip.so")   # (MIMO_HIP_LIB: kernel-variant experiments)

# error codes / flags (mirror include/mimo_hip.h)
OK = 0
E_INVALID, E_HIP, E_NODATA, E_UNSUPPORTED, E_STATE, E_NOMEM, E_INTERNAL = -1, -2, -3, -4, -5, -6, -7
F_KEEP_RESP, F_KEEP_LOGP, F_KEEP_LSE, F_NO_STATS, F_DEVICE_OUT, F_DEVICE_IN, F_ENTROPY_SPLIT, F_ASYNC = 1, 2, 4, 8, 0x10, 0x20, 0x40, 0x80
F_WEIGHTS_RESIDENT = 0x100
F_DIAG_VAR = 0x200

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# every exported symbol with its signature; tests check the .so exports exactly these
SIGNATURES = {
    "mimo_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "mimo_destroy": (C.c_int, [_vp]),
    "mimo_last_error": (C.c_char_p, [_vp]),
    "mimo_set_stream": (C.c_int, [_vp, _vp]),
    "mimo_upload": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "mimo_attach": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "mimo_set_row_offset": (C.c_int, [_vp, C.c_int64]),
    "mimo_set_structure": (C.c_int, [_vp, C.c_int]),
    "mimo_estep": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "mimo_estep_weighted": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "mimo_wait": (C.c_int, [_vp, _vp, _vp]),
    "mimo_gibbs_labels": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_uint64, C.c_uint64, _vp,
                                    C.c_int, _vp, _vp]),
    "mimo_weighted_stats": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "mimo_label_stats": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "mimo_table_entropy": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _dp]),
    "mimo_sample_from_log": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, _vp, C.c_uint64, C.c_uint64, C.c_int, _vp, _vp]),
    "mimo_random_resp_stats": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_int, _vp]),
    "mimo_host_nw_vi": (C.c_int, [C.c_int, C.c_int] + [_vp] * 13),
    "mimo_host_nw_vi_tied": (C.c_int, [C.c_int, C.c_int] + [_vp] * 14),
    "mimo_host_nw_vlb": (C.c_int, [C.c_int, C.c_int] + [_vp] * 16),
    "mimo_host_legacy_draws_inplace": (C.c_int, [_vp, _vp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 _vp, _vp, _vp, _vp, C.POINTER(C.c_int), _vp]),
    "mimo_host_py_sample": (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int64, C.c_int64, C.c_int, _vp]),
    "mimo_host_hier_vi": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_double, _vp, C.c_double] + [_vp] * 9),
    "mimo_host_legacy_draws": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                         C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "mimo_host_gmm_vi_sweep": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_vp] * 26),
    "mimo_host_gmm_vi_bound": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_vp] * 21),
    "mimo_host_mnw_vi": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int] + [_vp] * 15),
    "mimo_host_nw_gibbs": (C.c_int, [C.c_int, C.c_int] + [_vp] * 10),
    "mimo_host_digamma": (C.c_double, [C.c_double]),
    "mimo_host_checksum": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "mimo_predict": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                               _vp, _vp, _vp, _vp, _vp, _vp]),
    "mimo_predict_flags": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                     _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "mimo_get_resp": (C.c_int, [_vp, _vp]),
    "mimo_get_resp_columns": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "mimo_get_logp": (C.c_int, [_vp, _vp]),
    "mimo_get_lse": (C.c_int, [_vp, _vp]),
    "mimo_get_labels": (C.c_int, [_vp, _vp]),
    "mimo_philox_uniform": (C.c_double, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "mimo_profile": (C.c_int, [_vp, C.c_int]),
    "mimo_profile_read": (C.c_int, [_vp, _dp, C.POINTER(C.c_int64), C.c_int]),
    "mimo_profile_kernels": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mimo_shader_clock_mhz": (C.c_int, [_vp, _dp]),
    "mimo_plan": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "mimo_plan_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.c_char_p, C.c_int]),
    "mimo_nan_info": (C.c_int, [_vp, C.POINTER(C.c_int64), _vp, C.c_int, _vp]),
    "mimo_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mimo_comm_init": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "mimo_comm_destroy": (C.c_int, [_vp]),
    "mimo_tune": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "mimo_data_checksum": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "mimo_debug_fault": (C.c_int, [_vp, C.c_int]),
    "mimo_host_debug_fault": (C.c_int, [C.c_int]),
    "mimo_version": (C.c_char_p, []),
}

_lib = None


class MimoHipError(RuntimeError):
    pass


def load():
    """Load libmimo_hip.so once; raise loudly if it is absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MimoHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `make -C mimo_amd/csrc`.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the same SONAMEs this library
    # links against: the dynamic loader keeps whichever copy arrives first for BOTH.  With this library first,
    # torch ends up on the system runtime and reports "no GPUs found" (seen with the one-rank RCCL test); with
    # torch first both share the bundled one, which works.  So when torch is installed it is imported first.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
