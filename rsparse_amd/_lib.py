"""ctypes binding of librsparse_wrmf_hip.so -- the C ABI declared in include/rsparse_wrmf_hip.h.

There is no CPU fallback: if the library is missing the import of any compute entry point fails
loudly (build it with `python -m rsparse_amd.build` or `__graft_entry__.build()`).
"""
import ctypes
from pathlib import Path

import os

LIB_PATH = Path(os.environ.get("RSPARSE_HIP_LIB", Path(__file__).resolve().parent / "lib" / "librsparse_wrmf_hip.so"))

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_RUNTIME, ERR_NUMERIC = 0, 1, 2, 3, 4
SOLVER_CHOLESKY, SOLVER_CG, SOLVER_NNLS = 0, 1, 2

_c_int, _c_uint, _c_dbl, _c_i64 = ctypes.c_int, ctypes.c_uint, ctypes.c_double, ctypes.c_int64
_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/rsparse_wrmf_hip.h one to one
SIGNATURES = {
    "rsparse_hip_last_error": (ctypes.c_char_p, []),
    "rsparse_hip_abi_version": (_c_int, []),
    "rsparse_hip_device_count": (_c_int, []),
    "rsparse_hip_set_device": (_c_int, [_c_int]),
    "rsparse_hip_als_implicit_float": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_int,
                                                _c_uint, _c_uint, _c_int, _c_int, _c_dbl, _vp, _c_int, _c_int, _vp]),
    "rsparse_hip_als_implicit_double": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_int,
                                                 _c_uint, _c_uint, _c_int, _c_int, _c_dbl, _vp, _c_int, _c_int, _vp]),
    "rsparse_hip_als_explicit_float": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_uint,
                                                _c_uint, _c_uint, _c_int, _c_int, _c_int, _vp]),
    "rsparse_hip_als_explicit_double": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_uint,
                                                 _c_uint, _c_uint, _c_int, _c_int, _c_int, _vp]),
    "rsparse_hip_gramian_float": (_c_int, [_vp, _c_int, _c_i64, _c_dbl, _vp]),
    "rsparse_hip_gramian_double": (_c_int, [_vp, _c_int, _c_i64, _c_dbl, _vp]),
    "rsparse_hip_csc_create_host": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, ctypes.POINTER(_vp)]),
    "rsparse_hip_csc_create_device": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, ctypes.POINTER(_vp)]),
    "rsparse_hip_gramian_absmax_device": (_c_int, [_vp, _c_int, _c_i64, _c_dbl, _vp, _vp, _vp, _vp]),
    "rsparse_hip_csc_destroy": (_c_int, [_vp]),
    "rsparse_hip_csc_info": (_c_int, [_vp, ctypes.POINTER(_c_i64)]),
    "rsparse_hip_csc_freeze_values": (_c_int, [_vp, _c_int]),
    "rsparse_hip_als_implicit_bias_device": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_uint, _c_int, _vp, _vp]),
    "rsparse_hip_als_implicit_global_bias_device": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_uint, _c_uint, _c_int,
                                                             _c_int, _c_dbl, _c_int, _vp, _vp, _vp]),
    "rsparse_hip_initialize_biases_implicit_device": (_c_int, [_vp, _vp, _vp, _vp, _c_dbl, _c_int, _c_int,
                                                               ctypes.POINTER(_c_dbl), _vp]),
    "rsparse_hip_initialize_biases_float": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_dbl, _c_int,
                                                     _c_int, _c_int, _c_int, ctypes.POINTER(_c_dbl)]),
    "rsparse_hip_initialize_biases_double": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_dbl, _c_int,
                                                      _c_int, _c_int, _c_int, ctypes.POINTER(_c_dbl)]),
    "rsparse_hip_als_explicit_bias_device": (_c_int, [_vp, _vp, _vp, _c_int, _c_dbl, _c_uint, _c_uint, _c_int, _c_int,
                                                      _vp, _vp]),
    "rsparse_hip_initialize_biases_explicit_device": (_c_int, [_vp, _vp, _vp, _vp, _c_dbl, _c_int, _c_int, _c_int,
                                                               ctypes.POINTER(_c_dbl), _vp]),
    "rsparse_hip_values_subtract_mean_device": (_c_int, [_c_i64, _vp, _vp, ctypes.POINTER(_c_dbl), _vp]),
    "rsparse_hip_bias_sweep_explicit_device": (_c_int, [_vp, _vp, _c_dbl, _c_int, _c_int, _vp, _vp]),
    "rsparse_hip_bias_prep_implicit_device": (_c_int, [_vp, _c_int, _c_dbl, _vp, _vp, _vp]),
    "rsparse_hip_bias_sweep_implicit_device": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _c_dbl, _vp, _vp]),
    "rsparse_hip_bias_sweep_explicit_f64_device": (_c_int, [_vp, _vp, _c_dbl, _c_int, _c_int, _vp, _vp]),
    "rsparse_hip_bias_prep_implicit_f64_device": (_c_int, [_vp, _c_int, _c_dbl, _vp, _vp, _vp]),
    "rsparse_hip_bias_sweep_implicit_f64_device": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _c_dbl, _vp, _vp]),
    "rsparse_hip_csc_transpose_device": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rsparse_hip_values_to_float_device": (_c_int, [_c_i64, _vp, _vp, _vp]),
    "rsparse_hip_gramian_device": (_c_int, [_vp, _c_int, _c_i64, _c_dbl, _vp, _vp, _vp]),
    "rsparse_hip_als_implicit_device": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_dbl, _c_uint, _c_uint, _vp, _vp, _vp]),
    "rsparse_hip_als_explicit_device": (_c_int, [_vp, _vp, _vp, _c_int, _c_dbl, _c_uint, _c_uint, _c_int, _vp, _vp]),
    "rsparse_hip_weighted_sumsq_device": (_c_int, [_vp, _c_int, _c_i64, _vp, _vp, _vp]),
    "rsparse_hip_top_product": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_uint, _c_uint, _vp, _vp, _vp, _c_int, _c_dbl, _vp, _vp]),
    "rsparse_hip_top_product_device": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _c_int, _c_dbl, _vp, _vp, _vp]),
    "rsparse_hip_top_product_f64_device": (_c_int, [_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp,
                                                    _c_int, _c_dbl, _vp, _vp, _vp]),
    "rsparse_hip_csc_f64_create_device": (_c_int, [_c_int, _c_int, _vp, _vp, _vp, ctypes.POINTER(_vp)]),
    "rsparse_hip_csc_f64_destroy": (_c_int, [_vp]),
    "rsparse_hip_gramian_f64_device": (_c_int, [_vp, _c_int, _c_i64, _c_dbl, _vp, _vp, _vp]),
    "rsparse_hip_als_f64_device": (_c_int, [_vp, _c_int, _vp, _vp, _vp, _c_int, _c_dbl, _c_uint, _c_uint, _c_int, _c_int,
                                            _c_int, _c_dbl, _vp, _vp]),
    "rsparse_hip_initialize_biases_f64_device": (_c_int, [_vp, _vp, _vp, _vp, _c_dbl, _c_int, _c_int, _c_int, _c_int,
                                                          ctypes.POINTER(_c_dbl), _vp]),
    "rsparse_hip_values_subtract_mean_f64_device": (_c_int, [_c_i64, _vp, _vp, ctypes.POINTER(_c_dbl), _vp]),
    "rsparse_hip_weighted_sumsq_f64_device": (_c_int, [_vp, _c_int, _c_i64, _vp, _vp, _vp]),
    "rsparse_hip_profile_enable": (_c_int, [_c_int]),
    "rsparse_hip_set_launch_mode": (_c_int, [_c_int]),
    "rsparse_hip_set_f64_long_rows": (_c_int, [_c_int, _c_int]),
    "rsparse_hip_profile_last": (_c_int, [ctypes.POINTER(_c_dbl)]),
    "rsparse_hip_profile_last_names": (_c_int, [ctypes.c_char_p, _c_int]),
    "rsparse_hip_take_numeric_failures": (_c_int, [ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    # layer (4): the multi-GPU context (wrmf_ctx.cpp)
    "rsparse_hip_ctx_create": (_c_int, [_c_int, _vp, _c_int, ctypes.POINTER(_vp)]),
    "rsparse_hip_ctx_destroy": (_c_int, [_vp]),
    "rsparse_hip_ctx_set_matrix": (_c_int, [_vp, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp, _c_int, _c_int]),
    "rsparse_hip_ctx_set_factors": (_c_int, [_vp, _c_int, _vp, _vp]),
    "rsparse_hip_ctx_get_factors": (_c_int, [_vp, _vp, _vp]),
    "rsparse_hip_ctx_half_iteration": (_c_int, [_vp, _c_int, _c_int, _c_dbl, _c_uint, _c_uint, _c_int, ctypes.POINTER(_c_dbl)]),
    "rsparse_hip_ctx_take_numeric_failures": (_c_int, [_vp, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    "rsparse_hip_ctx_info": (_c_int, [_vp, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_dbl)]),
}


class RsparseHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librsparse_wrmf_hip: [%d] %s" % (code, msg))
        self.code = code


class UnsupportedOnDevice(RsparseHipError, NotImplementedError):
    """RSPARSE_HIP_ERR_UNSUPPORTED: the reference-side shim would keep its CPU path here."""


_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                "%s not found -- the HIP extension is required (no CPU fallback). "
                "Build it with `python -m rsparse_amd.build`." % LIB_PATH)
        try:  # let torch take its first look at the GPU before this library initialises the HIP runtime:
            import torch  # torch.cuda.is_available() answers False if it is first asked afterwards
            torch.cuda.is_available()
        except ImportError:
            pass
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def check(code):
    if code == OK:
        return
    msg = load().rsparse_hip_last_error().decode("utf-8", "replace")
    if code == ERR_UNSUPPORTED:
        raise UnsupportedOnDevice(code, msg)
    raise RsparseHipError(code, msg)
