"""
ctypes binding of libfdmi.so (C ABI declared in include/fdmi.h).

There is deliberately no fallback: if the shared library is missing or the call
fails, a RuntimeError / FdmiError is raised.  The product path never computes on
the CPU.
"""
import ctypes as C
import os
from typing import Optional

LIB_PATH = os.environ.get("FDMI_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libfdmi.so")  # FDMI_LIB: A/B builds

FD_OK = 0
FD_POS = {"absolute": 0, "relative_key": 1, "relative_key_query": 2}
FD_DEC = {"mlp": 0, "linear": 1}
FD_PREC_F32 = 0
FD_PREC_F16X3 = 1
FD_PREC = {"f32": 0, "f16x3": 1}
ABI_VERSION = 4


class FdmiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libfdmi error {code}: {msg}")
        self.code = code


class FdConfig(C.Structure):
    _fields_ = [
        ("n_features", C.c_int32),
        ("d_model", C.c_int32),
        ("n_heads", C.c_int32),
        ("d_ff", C.c_int32),
        ("n_layers", C.c_int32),
        ("max_pos", C.c_int32),
        ("pos_type", C.c_int32),
        ("decoder", C.c_int32),
        ("ln_eps", C.c_float),
    ]


# every symbol include/fdmi.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "fd_abi_version": (C.c_int, []),
    "fd_device_count": (C.c_int, []),
    "fd_create": (C.c_int, [C.POINTER(FdConfig), C.c_int, C.POINTER(_P)]),
    "fd_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "fd_finalize": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int]),
    "fd_destroy": (None, [_P]),
    "fd_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "fd_forward": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P]),
    "fd_forward_ex": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "fd_p_sample_step": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, _P]),
    "fd_sample": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_uint64, _P, C.c_int]),
    "fd_sample_ex": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_uint64, C.c_int64, _P, C.c_int]),
    "fd_sample_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_uint64, C.c_int64, _P, C.c_int, _P]),
    "fd_sample_begin_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, _P, C.c_int, _P]),
    "fd_sample_steps_dev": (C.c_int, [_P, C.c_int, _P, C.c_int, _P]),
    "fd_sample_end_dev": (C.c_int, [_P, _P, _P]),
    "fd_comm_unique_id": (C.c_int, [_P]),
    "fd_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "fd_gather_dev": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "fd_comm_destroy": (C.c_int, [_P]),
    "fd_philox_normal_dev": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P]),
    "fd_nerf": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "fd_shift_trim_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "fd_test_wrap": (C.c_int, [C.c_int, C.c_int, _P, C.c_int64, _P]),
    "fd_test_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "fd_test_gemm_ln": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_int, C.c_int, C.c_int]),
    "fd_test_gemm_time": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "fd_profile_every": (C.c_int, [_P, C.c_int]),
    "fd_profile_reset": (C.c_int, [_P]),
    "fd_profile_count": (C.c_int, [_P]),
    "fd_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fd_synchronize": (C.c_int, [_P]),
    "fd_check_finite": (C.c_int, [_P]),
    "fd_fused_attn_supported": (C.c_int, [_P, C.c_int]),
    "fd_debug_read": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "fd_last_error": (C.c_char_p, []),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen libfdmi.so and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `python -m foldingdiff_amd.build` (needs hipcc). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    # the version first: an older library then fails with a version message, not with a missing-symbol AttributeError
    lib.fd_abi_version.restype = C.c_int
    lib.fd_abi_version.argtypes = []
    v = lib.fd_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: libfdmi ABI version {v}, this binding expects {ABI_VERSION}; "
                           "rebuild with `python -m foldingdiff_amd.build --force`")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => header / library mismatch, surface it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES.keys())


def check(rc: int):
    if rc != FD_OK:
        msg = load().fd_last_error()
        raise FdmiError(rc, msg.decode("utf-8", "replace") if msg else "")
