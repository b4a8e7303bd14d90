"""ctypes binding of libstgcn_hip.so (C ABI declared in include/stgcn_hip.h).

The library is the product: if it is missing or fails to load, every op raises — there is
no CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_long, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STGCN_LIB") or os.path.join(_HERE, "libstgcn_hip.so")   # STGCN_LIB: diagnostic builds
ABI_VERSION = 8

# stgcn_math / flags (include/stgcn_hip.h)
MATH_F32 = 0
MATH_BF16X3 = 1
MATH_BF16 = 2
MATH_F32_VALU = 3
MATH_MASK = 0xF
OUT_BF16 = 0x10
RAW = 0x20
IN_NTVC = 0x40    # stem entry points: x is (N,T,V,Cin)
OUT_NTVC = 0x80   # stem entry points: out is (N,T,V,C)
EMBED_TS = 0x100  # stgcn_patch_embed: rows ordered (clip, joint, frame)
BN_FROZEN = 0x200  # training entry points: BatchNorm on its running statistics (eval mode under autograd)
STEM_F16MX = 0x400  # stem entry points, with MATH_BF16X3: fp16 x fp16 + two scaled-e4m3 residual products where KF7 covers the shape
CONV_ALONG_V = 0x800  # stgcn_tcn_forward[_packed], MATH_F32_VALU: convolve along the joint axis (Unit2D(dim=3))
MATH_F16MX = MATH_BF16X3 | STEM_F16MX   # as a "math mode" of the modules: bf16x3 everywhere, KF7 in the fused stem

STATUS = {0: "STGCN_OK", -1: "STGCN_ERR_ARG", -2: "STGCN_ERR_UNSUPPORTED",
          -3: "STGCN_ERR_WORKSPACE", -4: "STGCN_ERR_HIP"}

_P = c_void_p
# name -> (restype, argtypes); one entry per symbol declared in include/stgcn_hip.h
PROTOTYPES = {
    "stgcn_version": (c_int, []),
    "stgcn_last_error": (c_char_p, []),
    "stgcn_bn_fold": (c_int, [_P, _P, _P, _P, _P, c_float, _P, _P, c_int, _P]),
    "stgcn_agcn_attention": (c_int, [_P] * 7 + [c_int] * 6 + [_P]),
    "stgcn_agcn_forward": (c_int, [_P] * 16 + [c_int] * 7 + [_P]),
    "stgcn_tcn_packed_bytes": (c_size_t, [c_int, c_int, c_int, c_uint]),
    "stgcn_tcn_supported": (c_int, [c_int] * 6 + [c_uint]),
    "stgcn_tcn_pack": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_uint, _P]),
    "stgcn_tcn_forward_packed": (c_int, [_P] * 4 + [c_int] * 7 + [c_uint, _P]),
    "stgcn_tcn_forward": (c_int, [_P] * 5 + [c_int] * 7 + [_P, c_size_t, c_uint, _P]),
    "stgcn_stem_prep_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_uint]),
    "stgcn_stem_supported": (c_int, [c_int] * 6 + [c_uint]),
    "stgcn_stem_prepare": (c_int, [_P] * 11 + [c_int] * 4 + [c_uint, _P]),
    "stgcn_stem_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "stgcn_stem_features_used": (c_int, [c_int] * 6 + [c_uint]),
    "stgcn_stem_kernel_name": (c_char_p, [c_int] * 6 + [c_uint]),
    "stgcn_stem_attention": (c_int, [_P] * 7 + [c_size_t] + [c_int] * 8 + [c_uint, _P]),
    "stgcn_stem_tail_prepared": (c_int, [_P] * 2 + [c_size_t] + [_P] * 3 + [c_int] * 7 + [c_uint, _P]),
    "stgcn_stem_forward_prepared": (c_int, [_P] * 9 + [c_size_t] + [_P] + [c_int] * 8 + [c_uint, _P]),
    "stgcn_patch_embed": (c_int, [_P] * 5 + [c_int] * 5 + [c_uint, _P]),
    "stgcn_step_stats": (c_int, [_P, c_int, _P, c_int, c_int, c_long, c_long, c_float, _P, _P, _P, c_int, c_int, _P]),
    "stgcn_agcn_train_ws_bytes": (c_size_t, [c_int] * 7),
    "stgcn_agcn_forward_train": (c_int, [_P] * 18 + [c_float, c_float, _P, _P, c_size_t] + [_P] * 4 + [c_int] * 7 + [c_uint, _P]),
    "stgcn_agcn_backward_ws_bytes": (c_size_t, [c_int] * 7),
    "stgcn_agcn_backward_train": (c_int, [_P] * 34 + [_P, c_size_t] + [c_int] * 7 + [c_uint, _P]),
    "stgcn_tcn_train_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "stgcn_tcn_forward_train": (c_int, [_P] * 7 + [c_float, c_float, _P, c_size_t] + [_P] * 4 + [c_int] * 7 + [c_uint, _P]),
    "stgcn_tcn_backward_ws_bytes": (c_size_t, [c_int] * 7 + [c_uint]),
    "stgcn_tcn_backward_train": (c_int, [_P] * 13 + [_P, c_size_t] + [c_int] * 7 + [c_uint, _P]),
}

_lib = None
_lock = threading.Lock()


class StgcnError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, fn: str, code: int, msg: str):
        super().__init__(f"{fn} -> {STATUS.get(code, code)}: {msg}")
        self.code = code


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python st-gcn-altformer_amd/stgcn_amd/build.py` "
                "(needs hipcc; there is no fallback path)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError = ABI mismatch, on purpose
            fn.restype = res
            fn.argtypes = args
        ver = handle.stgcn_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"libstgcn_hip.so ABI {ver} != binding ABI {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


def call(name: str, *args):
    """Invoke a status-returning entry point and raise StgcnError on failure."""
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc != 0:
        msg = handle.stgcn_last_error()
        raise StgcnError(name, rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
