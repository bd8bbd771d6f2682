"""ctypes binding of libmm_native.so (C ABI declared in include/mm_native.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MM_NATIVE_LIB: another build of the same library for ONE process (A/B variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("MM_NATIVE_LIB") or os.path.join(HERE, "csrc", "libmm_native.so")

MM_F32, MM_F16, MM_BF16 = 0, 1, 2
MASK_NONE, MASK_LEN_I32, MASK_U8, MASK_I64, MASK_F32 = 0, 1, 2, 3, 4
TKL_SAT_EMBEDDING, TKL_SAT_LOG = 0, 1
SIM_ROUND, SUM_ROUND = 1, 2          # MM_SIM_ROUND / MM_SUM_ROUND
ABI_VERSION = 4

_c = ctypes
_vp, _i64, _i, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t

# symbol -> (restype, argtypes): must list every function declared in include/mm_native.h
SIGNATURES = {
    "mm_abi_version": (_i, []),
    "mm_last_error": (_c.c_char_p, []),
    "mm_maxsim_workspace_bytes": (_sz, [_i64, _i64, _i, _i, _i, _i]),
    "mm_maxsim_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_maxsim_fwd_batched": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mm_maxsim_inbatch_workspace_bytes": (_sz, [_i64, _i64, _i, _i, _i, _i]),
    "mm_maxsim_inbatch_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_maxsim_ragged_workspace_bytes": (_sz, [_i64, _i64, _i, _i]),
    "mm_maxsim_ragged_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_hbm_stream_probe": (_i, [_vp, _i64, _i, _vp]),
    "mm_maxsim_bwd_workspace_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "mm_maxsim_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_workspace_bytes": (_sz, [_i64, _i64, _i, _i, _i, _i]),
    "mm_kernel_pool_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_ex_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _c.c_float, _vp, _vp, _i64, _i64,
                                   _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_ex_fwd2": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _c.c_float, _vp, _vp, _vp, _i64, _i64,
                                    _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_ex_bwd2": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _c.c_float, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_ex_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _c.c_float, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_tkl_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "mm_tkl_bwd_workspace_bytes2": (_sz, [_i64, _i, _i, _i]),
    "mm_tkl_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_multi_workspace_bytes": (_sz, [_i64, _i64, _i, _i, _i, _i, _i, _i]),
    "mm_kernel_pool_multi_fwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _c.c_float, _vp, _i64, _i64,
                                      _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_kernel_pool_bwd_workspace_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "mm_kernel_pool_bwd_workspace_bytes2": (_sz, [_i64, _i, _i, _i, _i, _i]),
    "mm_kernel_pool_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_dot_topk_workspace_bytes": (_sz, [_i64, _i, _i]),
    "mm_dot_topk_fwd": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _c.c_float, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mm_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "mm_tkl_workspace_bytes": (_sz, [_i64, _i64, _i, _i, _i]),
    "mm_tkl_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "mm_tkl_fwd_peaks": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Loads the shared library once; raises if it was not built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -m matchmaker_amd.build` "
                "(matchmaker_amd has no CPU / eager fallback for the scoring path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if L.mm_abi_version() != ABI_VERSION:
            raise NativeError(f"{LIB_PATH} has ABI version {L.mm_abi_version()}, this binding needs {ABI_VERSION}: "
                              "rebuild with `python -m matchmaker_amd.build --force`")
        _lib = L
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = lib().mm_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what} failed ({code}): {msg}")
