"""ctypes binding of libmi_fp8.so (include/mi_fp8.h).  There is NO fallback: if the HIP library is
missing or a call fails, this module raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLM_FP8_AMD_LIB") or os.path.join(_HERE, "libmi_fp8.so")  # the override is for A/B timing of another build of the same ABI
# the lab build (same sources, -DMI_DIAG): timing / ablation / stamp builds for tools/ -- never loaded by the package itself
LAB_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "bin", "libmi_fp8_lab.so")

MI_FMT_E4M3 = 0
MI_FMT_E5M2 = 1
ABI_VERSION = 4

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_p = ctypes.c_void_p

class GemmProblem(ctypes.Structure):
    """include/mi_fp8.h `mi_gemm_problem`."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("D", ctypes.c_void_p), ("sa_inv", ctypes.c_void_p),
                ("sb_inv", ctypes.c_void_p), ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
                ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldd", ctypes.c_int64)]


# name -> argtypes (all return int unless noted); mirrors include/mi_fp8.h exactly
SIGNATURES = {
    "mi_abi_version": [],
    "mi_last_error": [],
    "mi_device_supported": [],
    "mi_cast_amax": [_p, _p, _p, _p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _p],
    "mi_transpose_u8": [_p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _p],
    "mi_scale_update": [_p, _p, _p, _p, _c_int, _c_int, _c_i64, _c_int, _c_int, _p],
    "mi_gemm_fp8_grouped": [_p, _c_int, _c_int, _c_int, _c_int, _p],
    "mi_gemm_fp8": [_p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64,
                    _c_int, _c_int, _c_int, _c_int, _p],
    "mi_mxfp8_quantize": [_p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_mxfp8_quantize_ex": [_p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_i64, _c_int, _p],
    "mi_mxfp8_rope_bwd_quantize": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_int, _p],
    "mi_rope_qkv": [_p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_int, _p],
    "mi_rope_qkv_bwd_cast": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_int, _p],
    "mi_swiglu_cast": [_p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_dswiglu_cast": [_p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_rmsnorm_stats": [_p, _p, _c_i64, _c_i64, ctypes.c_float, _p],
    "mi_norm_cast": [_p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_rmsnorm_bwd": [_p, _p, _p, _p, _p, _p, _p, _c_int, _c_i64, _c_i64, _p],
    "mi_ce_forward": [_p, _p, _p, _p, _c_i64, _c_i64, _p],
    "mi_ce_backward": [_p, _p, _p, _p, _p, _c_i64, _c_i64, _p],
    "mi_ce_backward_cast": [_p, _p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_sumsq_bf16": [_p, _c_i64, _p, _c_int, _p],
    "mi_sumsq_bf16_multi": [_p, _c_int, _p, _c_int, _c_int, _p, _p],
    "mi_adamw_bf16_multi": [_p, _c_int, _p, _c_int, _c_int, _p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                            ctypes.c_float, _c_i64, _p],
    "mi_adamw_cast_bf16_multi": [_p, _c_int, _p, _c_int, _c_int, _p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, _c_i64, _p],
    "mi_adamw_mxcast_bf16_multi": [_p, _c_int, _p, _c_int, _c_int, _p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                   ctypes.c_float, _c_i64, _p],
    "mi_embedding_grad_add": [_p, _p, _p, _p, _c_i64, _c_i64, _c_i64, ctypes.c_float, _c_i64, _p],
    "mi_adamw_bf16": [_p, _p, _p, _p, _c_i64, _p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                      ctypes.c_float, _c_i64, _p],
    "mi_add_rmsnorm_stats": [_p, _p, _p, _p, _c_i64, _c_i64, ctypes.c_float, _p],
    "mi_cast_amax_colsum": [_p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _p],
    "mi_colsum_finish": [_p, _c_i64, _c_i64, _p, _c_int, _p],
    "mi_colsum_finish_multi": [_p, _p, _p, _p, _p, _c_int, _p],
    "mi_gemm_workspace_bytes": [],
    "mi_gemm_set_workspace": [_p, _c_i64],
    "mi_attn_fwd": [_p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_i64, ctypes.c_float,
                    _c_int, _p],
    "mi_attn_bwd": [_p] * 10 + [_c_int] * 5 + [_c_i64] * 8 + [ctypes.c_float, _c_int, _p],
    "mi_mxfp8_norm_quantize": [_p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_mxfp8_swiglu_quantize": [_p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_mxfp8_dswiglu_quantize": [_p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_gemm_mxfp8": [_p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_int, _p],
    "mi_swiglu_cast_bias": [_p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_dswiglu_cast_bias": [_p, _p, _p, _p, _p, _p, _p, _p, _c_i64, _c_i64, _c_int, _p],
    "mi_add_bias_rmsnorm_stats": [_p, _p, _p, _p, _p, _c_i64, _c_i64, ctypes.c_float, _p],
    "mi_gemm_fp8_clock": [_p, _p, _p, _p, _p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _p, _p],
}
# entry points only the lab build exports (#ifdef MI_DIAG in include/mi_fp8.h)
LAB_SIGNATURES = {
    "mi_attn_fwd_diag": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_i64, ctypes.c_float, _p],
}

_lib = None


def _bind(path: str, signatures) -> ctypes.CDLL:
    lib = ctypes.CDLL(path)
    for name, argtypes in signatures.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "mi_last_error" else (_c_i64 if name == "mi_gemm_workspace_bytes" else _c_int)
    v = lib.mi_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{os.path.basename(path)} ABI version {v} != expected {ABI_VERSION}; rebuild it")
    return lib


def load() -> ctypes.CDLL:
    """Load the library once; raise ImportError loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C llm_fp8_amd/csrc`. "
            "llm_fp8_amd has no CPU/PyTorch fallback for the FP8 path.")
    _lib = _bind(LIB_PATH, SIGNATURES)
    return _lib


def use_lab_library() -> ctypes.CDLL:
    """tools/ only: make every later call of this process go to the lab build (the product entry points + the diagnostic algo
    values of mi_gemm_fp8 + mi_attn_fwd_diag).  Must run before anything loaded the product library."""
    global _lib
    if _lib is not None and getattr(_lib, "_mi_is_lab", False):
        return _lib
    if _lib is not None:
        raise RuntimeError("use_lab_library(): the product library is already loaded in this process")
    if not os.path.exists(LAB_LIB_PATH):
        raise ImportError(f"{LAB_LIB_PATH} not found: run `make -C llm_fp8_amd/csrc lab`")
    _lib = _bind(LAB_LIB_PATH, {**SIGNATURES, **LAB_SIGNATURES})
    _lib._mi_is_lab = True
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mi_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libmi_fp8 {what} failed (rc={rc}): {msg}")
