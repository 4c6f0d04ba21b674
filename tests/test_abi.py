"""CPU: the C-ABI library loads and exports every symbol include/mi_fp8.h declares; argument checks
that need no GPU work; the package refuses to run its ops on CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mi_fp8.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef MI_DIAG.*?#endif", "", src, flags=re.S)   # lab-build-only declarations
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for s in ("mi_cast_amax", "mi_scale_update", "mi_gemm_fp8", "mi_mxfp8_quantize", "mi_gemm_mxfp8",
              "mi_last_error", "mi_abi_version"):
        assert s in syms


def test_library_loads_and_exports_every_declared_symbol():
    from llm_fp8_amd import _lib
    lib = _lib.load()
    for s in _declared_symbols():
        assert hasattr(lib, s), f"libmi_fp8.so does not export {s}"
        assert s in _lib.SIGNATURES, f"ctypes binding missing for {s}"
    assert lib.mi_abi_version() == _lib.ABI_VERSION


def test_abi_version_is_one_number_in_header_library_and_binding():
    """include/mi_fp8.h `MI_ABI_VERSION` = what the built library reports = what the ctypes binding expects (bumped in round 2:
    entry points were added in round 1 without a bump)."""
    from llm_fp8_amd import _lib
    src = open(os.path.join(ROOT, "include", "mi_fp8.h")).read()
    m = re.search(r"#define\s+MI_ABI_VERSION\s+(\d+)", src)
    assert m is not None
    assert int(m.group(1)) == _lib.load().mi_abi_version() == _lib.ABI_VERSION >= 2


def test_product_library_has_no_lab_surface():
    """Round-2 verdict: product ABI and lab bench were one surface.  The timing / ablation / stamp builds now live in
    tools/bin/libmi_fp8_lab.so (-DMI_DIAG); the shipped library neither exports mi_attn_fwd_diag nor accepts a diagnostic algo."""
    from llm_fp8_amd import _lib
    lib = _lib.load()
    assert not hasattr(lib, "mi_attn_fwd_diag")
    fake = 0x10000  # non-null, 16-byte aligned; the algo is rejected before anything is launched or dereferenced
    for algo in (7, 8, 10, 11, 12, 13, 14, 15, 20, 21, 22, 27, 28, 29, 30, 46):
        rc = lib.mi_gemm_fp8(fake, fake, fake, fake, fake, None, 256, 256, 256, 256, 256, 256, 0, 0, 0, algo, None)
        assert rc == -1 and b"lab library" in lib.mi_last_error(), algo
    for algo in (18, 19):
        rc = lib.mi_gemm_mxfp8(fake, fake, fake, fake, fake, None, 256, 256, 256, 0, 0, 0, algo, None)
        assert rc == -1


def test_argument_errors_are_reported_not_thrown():
    from llm_fp8_amd import _lib
    lib = _lib.load()
    rc = lib.mi_cast_amax(None, None, None, None, None, 8, 8, 8, 8, 0, None)
    assert rc == -1 and b"non-null" in lib.mi_last_error()
    rc = lib.mi_scale_update(None, None, None, None, 16, 4, 4, 0, 0, None)
    assert rc == -1
    rc = lib.mi_gemm_fp8(None, None, None, None, None, None, 16, 16, 16, 16, 16, 16, 0, 0, 0, 0, None)
    assert rc == -1 and b"null operand" in lib.mi_last_error()
    with pytest.raises(RuntimeError, match="null operand"):
        _lib.check(rc, "mi_gemm_fp8")


def test_ops_refuse_cpu_tensors():
    from llm_fp8_amd.pytorch import ops
    x = torch.zeros((8, 8), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cast_amax(x, torch.ones(1), None, 0)
