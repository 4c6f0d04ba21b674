"""Thin tensor-level wrappers over the C ABI (include/mi_fp8.h).  Device tensors only: every
function raises if handed a CPU tensor -- there is no PyTorch/CPU fallback for this path."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from .. import _lib
from .profiler import KernelTimer

E4M3, E5M2 = _lib.MI_FMT_E4M3, _lib.MI_FMT_E5M2
FP8_MAX = {E4M3: 448.0, E5M2: 57344.0}


_SK_WORKSPACE = {}


def ensure_gemm_workspace(device: torch.device) -> None:
    """Registers the stream-K workspace of mi_gemm_fp8 / mi_gemm_mxfp8 for `device` (allocated once, zeroed flags; owned here).
    Only the explicit algo 44 uses it (stream-K measured slower than the whole-tile shapes at the reference's sizes)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _SK_WORKSPACE:
        return
    lib = _lib.load()
    with torch.cuda.device(idx):
        nbytes = int(lib.mi_gemm_workspace_bytes())
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=torch.device("cuda", idx))
        torch.cuda.synchronize(idx)
        _lib.check(lib.mi_gemm_set_workspace(ws.data_ptr(), nbytes), "mi_gemm_set_workspace")
    _SK_WORKSPACE[idx] = ws


def default_gemm_algo() -> int:
    """47 (auto: four-wave persistent kernel where it applies, else the eight-wave one) on a single GPU.  Under torch.distributed with more than one rank 5 = the same kernel
    launched with one workgroup per tile: RCCL's collectives overlap the GEMMs under FSDP / DDP and hold some CUs; a
    persistent grid with one workgroup per CU and a static tile list would wait for those CUs, while a plain grid is simply
    scheduled onto the CUs that are free (measured on a free chip: 5 is ~2 % slower than 4, the old 8-phase kernel 3 ~9 %).
    Override with LLM_FP8_AMD_GEMM_ALGO."""
    env = os.environ.get("LLM_FP8_AMD_GEMM_ALGO")
    if env is not None:
        return int(env)
    d = torch.distributed
    if d.is_available() and d.is_initialized() and d.get_world_size() > 1:
        return 5
    # 47 = auto with the persistent four-wave kernel (mi_gemm_w4.hip) on the shapes it takes (256-multiples where the eight-wave kernel's
    # own choice is 256 x 256 tiles), the eight-wave kernel elsewhere; bit-identical outputs.  Default since the cheap mid-tile cursor
    # advance and the scalar alpha multiplies (end of round 3): -2.5 % over the twelve 3B decoder shapes in a timing loop, GEMM rate
    # of the step 0.5310 against 0.5285 on one box (profiles/r03_instep_w4_ab.txt).  LLM_FP8_AMD_GEMM_W4=0: eight-wave kernel only.
    return 47 if os.environ.get("LLM_FP8_AMD_GEMM_W4", "1") != "0" else 0


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("llm_fp8_amd ops need device (HIP) tensors; there is no CPU fallback")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def colsum_finish(partial: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """[P, C] fp32 partial column sums -> [C] in `dtype` (bf16 / fp32): one launch instead of sum + cast."""
    _dev(partial)
    assert partial.dtype == torch.float32 and partial.dim() == 2 and partial.is_contiguous()
    P, C = partial.shape
    if dtype not in (torch.bfloat16, torch.float32):
        return partial.sum(0).to(dtype)
    out = torch.empty(C, dtype=dtype, device=partial.device)
    _lib.check(_lib.load().mi_colsum_finish(partial.data_ptr(), P, C, out.data_ptr(), 0 if dtype == torch.bfloat16 else 1, _stream()),
               "mi_colsum_finish")
    return out


def colsum_finish_multi(items):
    """colsum_finish for a list of (partial [P, C] fp32, dtype) in as few launches as possible (4 per launch); same results."""
    import ctypes
    outs = [None] * len(items)
    todo = []
    for i, (partial, dtype) in enumerate(items):
        _dev(partial)
        assert partial.dtype == torch.float32 and partial.dim() == 2 and partial.is_contiguous()
        if dtype not in (torch.bfloat16, torch.float32):
            outs[i] = partial.sum(0).to(dtype)
        else:
            outs[i] = torch.empty(partial.shape[1], dtype=dtype, device=partial.device)
            todo.append(i)
    for k in range(0, len(todo), 4):
        grp = todo[k:k + 4]
        n = len(grp)
        parts = (ctypes.c_void_p * n)(*[items[i][0].data_ptr() for i in grp])
        Ps = (ctypes.c_int64 * n)(*[items[i][0].shape[0] for i in grp])
        Cs = (ctypes.c_int64 * n)(*[items[i][0].shape[1] for i in grp])
        os_ = (ctypes.c_void_p * n)(*[outs[i].data_ptr() for i in grp])
        dts = (ctypes.c_int * n)(*[0 if items[i][1] == torch.bfloat16 else 1 for i in grp])
        _lib.check(_lib.load().mi_colsum_finish_multi(parts, Ps, Cs, os_, dts, n, _stream()), "mi_colsum_finish_multi")
    return outs


def cast_amax(x: torch.Tensor, scale: torch.Tensor, amax: Optional[torch.Tensor], fmt: int,
              want_y: bool = True, want_t: bool = True,
              y: Optional[torch.Tensor] = None, yT: Optional[torch.Tensor] = None, want_colsum: bool = False):
    """K1/K2.  x bf16 [R, C] contiguous -> (y u8 [R, C], yT u8 [C, R]); amax (1-elem f32 view) is
    atomically maxed with max|x|.  `y` / `yT` may be preallocated row-slices of larger buffers
    (their stride(0) is used as leading dimension)."""
    _dev(x, scale, amax, y, yT)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
    assert scale.dtype == torch.float32 and (amax is None or amax.dtype == torch.float32)
    R, C = x.shape
    if want_y and y is None:
        y = torch.empty((R, C), dtype=torch.uint8, device=x.device)
    if want_t and yT is None:
        yT = torch.empty((C, R), dtype=torch.uint8, device=x.device)
    if y is not None:
        assert y.dtype == torch.uint8 and y.shape == (R, C) and y.stride(1) == 1
    if yT is not None:
        assert yT.dtype == torch.uint8 and yT.shape == (C, R) and yT.stride(1) == 1
    cs = torch.empty(((R + 63) // 64, C), dtype=torch.float32, device=x.device) if want_colsum else None
    tail = (R, C, y.stride(0) if y is not None else C, yT.stride(0) if yT is not None else R, fmt, _stream())
    head = (x.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(), _ptr(amax))
    fn = _lib.load().mi_cast_amax_colsum if want_colsum else _lib.load().mi_cast_amax
    args = head + ((cs.data_ptr(),) if want_colsum else ()) + tail
    t = KernelTimer.active
    if t is None:
        rc = fn(*args)
    else:
        nb = R * C * (2 + (y is not None) + (yT is not None))
        with t.span("cast_amax", f"{R}x{C}", float(R * C), float(nb)):
            rc = fn(*args)
    _lib.check(rc, "mi_cast_amax")
    return (y, yT, cs) if want_colsum else (y, yT)


def scale_update(amax_history: torch.Tensor, scale: torch.Tensor, scale_inv: torch.Tensor,
                 fp8_max: torch.Tensor, margin: int = 0, algo: str = "max") -> None:
    """K3, in place on [H, S] history and [S] scale / scale_inv."""
    _dev(amax_history, scale, scale_inv, fp8_max)
    assert amax_history.dim() == 2 and amax_history.stride(1) == 1 and amax_history.dtype == torch.float32
    H, S = amax_history.shape
    assert scale.numel() == S and scale_inv.numel() == S and fp8_max.numel() == S
    assert scale.is_contiguous() and scale_inv.is_contiguous() and fp8_max.is_contiguous()
    rc = _lib.load().mi_scale_update(amax_history.data_ptr(), scale.data_ptr(), scale_inv.data_ptr(),
                                     fp8_max.data_ptr(), H, S, amax_history.stride(0), margin,
                                     0 if algo == "max" else 1, _stream())
    _lib.check(rc, "mi_scale_update")


def gemm_fp8(a8: torch.Tensor, b8: torch.Tensor, sa_inv: torch.Tensor, sb_inv: torch.Tensor,
             fmt_a: int, fmt_b: int, bias: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, out_dtype: torch.dtype = torch.bfloat16,
             algo: Optional[int] = None) -> torch.Tensor:
    """K4-K6.  D[M,N] = (A[M,K] . B[N,K]^T) * sa_inv * sb_inv (+ bias)."""
    if algo is None:
        algo = default_gemm_algo()
        M_, N_, K_ = a8.shape[0], b8.shape[0], a8.shape[1]
        if algo == 3 and (M_ % 256 or N_ % 256 or K_ % 128):
            algo = 0
        if algo == 47 and (out_dtype != torch.bfloat16 or (out is not None and out.dtype != torch.bfloat16)):
            algo = 0
        if algo == 5 and ((M_ % 256 and M_ % 192) or (N_ % 256 and N_ % 192) or K_ % 256 or out_dtype != torch.bfloat16
                          or (out is not None and out.dtype != torch.bfloat16)):
            algo = 0
    _dev(a8, b8, sa_inv, sb_inv, bias, out)
    if algo == 44:
        ensure_gemm_workspace(a8.device)
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and a8.dim() == 2 and b8.dim() == 2
    assert a8.stride(1) == 1 and b8.stride(1) == 1
    M, K = a8.shape
    N, K2 = b8.shape
    assert K == K2, f"contraction mismatch {K} vs {K2}"
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype in (torch.bfloat16, torch.float32)
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.numel() == N and bias.is_contiguous()
    args = (a8.data_ptr(), b8.data_ptr(), out.data_ptr(), sa_inv.data_ptr(), sb_inv.data_ptr(), _ptr(bias), M, N, K,
            a8.stride(0), b8.stride(0), out.stride(0), fmt_a, fmt_b, 0 if out.dtype == torch.bfloat16 else 1, algo,
            _stream())
    t = KernelTimer.active
    if t is None:
        rc = _lib.load().mi_gemm_fp8(*args)
    else:
        with t.span("gemm_fp8", f"{M}x{N}x{K}", 2.0 * M * N * K, M * K + N * K + out.element_size() * M * N):
            rc = _lib.load().mi_gemm_fp8(*args)
    _lib.check(rc, "mi_gemm_fp8")
    return out


def mxfp8_quantize(x: torch.Tensor, fmt: int = E4M3, rowwise: bool = True, colwise: bool = True, out=None,
                   want_colsum: bool = False):
    """K7.  Returns (y_row [R,C], s_row [C/32,R], y_colT [C,R], s_colT [R/32,C]) (None where not asked); scales are
    block-major: the E8M0 bytes of one 32-block of every row are contiguous.
    `out`: the same four tensors as row-block views of a larger operand's buffers (rows r.. of [N,C] / columns r.. of
    [C/32,N] and [C,N] / rows r/32.. of [N/32,C]) -- x is then quantised in place as part of that operand.
    `want_colsum`: also returns the fp32 partial column sums [ceil(R/128), C] of x (for colsum_finish)."""
    _dev(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
    R, C = x.shape
    y_row = s_row = y_colT = s_colT = None
    ld = R
    if out is not None:
        y_row, s_row, y_colT, s_colT = out
        _dev(y_row, s_row, y_colT, s_colT)
        assert (y_row is not None) == rowwise and (y_colT is not None) == colwise
        if rowwise:
            assert y_row.shape == (R, C) and y_row.is_contiguous() and s_row.shape == (C // 32, R) and s_row.stride(1) == 1
            ld = s_row.stride(0)
        if colwise:
            assert y_colT.shape == (C, R) and y_colT.stride(1) == 1 and s_colT.shape == (R // 32, C) and s_colT.is_contiguous()
            assert not rowwise or y_colT.stride(0) == ld
            ld = y_colT.stride(0)
    else:
        if rowwise:
            y_row = torch.empty((R, C), dtype=torch.uint8, device=x.device)
            s_row = torch.empty((C // 32, R), dtype=torch.uint8, device=x.device)
        if colwise:
            y_colT = torch.empty((C, R), dtype=torch.uint8, device=x.device)
            s_colT = torch.empty((R // 32, C), dtype=torch.uint8, device=x.device)
    cs = torch.empty(((R + 127) // 128, C), dtype=torch.float32, device=x.device) if want_colsum else None
    if out is None and not want_colsum:
        fn, args = _lib.load().mi_mxfp8_quantize, (x.data_ptr(), _ptr(y_row), _ptr(s_row), _ptr(y_colT), _ptr(s_colT), R, C, fmt, _stream())
    else:
        fn, args = _lib.load().mi_mxfp8_quantize_ex, (x.data_ptr(), _ptr(y_row), _ptr(s_row), _ptr(y_colT), _ptr(s_colT), _ptr(cs),
                                                      R, C, ld, fmt, _stream())
    t = KernelTimer.active
    if t is None:
        rc = fn(*args)
    else:
        nb = R * C * (2 + (1 + 1 / 32) * (int(rowwise) + int(colwise)))
        with t.span("mxfp8_quantize", f"{R}x{C}", float(R * C), float(nb)):
            rc = fn(*args)
    _lib.check(rc, "mi_mxfp8_quantize")
    return (y_row, s_row, y_colT, s_colT, cs) if want_colsum else (y_row, s_row, y_colT, s_colT)


def gemm_mxfp8(a8, sa, b8, sb, fmt_a: int = E4M3, fmt_b: int = E4M3, bias=None, out=None,
               out_dtype: torch.dtype = torch.bfloat16, algo: Optional[int] = None) -> torch.Tensor:
    """K8.  Block-scaled D[M,N] = sum_blk 2^(sa+sb-254) sum_32 A.B (+bias)."""
    _dev(a8, sa, b8, sb, bias, out)
    if algo is None:
        algo = default_gemm_algo()
        M_, N_, K_ = a8.shape[0], b8.shape[0], a8.shape[1]
        if algo not in (0, 1, 4, 5) or (algo == 5 and ((M_ % 256 and M_ % 192) or (N_ % 256 and N_ % 192) or K_ % 256
                                                      or out_dtype != torch.bfloat16 or (out is not None and out.dtype != torch.bfloat16))):
            algo = 0
    if algo == 44:
        ensure_gemm_workspace(a8.device)
    M, K = a8.shape
    N, K2 = b8.shape
    assert K == K2 and a8.is_contiguous() and b8.is_contiguous() and sa.is_contiguous() and sb.is_contiguous()
    assert sa.shape == (K // 32, M) and sb.shape == (K // 32, N), "scales must be block-major [K/32, rows]"
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a8.device)
    assert out.is_contiguous() and out.shape == (M, N)
    args = (a8.data_ptr(), sa.data_ptr(), b8.data_ptr(), sb.data_ptr(), out.data_ptr(), _ptr(bias), M, N, K, fmt_a,
            fmt_b, 0 if out.dtype == torch.bfloat16 else 1, algo, _stream())
    t = KernelTimer.active
    if t is None:
        rc = _lib.load().mi_gemm_mxfp8(*args)
    else:
        with t.span("gemm_mxfp8", f"{M}x{N}x{K}", 2.0 * M * N * K, M * K + N * K + out.element_size() * M * N):
            rc = _lib.load().mi_gemm_mxfp8(*args)
    _lib.check(rc, "mi_gemm_mxfp8")
    return out


def rope_qkv_forward(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_q: int, n_kv: int, head_dim: int, seq: int):
    """Split fused [T, (n_q + 2 n_kv) * D] bf16 into q [T, n_q*D], k, v [T, n_kv*D]; q, k rotated (row position = row % seq)."""
    _dev(qkv, cos, sin)
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv.dim() == 2
    T, W = qkv.shape
    assert W == (n_q + 2 * n_kv) * head_dim and cos.dtype == torch.float32 and cos.shape[-1] == head_dim // 2 and cos.shape[0] >= seq
    q = torch.empty((T, n_q * head_dim), dtype=torch.bfloat16, device=qkv.device)
    k = torch.empty((T, n_kv * head_dim), dtype=torch.bfloat16, device=qkv.device)
    v = torch.empty((T, n_kv * head_dim), dtype=torch.bfloat16, device=qkv.device)
    rc = _lib.load().mi_rope_qkv(qkv.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                 T, seq, n_q, n_kv, head_dim, 0, _stream())
    _lib.check(rc, "mi_rope_qkv")
    return q, k, v


def rope_qkv_backward(dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                      n_q: int, n_kv: int, head_dim: int, seq: int) -> torch.Tensor:
    _dev(dq, dk, dv, cos, sin)
    dq, dk, dv = (t.contiguous() for t in (dq, dk, dv))
    T = dq.shape[0]
    out = torch.empty((T, (n_q + 2 * n_kv) * head_dim), dtype=torch.bfloat16, device=dq.device)
    rc = _lib.load().mi_rope_qkv(out.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                 T, seq, n_q, n_kv, head_dim, 1, _stream())
    _lib.check(rc, "mi_rope_qkv")
    return out


def mxfp8_rope_bwd_quantize(dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_q: int,
                            n_kv: int, head_dim: int, seq: int, fmt: int, rowwise: bool = True, colwise: bool = True):
    """rope_qkv_backward + mxfp8_quantize of its result in one launch (same four outputs); the bf16 d(qkv) is never written."""
    _dev(dq, dk, dv, cos, sin)
    dq, dk, dv = (t.contiguous() for t in (dq, dk, dv))
    T, W = dq.shape[0], (n_q + 2 * n_kv) * head_dim
    y_row = s_row = y_colT = s_colT = None
    if rowwise:
        y_row = torch.empty((T, W), dtype=torch.uint8, device=dq.device)
        s_row = torch.empty((W // 32, T), dtype=torch.uint8, device=dq.device)
    if colwise:
        y_colT = torch.empty((W, T), dtype=torch.uint8, device=dq.device)
        s_colT = torch.empty((T // 32, W), dtype=torch.uint8, device=dq.device)
    args = (dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), cos.data_ptr(), sin.data_ptr(), _ptr(y_row), _ptr(s_row), _ptr(y_colT),
            _ptr(s_colT), T, seq, n_q, n_kv, head_dim, fmt, _stream())
    t = KernelTimer.active
    if t is None:
        rc = _lib.load().mi_mxfp8_rope_bwd_quantize(*args)
    else:
        with t.span("mxfp8_quantize", f"rope {T}x{W}", float(T * W), float(T * W * (2 + (1 + 1 / 32) * (int(rowwise) + int(colwise))))):
            rc = _lib.load().mi_mxfp8_rope_bwd_quantize(*args)
    _lib.check(rc, "mi_mxfp8_rope_bwd_quantize")
    return y_row, s_row, y_colT, s_colT


def gemm_fp8_grouped(problems, fmt_a: int, fmt_b: int, tile_cfg: int = -1) -> None:
    """ONE persistent launch for up to 4 GEMMs D = (A . B^T) * (sa_inv * sb_inv) (mi_gemm_fp8_grouped): `problems` is a list of
    (a8 [M,K], b8 [N,K], sa_inv, sb_inv, out bf16 [M,N]); all share the operand formats.  Outputs are written in place."""
    n = len(problems)
    arr = (_lib.GemmProblem * n)()
    for i, (a8, b8, sa, sb, out) in enumerate(problems):
        _dev(a8, b8, sa, sb, out)
        assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and out.dtype == torch.bfloat16
        assert a8.stride(1) == 1 and b8.stride(1) == 1 and out.stride(1) == 1 and a8.shape[1] == b8.shape[1]
        assert out.shape == (a8.shape[0], b8.shape[0])
        arr[i] = _lib.GemmProblem(a8.data_ptr(), b8.data_ptr(), out.data_ptr(), sa.data_ptr(), sb.data_ptr(), a8.shape[0], b8.shape[0],
                                  a8.shape[1], a8.stride(0), b8.stride(0), out.stride(0))
    t = KernelTimer.active
    if t is None:
        rc = _lib.load().mi_gemm_fp8_grouped(ctypes.byref(arr), n, fmt_a, fmt_b, tile_cfg, _stream())
    else:
        work = sum(2.0 * a.shape[0] * b.shape[0] * a.shape[1] for a, b, _, _, _ in problems)
        nbytes = sum(a.numel() + b.numel() + 2 * o.numel() for a, b, _, _, o in problems)
        tag = "+".join(f"{a.shape[0]}x{b.shape[0]}x{a.shape[1]}" for a, b, _, _, _ in problems)
        with t.span("gemm_fp8", tag, work, nbytes):
            rc = _lib.load().mi_gemm_fp8_grouped(ctypes.byref(arr), n, fmt_a, fmt_b, tile_cfg, _stream())
    _lib.check(rc, "mi_gemm_fp8_grouped")


def grouped_gemm_ok(shapes, strides_ok: bool = True) -> bool:
    """Can mi_gemm_fp8_grouped take these (M, N, K) problems together?  (one tile shape dividing all, K % 256, < 2 GiB operands)"""
    if not strides_ok or not shapes or len(shapes) > 4:
        return False
    if not any(all(M % bm == 0 and N % bn == 0 for M, N, K in shapes) for bm, bn in _TILE_CFGS):
        return False
    if any(K % 256 or M * K >= 2 ** 31 or N * K >= 2 ** 31 or M * N * 2 >= 2 ** 31 for M, N, K in shapes):
        return False
    return True


_TILE_CFGS = ((256, 256), (256, 192), (192, 256), (192, 192))
_TILE_EFF = (1.0, 0.90, 0.90, 0.80)   # relative MFMA-time efficiency of the shorter phases (mi_gemm.hip pick_tile_cfg)
_GROUP_PLAN: dict = {}


def grouped_gemm_plan(shapes, n_cu: int = 256) -> int:
    """Tile shape (0-3) with which ONE grouped launch of these (M, N, K) problems is predicted to beat separate launches, or -1.
    Model, in K-tile steps of a 256 x 256 tile: separate = sum over problems of rounds x K-tiles x area / efficiency at the
    problem's best tile shape, + a fixed cost per launch (ramp + exposed last epilogue ~ 4 steps); grouped = the makespan of the
    longest-processing-time schedule the library builds (same greedy, here on counts), x 1.02 for the slightly heavier kernel,
    + one fixed cost.  Calibrated on tools/bench_kernels.py --which grouped (MI355X): grouped wins where the short problem fills
    the long one's last round (Llama-3.2-3B fc1 / fc2 / o-proj / lm_head backward: +4..10 %), not where the tile counts
    already divide the chip (3B q|k|v, 8B o-proj: -6 %)."""
    key = (tuple(shapes), n_cu)
    if key in _GROUP_PLAN:
        return _GROUP_PLAN[key]
    import heapq
    best = -1
    if grouped_gemm_ok(shapes):
        fixed = 4.0
        sep = 0.0
        for M, N, K in shapes:
            c = min((-(-((M // bm) * (N // bn)) // n_cu)) * (K // 128) * (bm * bn / 65536.0) / eff
                    for (bm, bn), eff in zip(_TILE_CFGS, _TILE_EFF) if M % bm == 0 and N % bn == 0)
            sep += c + fixed
        best_cost = None
        for cfg, ((bm, bn), eff) in enumerate(zip(_TILE_CFGS, _TILE_EFF)):
            if any(M % bm or N % bn for M, N, K in shapes):
                continue
            probs = sorted(((K // 128, (M // bm) * (N // bn)) for M, N, K in shapes), reverse=True)
            total = sum(t for _, t in probs)
            g = min(total, n_cu)
            loads = [(0, v) for v in range(g)]
            heapq.heapify(loads)
            per_wg = [0] * g
            for nk, tiles in probs:
                for _ in range(tiles):
                    load, v = heapq.heappop(loads)
                    per_wg[v] += 1
                    heapq.heappush(loads, (load + nk, v))
            if max(per_wg) > 64:
                continue
            cost = max(l for l, _ in loads) * (bm * bn / 65536.0) / eff * 1.02 + fixed
            if best_cost is None or cost < best_cost:
                best_cost, best = cost, cfg
        if best_cost is None or best_cost >= 0.98 * sep:
            best = -1
    _GROUP_PLAN[key] = best
    return best


_GROUP_TUNED: dict = {}


def grouped_gemm_choice(problems, fmt_a: int, fmt_b: int) -> int:
    """-1 = launch a Linear's dgrad and wgrad separately, 0-3 = one grouped launch with that tile shape.  Policy
    (env LLM_FP8_AMD_GROUPED_GEMM):
      auto (default)  single process: `autotune` -- MEASURED once per (shapes, formats) the first time the shape set shows up, i.e.
                      inside the first backward of a run: ~7 timed relaunches per candidate on SCRATCH outputs and one host
                      synchronisation per candidate; every later step only reads the cache (no host sync).  Under torch.distributed:
                      `plan`, so that every rank takes the same decision without timing anything between collectives.
      plan            the count model (grouped_gemm_plan): no timing, no host sync, box-independent.
      autotune        always measure (also under torch.distributed).
      off             always two launches (same as LLM_FP8_AMD_NO_GROUPED_GEMM=1).
    All choices give bitwise identical results."""
    mode = os.environ.get("LLM_FP8_AMD_GROUPED_GEMM", "auto")
    if mode == "off":
        return -1
    if mode == "auto":
        import torch.distributed as dist
        mode = "plan" if (dist.is_available() and dist.is_initialized()) else "autotune"
    if mode == "plan":
        return grouped_gemm_plan(tuple((a.shape[0], b.shape[0], a.shape[1]) for a, b, _, _, _ in problems))
    if mode != "autotune":
        raise ValueError(f"LLM_FP8_AMD_GROUPED_GEMM={mode!r}: expected auto, plan, autotune or off")
    return grouped_gemm_autotune(problems, fmt_a, fmt_b)


def grouped_gemm_autotune(problems, fmt_a: int, fmt_b: int, iters: int = 5) -> int:
    """Measured choice for a recurring group of GEMMs (a Linear's dgrad + wgrad): -1 = separate launches, 0-3 = one grouped launch
    with that tile shape, 4 = one grouped launch of 256 x 256 tiles on the four-wave kernel.  Timed once per (shapes, formats) on the operands at hand but into SCRATCH outputs (the live dX buffer and
    gradient-arena slot are not touched), and cached.  The model (grouped_gemm_plan) ranks the same candidates from counts alone;
    the measurement also sees what the model leaves out (per-tile epilogue cost, L2 behaviour, clock)."""
    shapes = tuple((a.shape[0], b.shape[0], a.shape[1]) for a, b, _, _, _ in problems)
    key = (shapes, fmt_a, fmt_b)
    hit = _GROUP_TUNED.get(key)
    if hit is not None:
        return hit
    problems = [(a8, b8, sa, sb, torch.empty_like(out)) for a8, b8, sa, sb, out in problems]
    cands = {-1: None}
    if grouped_gemm_ok(shapes):
        for cfg, (bm, bn) in enumerate(_TILE_CFGS):
            if all(M % bm == 0 and N % bn == 0 for M, N, K in shapes):
                cands[cfg] = cfg
        if all(M % 256 == 0 and N % 256 == 0 and K >= 512 for M, N, K in shapes) and os.environ.get("LLM_FP8_AMD_GEMM_W4G", "1") != "0":
            cands[4] = 4  # 256 x 256 tiles on the four-wave kernel (mi_gemm_w4.hip)

    def run(c):
        if c == -1:
            for a8, b8, sa, sb, out in problems:
                gemm_fp8(a8, b8, sa, sb, fmt_a, fmt_b, out=out)  # the default algo: what two separate launches would really run
        else:
            gemm_fp8_grouped(problems, fmt_a, fmt_b, tile_cfg=c)

    saved, KernelTimer.active = KernelTimer.active, None  # (the candidates are not part of any timed span)
    try:
        live = []
        for c in cands:
            try:
                run(c)  # warm-up; a candidate the library refuses drops out
                live.append(c)
            except RuntimeError:
                continue
        # interleaved rounds (candidate order rotates), not one block per candidate: the chip's clock drifts under load and a
        # block-wise comparison favours whoever ran first
        total = {c: 0.0 for c in live}
        for r in range(iters):
            order = live[r % len(live):] + live[:r % len(live)]
            for c in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(c)
                run(c)
                e1.record()
                e1.synchronize()
                total[c] += e0.elapsed_time(e1)
        best, best_t = -1, total.get(-1)
        for c in live:
            if c == -1:
                continue
            t = total[c]
            if best_t is None or t < best_t * (0.98 if best == -1 else 1.0):  # a grouped launch must win by 2 % over separate ones
                best, best_t = c, t
        # the four-wave grouped form is 3-5 % ahead of the eight-wave one on every 3B site in a quiet timing loop
        # (profiles/r03_grouped_w4_ab.txt); inside a first backward the five rounds above still carry ~1 % of noise: it takes ties
        if 4 in total and best not in (-1, 4) and total[4] <= 1.01 * best_t:
            best, best_t = 4, total[4]
    finally:
        KernelTimer.active = saved
    _GROUP_TUNED[key] = best
    return best


def transpose_u8(y: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FP8 bytes [R, C] -> [C, R] (mi_transpose_u8); rows / cols multiples of 8; `out` may be a strided [C, R] view."""
    _dev(y)
    assert y.dtype == torch.uint8 and y.dim() == 2 and y.stride(1) == 1
    R, C = y.shape
    if out is None:
        out = torch.empty((C, R), dtype=torch.uint8, device=y.device)
    assert out.shape == (C, R) and out.dtype == torch.uint8 and out.stride(1) == 1
    _lib.check(_lib.load().mi_transpose_u8(y.data_ptr(), out.data_ptr(), R, C, y.stride(0), out.stride(0), _stream()), "mi_transpose_u8")
    return out


def embedding_grad_add_(grad: torch.Tensor, dy: torch.Tensor, ids: torch.Tensor, alpha: float = 1.0, padding_idx: int = -1) -> None:
    """grad [V, H] bf16 += alpha * scatter-sum of dy [T, H] bf16 rows by ids [T] (int64), in place, deterministic."""
    _dev(grad, dy, ids)
    assert grad.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and grad.is_contiguous() and grad.dim() == 2
    dy = dy.reshape(-1, dy.shape[-1])
    dy = dy if dy.is_contiguous() else dy.contiguous()
    ids = ids.reshape(-1).to(torch.int64)
    assert dy.shape == (ids.numel(), grad.shape[1])
    sorted_ids, perm = torch.sort(ids, stable=True)
    rc = _lib.load().mi_embedding_grad_add(grad.data_ptr(), dy.data_ptr(), sorted_ids.data_ptr(), perm.data_ptr(), ids.numel(),
                                           grad.shape[1], grad.shape[0], float(alpha), int(padding_idx), _stream())
    _lib.check(rc, "mi_embedding_grad_add")


def rope_qkv_backward_cast(dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                           n_q: int, n_kv: int, head_dim: int, seq: int, scale: torch.Tensor, amax: Optional[torch.Tensor],
                           fmt: int, want_y: bool = True, want_t: bool = True):
    """rope_qkv_backward + cast_amax of its result in one launch: (g8 [T, W], g8T [W, T]); the bf16 d(qkv) is never written."""
    _dev(dq, dk, dv, cos, sin, scale, amax)
    dq, dk, dv = (t.contiguous() for t in (dq, dk, dv))
    T = dq.shape[0]
    W = (n_q + 2 * n_kv) * head_dim
    y = torch.empty((T, W), dtype=torch.uint8, device=dq.device) if want_y else None
    yT = torch.empty((W, T), dtype=torch.uint8, device=dq.device) if want_t else None
    args = (dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), cos.data_ptr(), sin.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(),
            _ptr(amax), T, seq, n_q, n_kv, head_dim, fmt, _stream())
    t = KernelTimer.active
    if t is None:
        rc = _lib.load().mi_rope_qkv_bwd_cast(*args)
    else:
        with t.span("cast_amax", f"rope {T}x{W}", float(T * W), float(T * W * (2 + (y is not None) + (yT is not None)))):
            rc = _lib.load().mi_rope_qkv_bwd_cast(*args)
    _lib.check(rc, "mi_rope_qkv_bwd_cast")
    return y, yT


def swiglu_cast(h: torch.Tensor, scale: torch.Tensor, amax: Optional[torch.Tensor], fmt: int, want_y: bool = True,
                want_t: bool = True, bias: Optional[torch.Tensor] = None):
    """K10 fwd.  h bf16 [R, 2F] -> (act8 [R, F], act8T [F, R]) with act = silu(gate) * up computed in fp32.
    `bias` (bf16 [2F]): h is the fc1 output WITHOUT bias; the kernel adds it (fp32) before the activation."""
    _dev(h, scale, amax, bias)
    assert h.dtype == torch.bfloat16 and h.dim() == 2 and h.is_contiguous() and h.shape[1] % 2 == 0
    R, F2 = h.shape
    F = F2 // 2
    y = torch.empty((R, F), dtype=torch.uint8, device=h.device) if want_y else None
    yT = torch.empty((F, R), dtype=torch.uint8, device=h.device) if want_t else None
    if bias is None:
        fn, args = _lib.load().mi_swiglu_cast, (h.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(), _ptr(amax), R, F, fmt, _stream())
    else:
        assert bias.dtype == torch.bfloat16 and bias.numel() == F2 and bias.is_contiguous()
        fn, args = _lib.load().mi_swiglu_cast_bias, (h.data_ptr(), bias.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(), _ptr(amax), R, F, fmt, _stream())
    t = KernelTimer.active
    if t is None:
        rc = fn(*args)
    else:
        with t.span("swiglu_cast", f"{R}x{F}", float(R * F), float(R * F * (4 + int(want_y) + int(want_t)))):
            rc = fn(*args)
    _lib.check(rc, "mi_swiglu_cast")
    return y, yT


def dswiglu_cast(h: torch.Tensor, dact: torch.Tensor, scale: torch.Tensor, amax: Optional[torch.Tensor], fmt: int,
                 want_y: bool = True, want_t: bool = True, want_colsum: bool = False, bias: Optional[torch.Tensor] = None):
    """K10 bwd.  -> (dh8 [R, 2F], dh8T [2F, R], colsum fp32 [ceil(R/128), 2F] or None).  `bias`: as swiglu_cast."""
    _dev(h, dact, scale, amax, bias)
    assert h.dtype == torch.bfloat16 and dact.dtype == torch.bfloat16 and h.is_contiguous() and dact.is_contiguous()
    R, F2 = h.shape
    F = F2 // 2
    assert dact.shape == (R, F)
    y = torch.empty((R, F2), dtype=torch.uint8, device=h.device) if want_y else None
    yT = torch.empty((F2, R), dtype=torch.uint8, device=h.device) if want_t else None
    cs = torch.empty(((R + 127) // 128, F2), dtype=torch.float32, device=h.device) if want_colsum else None
    if bias is None:
        fn = _lib.load().mi_dswiglu_cast
        args = (h.data_ptr(), dact.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(), _ptr(amax), _ptr(cs), R, F, fmt, _stream())
    else:
        assert bias.dtype == torch.bfloat16 and bias.numel() == F2 and bias.is_contiguous()
        fn = _lib.load().mi_dswiglu_cast_bias
        args = (h.data_ptr(), bias.data_ptr(), dact.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(), _ptr(amax), _ptr(cs), R, F, fmt, _stream())
    t = KernelTimer.active
    if t is None:
        rc = fn(*args)
    else:
        with t.span("dswiglu_cast", f"{R}x{F}", float(R * F2), float(R * F * (6 + 2 * int(want_y) + 2 * int(want_t)))):
            rc = fn(*args)
    _lib.check(rc, "mi_dswiglu_cast")
    return y, yT, cs


def rmsnorm_stats(x: torch.Tensor, eps: float) -> torch.Tensor:
    """K9: rstd[r] = rsqrt(mean(x[r]^2) + eps), fp32 [R]."""
    _dev(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
    R, C = x.shape
    rstd = torch.empty(R, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mi_rmsnorm_stats(x.data_ptr(), rstd.data_ptr(), R, C, float(eps), _stream()), "mi_rmsnorm_stats")
    return rstd


def add_rmsnorm_stats(a: torch.Tensor, b: torch.Tensor, eps: float, bias: Optional[torch.Tensor] = None):
    """(a + b [same shape, bf16], rstd fp32 [rows]) in one pass; rows = all leading dims, statistics over the last dim.
    `bias` (bf16 [cols]): out = a + (b + bias) -- b is a GEMM output whose bias was left for this add."""
    _dev(a, b, bias)
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    C = a.shape[-1]
    R = a.numel() // C
    out = torch.empty_like(a)
    rstd = torch.empty(R, dtype=torch.float32, device=a.device)
    if bias is None:
        rc = _lib.load().mi_add_rmsnorm_stats(a.data_ptr(), b.data_ptr(), out.data_ptr(), rstd.data_ptr(), R, C, float(eps), _stream())
    else:
        assert bias.dtype == torch.bfloat16 and bias.numel() == C and bias.is_contiguous()
        rc = _lib.load().mi_add_bias_rmsnorm_stats(a.data_ptr(), b.data_ptr(), bias.data_ptr(), out.data_ptr(), rstd.data_ptr(), R, C,
                                                   float(eps), _stream())
    _lib.check(rc, "mi_add_rmsnorm_stats")
    return out, rstd


def norm_cast(x: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor, scale: torch.Tensor, amax: Optional[torch.Tensor],
              fmt: int, want_y: bool = True, want_t: bool = True):
    """K9: (x * rstd[:, None]) * gamma in fp32 -> (y8 [R, C], y8T [C, R]) + amax; no bf16 normalised tensor is written."""
    _dev(x, rstd, gamma, scale, amax)
    assert x.dtype == torch.bfloat16 and gamma.dtype == torch.bfloat16 and x.is_contiguous() and gamma.is_contiguous()
    R, C = x.shape
    y = torch.empty((R, C), dtype=torch.uint8, device=x.device) if want_y else None
    yT = torch.empty((C, R), dtype=torch.uint8, device=x.device) if want_t else None
    args = (x.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), _ptr(y), _ptr(yT), scale.data_ptr(), _ptr(amax), R, C, fmt, _stream())
    t = KernelTimer.active
    if t is None:
        rc = _lib.load().mi_norm_cast(*args)
    else:
        with t.span("norm_cast", f"{R}x{C}", float(R * C), float(R * C * (2 + int(want_y) + int(want_t)))):
            rc = _lib.load().mi_norm_cast(*args)
    _lib.check(rc, "mi_norm_cast")
    return y, yT


def rmsnorm_bwd(dy: torch.Tensor, x: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor,
                dres: Optional[torch.Tensor] = None, n_partials: int = 512, dgamma_dtype: torch.dtype = torch.float32,
                finish: bool = True):
    """K9 backward: (dx bf16 [R, C], dgamma [C] in `dgamma_dtype`).  dgamma is the fixed-order sum of per-block partials;
    `dres` (bf16 [R, C]) is added to dx (the gradient arriving over the residual connection)."""
    _dev(dy, x, rstd, gamma, dres)
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous()
    R, C = x.shape
    n_partials = max(1, min(n_partials, (R + 3) // 4))
    dx = torch.empty_like(x)
    part = torch.empty((n_partials, C), dtype=torch.float32, device=x.device)
    rc = _lib.load().mi_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), _ptr(dres), dx.data_ptr(),
                                    part.data_ptr(), n_partials, R, C, _stream())
    _lib.check(rc, "mi_rmsnorm_bwd")
    return (dx, colsum_finish(part, dgamma_dtype)) if finish else (dx, part)


def _mx_alloc(R: int, C: int, dev, rowwise: bool, colwise: bool):
    y_row = s_row = y_colT = s_colT = None
    if rowwise:
        y_row = torch.empty((R, C), dtype=torch.uint8, device=dev)
        s_row = torch.empty((C // 32, R), dtype=torch.uint8, device=dev)
    if colwise:
        y_colT = torch.empty((C, R), dtype=torch.uint8, device=dev)
        s_colT = torch.empty((R // 32, C), dtype=torch.uint8, device=dev)
    return y_row, s_row, y_colT, s_colT


def _mx_call(kind: str, fn, args, R: int, C: int, in_bytes: float, rowwise: bool, colwise: bool):
    t = KernelTimer.active
    if t is None:
        rc = fn(*args)
    else:
        with t.span("mxfp8_quantize", f"{kind} {R}x{C}", float(R * C), in_bytes + R * C * (1 + 1 / 32) * (int(rowwise) + int(colwise))):
            rc = fn(*args)
    _lib.check(rc, f"mi_mxfp8_{kind}_quantize")


def mxfp8_norm_quantize(x: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor, fmt: int = E4M3, rowwise: bool = True,
                        colwise: bool = True):
    """K9 for MXFP8: quantises (x * rstd[:, None]) * gamma without materialising it.  Same outputs as mxfp8_quantize."""
    _dev(x, rstd, gamma)
    assert x.dtype == torch.bfloat16 and gamma.dtype == torch.bfloat16 and x.is_contiguous() and gamma.is_contiguous()
    R, C = x.shape
    out = _mx_alloc(R, C, x.device, rowwise, colwise)
    _mx_call("norm", _lib.load().mi_mxfp8_norm_quantize,
             (x.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), *[_ptr(o) for o in out], R, C, fmt, _stream()), R, C, 2.0 * R * C,
             rowwise, colwise)
    return out


def mxfp8_swiglu_quantize(h: torch.Tensor, fmt: int = E4M3, rowwise: bool = True, colwise: bool = True):
    """K10 for MXFP8: quantises silu(h[:, :F]) * h[:, F:] ([R, F])."""
    _dev(h)
    assert h.dtype == torch.bfloat16 and h.is_contiguous() and h.shape[1] % 2 == 0
    R, F = h.shape[0], h.shape[1] // 2
    out = _mx_alloc(R, F, h.device, rowwise, colwise)
    _mx_call("swiglu", _lib.load().mi_mxfp8_swiglu_quantize, (h.data_ptr(), *[_ptr(o) for o in out], R, F, fmt, _stream()), R, F,
             4.0 * R * F, rowwise, colwise)
    return out


def mxfp8_dswiglu_quantize(h: torch.Tensor, dact: torch.Tensor, fmt: int = E4M3, rowwise: bool = True, colwise: bool = True,
                           want_colsum: bool = False):
    """K10 backward for MXFP8: quantises dh = [dact*dsilu(g)*u | dact*silu(g)] ([R, 2F]); returns (..., colsum or None)."""
    _dev(h, dact)
    assert h.dtype == torch.bfloat16 and dact.dtype == torch.bfloat16 and h.is_contiguous() and dact.is_contiguous()
    R, F2 = h.shape
    F = F2 // 2
    assert dact.shape == (R, F)
    out = _mx_alloc(R, F2, h.device, rowwise, colwise)
    cs = torch.empty(((R + 127) // 128, F2), dtype=torch.float32, device=h.device) if want_colsum else None
    _mx_call("dswiglu", _lib.load().mi_mxfp8_dswiglu_quantize,
             (h.data_ptr(), dact.data_ptr(), *[_ptr(o) for o in out], _ptr(cs), R, F, fmt, _stream()), R, F2, 6.0 * R * F,
             rowwise, colwise)
    return (*out, cs)


def _bshd_ok(t: torch.Tensor, D: int):
    assert t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[3] == D and t.stride(3) == 1 and t.stride(2) == D
    assert t.stride(0) == t.shape[1] * t.stride(1), "batch and sequence must collapse to one token stride"


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, causal: bool = True):
    """Attention core.  q [B, S, H, D], k/v [B, S, G, D] bf16 (views with a token stride are fine) -> (o [B, S, H, D], lse [B, H, S])."""
    _dev(q, k, v)
    B, S, H, D = q.shape
    G = k.shape[2]
    for t in (q, k, v):
        _bshd_ok(t, D)
    o = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, S, H, G, D, q.stride(1), k.stride(1),
            v.stride(1), o.stride(1), float(scale), int(causal), _stream())
    t = KernelTimer.active
    flops = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    if t is None:
        rc = _lib.load().mi_attn_fwd(*args)
    else:
        with t.span("attn_fwd", f"{B}x{S}x{H}x{D}", flops, 2.0 * (q.numel() * 2 + k.numel() * 2)):
            rc = _lib.load().mi_attn_fwd(*args)
    _lib.check(rc, "mi_attn_fwd")
    return o, lse


def attn_bwd(do: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, lse: torch.Tensor,
             scale: float, causal: bool = True, out=None):
    """Backward of attn_fwd -> (dq [B,S,H,D], dk, dv [B,S,G,D]).  `out` = optional preallocated (dq, dk, dv) views."""
    _dev(do, q, k, v, o, lse)
    B, S, H, D = q.shape
    G = k.shape[2]
    for t in (q, k, v, o, do):
        _bshd_ok(t, D)
    if out is None:
        dq = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=q.device)
        dk = torch.empty((B, S, G, D), dtype=torch.bfloat16, device=q.device)
        dv = torch.empty((B, S, G, D), dtype=torch.bfloat16, device=q.device)
    else:
        dq, dk, dv = out
        for t in out:
            _bshd_ok(t, D)
    delta = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, S, H, G, D, q.stride(1), k.stride(1), v.stride(1), o.stride(1),
            do.stride(1), dq.stride(1), dk.stride(1), dv.stride(1), float(scale), int(causal), _stream())
    t = KernelTimer.active
    flops = 14.0 * B * H * S * S * D * (0.5 if causal else 1.0)  # 7 products: S, dP, dQ | S, dP, dV, dK
    if t is None:
        rc = _lib.load().mi_attn_bwd(*args)
    else:
        with t.span("attn_bwd", f"{B}x{S}x{H}x{D}", flops, 2.0 * (3 * q.numel() + 4 * k.numel())):
            rc = _lib.load().mi_attn_bwd(*args)
    _lib.check(rc, "mi_attn_bwd")
    return dq, dk, dv
