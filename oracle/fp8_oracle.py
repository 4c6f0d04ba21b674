"""CPU oracle for the FP8 Linear hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``llm_fp8_amd/`` imports it and the
product path has no CPU fallback.

PARITY UNPINNED.  The arithmetic this file restates lives in a third-party
dependency of the reference that is absent from ``/root/reference`` and from
this container: NVIDIA Transformer Engine (``transformer_engine[pytorch]``,
unpinned in ``install.sh:4``; "Transformer Engine 2.5.0" per
``paper/conference_101719.tex:236``).  The reference holds no test, golden
vector or fixture for this path.  What is restated
here is the published OCP FP8 (E4M3FN / E5M2) and OCP MX (E8M0 block scale)
formats plus TE's delayed-scaling recipe as summarised in SURVEY.md Appendix A,
anchored on the reference's call sites:

* recipes        te_llama.py:39-40, te_llama_hybrid.py:39, te_llama_mxfp8.py:28-29
* module layout  te_llama.py:41-82 (MultiheadAttention + LayerNormMLP)
* weight layout  te_llama.py:181-239 (replace_params: q|k|v, gate|up order)
* outer recipe   train_fp8.py:126-165, accelerate utils/transformer_engine.py:118-186

The fp8 encoders are written with integer/bit arithmetic only (numpy) and are
cross-checked in ``tests/test_oracle.py`` against torch's independent
``float8_e4m3fn`` / ``float8_e5m2`` casts and hand-derived known answers.
"""
from __future__ import annotations

import numpy as np

E4M3 = 0  # OCP E4M3FN: bias 7, max 448, no inf, NaN = S.1111.111
E5M2 = 1  # OCP E5M2  : bias 15, max 57344, inf/NaN IEEE-like

FP8_MAX = {E4M3: np.float32(448.0), E5M2: np.float32(57344.0)}
_MBITS = {E4M3: 3, E5M2: 2}
_BIAS = {E4M3: 7, E5M2: 15}
FP8_NAN_BYTE = 0x7F  # canonical NaN byte produced by the quantiser for either format


# --------------------------------------------------------------------------- bf16 helpers
def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float32 (exact)."""
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round-to-nearest-even, NaN stays NaN (quiet)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if np.any(nan):
        rounded = np.where(nan, ((u >> 16) | 0x40).astype(np.uint16), rounded)
    return rounded


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# --------------------------------------------------------------------------- fp8 codec
def fp8_decode_table(fmt: int) -> np.ndarray:
    """256-entry float32 table byte -> value (OCP FP8)."""
    m, bias = _MBITS[fmt], _BIAS[fmt]
    out = np.zeros(256, dtype=np.float64)
    for b in range(256):
        s = -1.0 if b & 0x80 else 1.0
        e = (b & 0x7F) >> m
        f = b & ((1 << m) - 1)
        if fmt == E4M3 and (b & 0x7F) == 0x7F:
            v = np.nan
        elif fmt == E5M2 and e == 31:
            v = np.inf if f == 0 else np.nan
        elif e == 0:
            v = f * 2.0 ** (1 - bias - m)
        else:
            v = (1 + f / (1 << m)) * 2.0 ** (e - bias)
        out[b] = s * v
    return out.astype(np.float32)


_DEC = {E4M3: fp8_decode_table(E4M3), E5M2: fp8_decode_table(E5M2)}


def fp8_decode(bytes_: np.ndarray, fmt: int) -> np.ndarray:
    return _DEC[fmt][np.asarray(bytes_, dtype=np.uint8)]


def fp8_encode_sat(v: np.ndarray, fmt: int) -> np.ndarray:
    """float32 -> fp8 byte: saturate to +-max (inf included), then round-to-nearest-even.

    NaN -> 0x7F (sign dropped) for both formats.  -0.0 -> 0x80.  Works on the
    float32 bit pattern with integer arithmetic only.
    """
    v = np.ascontiguousarray(v, dtype=np.float32)
    m, bias = _MBITS[fmt], _BIAS[fmt]
    u = v.view(np.uint32).astype(np.int64)
    sign = ((u >> 31) & 1).astype(np.int64)
    mag = u & 0x7FFFFFFF
    isnan = mag > 0x7F800000
    max_bits = np.int64(np.float32(FP8_MAX[fmt]).view(np.uint32))
    mag = np.minimum(mag, max_bits)  # saturate (covers inf); NaN fixed up below
    e32 = mag >> 23  # biased fp32 exponent
    man = mag & 0x7FFFFF
    # target exponent (unbiased) and significand with hidden bit
    e_unb = e32 - 127
    sig = np.where(e32 > 0, man | 0x800000, man)  # 24-bit significand; fp32 subnormals: no hidden bit
    e_unb = np.where(e32 > 0, e_unb, -126)
    emin = 1 - bias  # exponent of min normal
    # number of fp32 fraction bits to drop: 23 - m for normals, more below emin
    drop = (23 - m) + np.maximum(emin - e_unb, 0)
    drop = np.minimum(drop, 40)
    half = np.int64(1) << (drop - 1)
    mask = (np.int64(1) << drop) - 1
    q = sig >> drop
    rem = sig & mask
    q = q + ((rem > half) | ((rem == half) & ((q & 1) == 1))).astype(np.int64)
    # q is in units of 2^(max(e_unb,emin) - m).  Build the byte arithmetically:
    # normal: q in [2^m, 2^(m+1)] ; byte = ((e_unb + bias - 1) << m) + q handles mantissa carry.
    e_eff = np.maximum(e_unb, emin)
    byte = ((e_eff + bias - 1) << m) + q
    byte = np.where(e_unb < emin, q, byte)  # subnormal / rounds-up-to-min-normal: exponent field comes from q
    byte = np.where(mag == 0, 0, byte)
    byte = byte | (sign << 7)
    byte = np.where(isnan, FP8_NAN_BYTE, byte)
    return byte.astype(np.uint8)


# --------------------------------------------------------------------------- delayed scaling (K1/K2/K3)
def amax_f32(x: np.ndarray) -> np.float32:
    """max(|x|) with fmaxf semantics: NaN elements are ignored, +-inf propagates; empty -> 0."""
    a = np.abs(np.asarray(x, dtype=np.float32)).ravel()
    a = a[~np.isnan(a)]
    return np.float32(a.max()) if a.size else np.float32(0.0)


def quantize_delayed(x_bf16_bits: np.ndarray, scale: np.float32, fmt: int):
    """K1: y = sat_cast(x_f32 * scale) RNE; amax over the *unscaled* input.

    x is given as bf16 bit patterns [rows, cols]; returns (bytes[rows, cols], amax f32).
    The product x*scale is a single IEEE float32 multiply.
    """
    x = bf16_bits_to_f32(x_bf16_bits)
    with np.errstate(over="ignore", invalid="ignore"):
        y = x * np.float32(scale)
    return fp8_encode_sat(y, fmt), amax_f32(x)


def quantize_delayed_transpose(x_bf16_bits, scale, fmt):
    """K2: K1 plus the transposed fp8 copy used by the dgrad / wgrad GEMMs."""
    q, amax = quantize_delayed(x_bf16_bits, scale, fmt)
    return q, np.ascontiguousarray(q.T), amax


FLT_MAX = np.finfo(np.float32).max


def scale_update(amax_history: np.ndarray, scale: np.ndarray, fp8_max, margin: int = 0,
                 algo: str = "max"):
    """K3: TE fused amax-and-scale update, float32 arithmetic.

    amax_history [H, S] (row 0 = this iteration's amax), scale [S]; fp8_max scalar or [S].
    Returns (new_history, new_scale, new_scale_inv).
    """
    h = np.array(amax_history, dtype=np.float32, copy=True)
    scale = np.asarray(scale, dtype=np.float32)
    fp8_max = np.broadcast_to(np.asarray(fp8_max, dtype=np.float32), scale.shape)
    if algo == "max":
        with np.errstate(invalid="ignore"):
            # fmaxf semantics: ignore NaN unless every entry is NaN
            amax = np.where(np.all(np.isnan(h), axis=0), np.float32(np.nan),
                            np.nanmax(np.where(np.isnan(h), -np.inf, h), axis=0)).astype(np.float32)
    elif algo == "most_recent":
        amax = h[0].copy()
    else:
        raise ValueError(algo)
    h = np.roll(h, -1, axis=0)
    h[0, :] = 0.0
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        sf = (fp8_max / amax).astype(np.float32)
        sf = (sf / np.float32(2.0 ** margin)).astype(np.float32)
    keep = ~(amax > 0) | ~np.isfinite(amax)
    sf = np.where(np.isinf(sf), np.float32(FLT_MAX), sf)
    sf = np.where(keep, scale, sf).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = (np.float32(1.0) / sf).astype(np.float32)
    return h, sf, inv


# --------------------------------------------------------------------------- FP8 GEMM (K4-K6)
def gemm_fp8_tn(a8, b8, fmt_a, fmt_b, sa_inv, sb_inv, bias_bf16_bits=None, out_f32=False):
    """D[M,N] = (A[M,K] . B[N,K]^T) * (sa_inv*sb_inv) (+bias) -> bf16 bits (or f32).

    Accumulation in float64 (products of two fp8 values are exact; the sum is the
    infinitely-precise reference the fp32-accumulating kernel is compared against).
    alpha = sa_inv*sb_inv is one float32 product, as in the kernel epilogue.
    """
    a = fp8_decode(a8, fmt_a).astype(np.float64)
    b = fp8_decode(b8, fmt_b).astype(np.float64)
    alpha = np.float32(sa_inv) * np.float32(sb_inv)
    d = (a @ b.T) * np.float64(alpha)
    if bias_bf16_bits is not None:
        d = d + bf16_bits_to_f32(bias_bf16_bits).astype(np.float64)[None, :]
    d = d.astype(np.float32)
    return d if out_f32 else f32_to_bf16_bits(d)


def gemm_tolerance(ref_f32: np.ndarray) -> np.ndarray:
    """|delta| <= 2^-7 |ref| + 1e-3 rms(ref)  (SURVEY 8c): 1 bf16 ulp + fp32 accumulation-order noise."""
    ref = np.asarray(ref_f32, dtype=np.float64)
    rms = np.sqrt(np.mean(ref * ref)) if ref.size else 0.0
    return 2.0 ** -7 * np.abs(ref) + 1e-3 * rms


# --------------------------------------------------------------------------- MXFP8 (K7/K8)
MX_BLOCK = 32
_MAX_NORM_RCP = {E4M3: np.float32(1.0) / np.float32(448.0), E5M2: np.float32(1.0) / np.float32(57344.0)}


def float_to_e8m0_roundup(val: np.ndarray) -> np.ndarray:
    """E8M0 biased exponent of `val`, rounded UP to the next power of two (TE MXFP8 rule).

    NaN -> 0xFF, inf -> 0xFE, 0 -> 0.  For fp32 subnormal inputs (exponent field 0) the
    exponent is bumped only if the mantissa exceeds 0x400000 (i.e. val > 2^-127).
    """
    val = np.ascontiguousarray(val, dtype=np.float32)
    u = val.view(np.uint32)
    exp = ((u >> 23) & 0xFF).astype(np.int64)
    man = (u & 0x7FFFFF).astype(np.int64)
    bump = (man > 0) & (exp != 0xFE) & ~((exp == 0) & (man <= 0x400000))
    out = exp + bump.astype(np.int64)
    out = np.where(np.isnan(val), 0xFF, out)
    out = np.where(np.isinf(val), 0xFE, out)
    out = np.where(val == 0, 0, out)
    return out.astype(np.uint8)


def e8m0_to_f32(e: np.ndarray) -> np.ndarray:
    e = np.asarray(e, dtype=np.uint8).astype(np.int64)
    with np.errstate(over="ignore"):
        v = np.ldexp(np.float64(1.0), e - 127).astype(np.float32)
    return np.where(e == 0xFF, np.float32(np.nan), v)


def mxfp8_quantize_rowwise(x_bf16_bits: np.ndarray, fmt: int = E4M3):
    """K7: one E8M0 scale per 32 consecutive elements of the LAST axis.

    Returns (fp8 bytes [R, C], e8m0 bytes [R, C/32]).  y = sat_cast(x * 2^(127-e)).
    """
    x = bf16_bits_to_f32(x_bf16_bits)
    r, c = x.shape
    assert c % MX_BLOCK == 0
    xb = x.reshape(r, c // MX_BLOCK, MX_BLOCK)
    ab = np.abs(xb)
    ab = np.where(np.isnan(ab), np.float32(0), ab)  # fmaxf semantics
    amax_b = ab.max(axis=2).astype(np.float32)
    e = float_to_e8m0_roundup((amax_b * _MAX_NORM_RCP[fmt]).astype(np.float32))
    with np.errstate(over="ignore", invalid="ignore"):
        inv = np.ldexp(np.float32(1.0), 127 - e.astype(np.int64)).astype(np.float32)  # 2^(127-e), exact
        y = (xb * inv[:, :, None]).astype(np.float32)
    return fp8_encode_sat(y, fmt).reshape(r, c), e


def mxfp8_quantize_colwise(x_bf16_bits: np.ndarray, fmt: int = E4M3):
    """K7 column-wise copy: blocks of 32 along the FIRST axis, returned TRANSPOSED.

    Returns (fp8 bytes [C, R], e8m0 bytes [C, R/32]) i.e. the row-wise quantisation of x^T,
    which is the operand layout the TN GEMM consumes for dgrad / wgrad.
    """
    xt = np.ascontiguousarray(np.asarray(x_bf16_bits, dtype=np.uint16).T)
    return mxfp8_quantize_rowwise(xt, fmt)


def gemm_mxfp8_tn(a8, a_e8m0, b8, b_e8m0, fmt_a=E4M3, fmt_b=E4M3, bias_bf16_bits=None, out_f32=False):
    """K8: D = sum_blocks 2^(ea+eb-254) * sum_32 a8*b8 (+bias) -> bf16 bits."""
    a = fp8_decode(a8, fmt_a).astype(np.float64)
    b = fp8_decode(b8, fmt_b).astype(np.float64)
    sa = np.repeat(e8m0_to_f32(a_e8m0).astype(np.float64), MX_BLOCK, axis=1)
    sb = np.repeat(e8m0_to_f32(b_e8m0).astype(np.float64), MX_BLOCK, axis=1)
    d = (a * sa) @ (b * sb).T
    if bias_bf16_bits is not None:
        d = d + bf16_bits_to_f32(bias_bf16_bits).astype(np.float64)[None, :]
    d = d.astype(np.float32)
    return d if out_f32 else f32_to_bf16_bits(d)


# --------------------------------------------------------------------------- one FP8 Linear, fwd + bwd, with state
class DelayedLinearOracle:
    """One te.Linear under DelayedScaling: state carried across steps.

    fwd slots: 0 input, 1 weight, 2 output(unused);  bwd slots: 0 grad_output, 1 grad_input(unused).
    fmt_fwd/fmt_bwd follow recipe.Format: E4M3 -> (E4M3, E4M3); HYBRID -> (E4M3, E5M2).
    """

    def __init__(self, fmt_fwd=E4M3, fmt_bwd=E5M2, history_len=16, algo="max", margin=0):
        self.fmt_fwd, self.fmt_bwd = fmt_fwd, fmt_bwd
        self.algo, self.margin = algo, margin
        self.h_fwd = np.zeros((history_len, 3), np.float32)
        self.h_bwd = np.zeros((history_len, 2), np.float32)
        self.s_fwd = np.ones(3, np.float32)
        self.s_bwd = np.ones(2, np.float32)
        self.si_fwd = np.ones(3, np.float32)
        self.si_bwd = np.ones(2, np.float32)

    def forward(self, x_bits, w_bits, bias_bits=None):
        x8, x8t, ax = quantize_delayed_transpose(x_bits, self.s_fwd[0], self.fmt_fwd)
        w8, w8t, aw = quantize_delayed_transpose(w_bits, self.s_fwd[1], self.fmt_fwd)
        self.h_fwd[0, 0] = max(self.h_fwd[0, 0], ax)
        self.h_fwd[0, 1] = max(self.h_fwd[0, 1], aw)
        y = gemm_fp8_tn(x8, w8, self.fmt_fwd, self.fmt_fwd, self.si_fwd[0], self.si_fwd[1], bias_bits)
        self.saved = (x8t, w8t, self.si_fwd[0].copy(), self.si_fwd[1].copy())
        return y

    def end_forward(self):
        fmax = FP8_MAX[self.fmt_fwd]
        self.h_fwd, self.s_fwd, self.si_fwd = scale_update(self.h_fwd, self.s_fwd, fmax, self.margin, self.algo)

    def backward(self, dy_bits):
        x8t, w8t, six, siw = self.saved
        g8, g8t, ag = quantize_delayed_transpose(dy_bits, self.s_bwd[0], self.fmt_bwd)
        self.h_bwd[0, 0] = max(self.h_bwd[0, 0], ag)
        sig = self.si_bwd[0]
        dx = gemm_fp8_tn(g8, w8t, self.fmt_bwd, self.fmt_fwd, sig, siw)      # [M,N].[K,N]^T -> [M,K]
        dw = gemm_fp8_tn(g8t, x8t, self.fmt_bwd, self.fmt_fwd, sig, six)     # [N,M].[K,M]^T -> [N,K]
        db = f32_to_bf16_bits(bf16_bits_to_f32(dy_bits).astype(np.float64).sum(axis=0).astype(np.float32))
        return dx, dw, db

    def end_backward(self):
        fmax = FP8_MAX[self.fmt_bwd]
        self.h_bwd, self.s_bwd, self.si_bwd = scale_update(self.h_bwd, self.s_bwd, fmax, self.margin, self.algo)


def mxfp8_linear_fwd_bwd(x_bits, w_bits, dy_bits, bias_bits=None, fmt=E4M3):
    """One te.Linear under MXFP8BlockScaling(E4M3) (te_llama_mxfp8.py:28-29): stateless."""
    x8, xe = mxfp8_quantize_rowwise(x_bits, fmt)       # blocks along K
    w8, we = mxfp8_quantize_rowwise(w_bits, fmt)
    y = gemm_mxfp8_tn(x8, xe, w8, we, fmt, fmt, bias_bits)
    g8, ge = mxfp8_quantize_rowwise(dy_bits, fmt)      # [M,N], blocks along N  (dgrad contracts N)
    wt8, wte = mxfp8_quantize_colwise(w_bits, fmt)     # [K,N], blocks along N
    dx = gemm_mxfp8_tn(g8, ge, wt8, wte, fmt, fmt)
    gt8, gte = mxfp8_quantize_colwise(dy_bits, fmt)    # [N,M], blocks along M (wgrad contracts M)
    xt8, xte = mxfp8_quantize_colwise(x_bits, fmt)     # [K,M], blocks along M
    dw = gemm_mxfp8_tn(gt8, gte, xt8, xte, fmt, fmt)
    db = f32_to_bf16_bits(bf16_bits_to_f32(dy_bits).astype(np.float64).sum(axis=0).astype(np.float32))
    return y, dx, dw, db


# --------------------------------------------------------------------------- fused elementwise neighbours (K10, RoPE)
def swiglu_f32(h_bits: np.ndarray) -> np.ndarray:
    """act = silu(gate) * up in float32 from bf16 h = [gate | up] (te_llama.py:62, SURVEY App. A "SwiGLU")."""
    h = bf16_bits_to_f32(h_bits).astype(np.float64)
    f = h.shape[1] // 2
    g, u = h[:, :f], h[:, f:]
    with np.errstate(over="ignore"):
        return (g / (1.0 + np.exp(-g)) * u).astype(np.float32)


def dswiglu_f32(h_bits: np.ndarray, dact_bits: np.ndarray) -> np.ndarray:
    """dh = [dact * dsilu(g) * u | dact * silu(g)] in float32."""
    h = bf16_bits_to_f32(h_bits).astype(np.float64)
    d = bf16_bits_to_f32(dact_bits).astype(np.float64)
    f = h.shape[1] // 2
    g, u = h[:, :f], h[:, f:]
    with np.errstate(over="ignore"):
        s = 1.0 / (1.0 + np.exp(-g))
    return np.concatenate([d * u * (s * (1.0 + g * (1.0 - s))), d * (g * s)], axis=1).astype(np.float32)


# Device-order float32 restatements of the fused front ends (the kernels' arithmetic is
# llm_fp8_amd/csrc/mi_fused.hip: swiglu_cast_kernel, norm_cast_kernel, rmsnorm_stats_kernel).  Same operations in the same
# order, every one rounded to float32 (numpy float32 arithmetic is IEEE per operation; the kernels are compiled with
# -ffp-contract=off, so no FMA contraction).  What can still differ from the device is the transcendental alone: `__expf`
# (v_exp_f32 of x * log2 e) and `rsqrtf` (v_rsq_f32) are accurate to a few ulps, numpy's exp / sqrt to <= 1 ulp.  The FP8
# bytes of the fused kernels therefore equal the bytes of these values except where the value lies within a few float32
# ulps of an FP8 rounding boundary -- the property tests/test_kernels_gpu.py asserts (fp8_mismatches_near_boundary).
def swiglu_f32_device_order(h_bits: np.ndarray, bias_bits: np.ndarray = None) -> np.ndarray:
    """(g * sigmoid(g)) * u with sigmoid(g) = 1 / (1 + exp(-g)), all float32, left to right as in swiglu_cast_kernel MODE 0.
    `bias_bits` (bf16 [2F]): the fused-bias form -- one float32 add per element on the unpacked gate / up values first."""
    h = bf16_bits_to_f32(h_bits)
    if bias_bits is not None:
        h = (h + bf16_bits_to_f32(bias_bits)[None, :]).astype(np.float32)
    f = h.shape[1] // 2
    g, u = h[:, :f], h[:, f:]
    one = np.float32(1.0)
    with np.errstate(over="ignore"):
        sg = one / (one + np.exp(-g, dtype=np.float32))
    return ((g * sg) * u).astype(np.float32)


def dswiglu_f32_device_order(h_bits: np.ndarray, dact_bits: np.ndarray, bias_bits: np.ndarray = None) -> np.ndarray:
    """[(d * u) * (sg * (1 + g * (1 - sg))) | d * (g * sg)] in float32, the grouping of swiglu_cast_kernel MODE 1.
    `bias_bits`: as swiglu_f32_device_order."""
    h = bf16_bits_to_f32(h_bits)
    if bias_bits is not None:
        h = (h + bf16_bits_to_f32(bias_bits)[None, :]).astype(np.float32)
    d = bf16_bits_to_f32(dact_bits)
    f = h.shape[1] // 2
    g, u = h[:, :f], h[:, f:]
    one = np.float32(1.0)
    with np.errstate(over="ignore"):
        sg = one / (one + np.exp(-g, dtype=np.float32))
    left = (d * u) * (sg * (one + g * (one - sg)))
    right = d * (g * sg)
    return np.concatenate([left, right], axis=1).astype(np.float32)


def rmsnorm_sumsq_device_order(x_bits: np.ndarray) -> np.ndarray:
    """Row sums of squares in the order of rmsnorm_stats_kernel: lane l (of 64) walks columns 8 l + 512 t, adds
    (a*a + b*b) for its 4 bf16 pairs in turn, then the 64 partial sums meet in a butterfly (xor 32, 16, 8, 4, 2, 1)."""
    x = bf16_bits_to_f32(x_bits)
    rows, cols = x.shape
    assert cols % 8 == 0
    pad = (-cols) % 512
    if pad:
        x = np.concatenate([x, np.zeros((rows, pad), np.float32)], axis=1)  # lanes past the row end simply do not iterate
    xt = x.reshape(rows, -1, 64, 4, 2)                                       # [row, t, lane, pair, (a, b)]
    acc = np.zeros((rows, 64), np.float32)
    for t in range(xt.shape[1]):
        valid = (np.arange(64) * 8 + 512 * t) < cols
        for j in range(4):
            a, b = xt[:, t, :, j, 0], xt[:, t, :, j, 1]
            term = (a * a + b * b).astype(np.float32)
            acc = np.where(valid[None, :], (acc + term).astype(np.float32), acc)
    for o in (32, 16, 8, 4, 2, 1):
        acc = (acc + acc[:, np.arange(64) ^ o]).astype(np.float32)
    return acc[:, 0]


def rmsnorm_rstd_device_order(x_bits: np.ndarray, eps: float) -> np.ndarray:
    """1 / sqrt(sumsq / cols + eps) in float32 (the device uses v_rsq_f32: equal up to its 1-2 ulp approximation error)."""
    ss = rmsnorm_sumsq_device_order(x_bits)
    cols = np.float32(x_bits.shape[1])
    return (np.float32(1.0) / np.sqrt((ss / cols + np.float32(eps)).astype(np.float32))).astype(np.float32)


def norm_apply_f32_device_order(x_bits: np.ndarray, rstd_f32: np.ndarray, gamma_bits: np.ndarray) -> np.ndarray:
    """(x * rstd[row]) * gamma[col] in float32, the grouping of norm_cast_kernel: no transcendental, so with the device's own
    rstd as input the fused kernel's FP8 bytes must equal the bytes of this value exactly."""
    x = bf16_bits_to_f32(x_bits)
    g = bf16_bits_to_f32(gamma_bits)
    return ((x * rstd_f32.astype(np.float32)[:, None]).astype(np.float32) * g[None, :]).astype(np.float32)


def fp8_mismatches_near_boundary(got_bytes: np.ndarray, v_scaled_f32: np.ndarray, fmt: int, rel: float = 2.0 ** -17, abs_slack=None):
    """For FP8 bytes produced from a float32 value that may differ from `v_scaled_f32` by a few ulps: returns
    (n_mismatch, n_unexplained) where a mismatch is EXPLAINED when the two codes are neighbours and `v_scaled_f32` lies within
    `rel` (relative) of the rounding boundary between them (the midpoint of the two FP8 values; the saturation edge counts).
    `abs_slack` (same shape as the value, optional): additional absolute distance allowed per element, for expressions with
    a cancellation (dsilu has a zero near g = -1.2785: there the error is a few ulps of the TERMS, not of the tiny result)."""
    want = fp8_encode_sat(v_scaled_f32.astype(np.float32), fmt)
    got = np.asarray(got_bytes)
    both_zero = ((got & 0x7F) == 0) & ((want & 0x7F) == 0)
    mism = (got != want) & ~both_zero
    if not mism.any():
        return 0, 0
    tab = fp8_decode_table(fmt).astype(np.float64)
    gv, wv = tab[got[mism]], tab[want[mism]]
    v = v_scaled_f32[mism].astype(np.float64)
    neighbours = np.abs((got[mism].astype(np.int16) & 0x7F) - (want[mism].astype(np.int16) & 0x7F)) == 1
    same_sign = ((got[mism] ^ want[mism]) & 0x80) == 0
    mid = 0.5 * (gv + wv)
    slack = 0.0 if abs_slack is None else np.asarray(abs_slack, dtype=np.float64)[mism]
    near = np.abs(v - mid) <= rel * np.maximum(np.abs(mid), 1e-30) + slack
    explained = neighbours & same_sign & near & np.isfinite(mid)
    return int(mism.sum()), int((~explained).sum())


def rope_f32(x_bits: np.ndarray, pos: np.ndarray, head_dim: int, base: float = 10000.0, conj: bool = False) -> np.ndarray:
    """TE-style non-interleaved RoPE on [T, heads*D] (row t at position pos[t]); float32 math, one bf16 rounding."""
    x = bf16_bits_to_f32(x_bits).astype(np.float64)
    t, w = x.shape
    half = head_dim // 2
    inv = 1.0 / (base ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim))).astype(np.float32)
    ang = (pos.astype(np.float32)[:, None] * inv[None, :]).astype(np.float32)
    c, s = np.cos(ang.astype(np.float32)).astype(np.float64), np.sin(ang.astype(np.float32)).astype(np.float64)
    if conj:
        s = -s
    xh = x.reshape(t, w // head_dim, head_dim)
    x1, x2 = xh[..., :half], xh[..., half:]
    out = np.concatenate([x1 * c[:, None, :] - x2 * s[:, None, :], x2 * c[:, None, :] + x1 * s[:, None, :]], axis=-1)
    return f32_to_bf16_bits(out.reshape(t, w).astype(np.float32))


def rmsnorm_f32(x_bits: np.ndarray, gamma_bits: np.ndarray, eps: float):
    """(y float32 [R, C], rstd float32 [R]): y = x * rsqrt(mean(x^2) + eps) * gamma, float64 math (K9)."""
    x = bf16_bits_to_f32(x_bits).astype(np.float64)
    g = bf16_bits_to_f32(gamma_bits).astype(np.float64)
    rstd = 1.0 / np.sqrt((x * x).mean(axis=1) + eps)
    return (x * rstd[:, None] * g[None, :]).astype(np.float32), rstd.astype(np.float32)


def rmsnorm_bwd_f32(dy_bits: np.ndarray, x_bits: np.ndarray, gamma_bits: np.ndarray, eps: float):
    """(dx float32, dgamma float32) of y = x * rstd * gamma."""
    dy = bf16_bits_to_f32(dy_bits).astype(np.float64)
    x = bf16_bits_to_f32(x_bits).astype(np.float64)
    g = bf16_bits_to_f32(gamma_bits).astype(np.float64)
    rstd = 1.0 / np.sqrt((x * x).mean(axis=1) + eps)
    xh = x * rstd[:, None]
    wdy = dy * g[None, :]
    c = (wdy * xh).mean(axis=1)
    dx = rstd[:, None] * (wdy - xh * c[:, None])
    return dx.astype(np.float32), (dy * xh).sum(axis=0).astype(np.float32)


def attention_f64(q_bits, k_bits, v_bits, scale: float, causal: bool = True):
    """Attention core (te_llama.py:45-56: causal, GQA): q [B,S,H,D], k/v [B,S,G,D] bf16 bits -> (o f32 [B,S,H,D], lse_log2 [B,H,S]).
    float64 math; lse in the log2 domain of the scaled scores (the layout the HIP kernel keeps for its backward)."""
    q = bf16_bits_to_f32(q_bits).astype(np.float64)
    k = bf16_bits_to_f32(k_bits).astype(np.float64)
    v = bf16_bits_to_f32(v_bits).astype(np.float64)
    B, S, H, D = q.shape
    G = k.shape[2]
    rep = H // G
    o = np.zeros((B, S, H, D))
    lse = np.zeros((B, H, S))
    mask = np.triu(np.ones((S, S), dtype=bool), 1)
    for h in range(H):
        g = h // rep
        s = np.einsum("bqd,bkd->bqk", q[:, :, h], k[:, :, g]) * scale
        if causal:
            s = np.where(mask[None], -np.inf, s)
        m = s.max(-1, keepdims=True)
        p = np.exp(s - m)
        l = p.sum(-1, keepdims=True)
        o[:, :, h] = np.einsum("bqk,bkd->bqd", p / l, v[:, :, g])
        lse[:, h] = (m[..., 0] + np.log(l[..., 0])) / np.log(2.0)
    return o.astype(np.float32), lse.astype(np.float32)


def attention_bwd_f64(q_bits, k_bits, v_bits, do_bits, scale: float, causal: bool = True):
    """(dq, dk, dv) float32 of attention_f64's o w.r.t. its inputs, given dO (bf16 bits)."""
    q = bf16_bits_to_f32(q_bits).astype(np.float64)
    k = bf16_bits_to_f32(k_bits).astype(np.float64)
    v = bf16_bits_to_f32(v_bits).astype(np.float64)
    do = bf16_bits_to_f32(do_bits).astype(np.float64)
    B, S, H, D = q.shape
    G = k.shape[2]
    rep = H // G
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    mask = np.triu(np.ones((S, S), dtype=bool), 1)
    for h in range(H):
        g = h // rep
        s = np.einsum("bqd,bkd->bqk", q[:, :, h], k[:, :, g]) * scale
        if causal:
            s = np.where(mask[None], -np.inf, s)
        p = np.exp(s - s.max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        dv[:, :, g] += np.einsum("bqk,bqd->bkd", p, do[:, :, h])
        dp = np.einsum("bqd,bkd->bqk", do[:, :, h], v[:, :, g])
        ds = p * (dp - (p * dp).sum(-1, keepdims=True)) * scale
        dq[:, :, h] = np.einsum("bqk,bkd->bqd", ds, k[:, :, g])
        dk[:, :, g] += np.einsum("bqk,bqd->bkd", ds, q[:, :, h])
    return dq.astype(np.float32), dk.astype(np.float32), dv.astype(np.float32)
