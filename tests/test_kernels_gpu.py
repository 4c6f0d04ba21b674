"""GPU parity tests: every HIP kernel behind the C ABI against the CPU oracle.
Casts / scales / MX bytes: bit-exact.  GEMMs: |d| <= 2^-7 |ref| + 1e-3 rms(ref)."""
import os

import numpy as np
import pytest
import torch

from oracle import fp8_oracle as O
from tests.util import assert_gemm_close, bf16_bits, bits_to_bf16, dequant_table, u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from llm_fp8_amd.pytorch import ops as _ops
    from llm_fp8_amd import _lib
    assert _lib.load().mi_device_supported() == 1, "not a gfx950 device"
    return _ops


def _f32(v, dev):
    return torch.tensor([v], dtype=torch.float32, device=dev)


# ----------------------------------------------------------------------------------------- K1/K2
@pytest.mark.parametrize("fmt", [O.E4M3, O.E5M2])
@pytest.mark.parametrize("scale", [1.0, 0.5, 3.7, 448.0 / 3.3, 1e-3, 1e4, 2.0 ** -20, 3e38])
def test_cast_exhaustive_all_bf16(ops, dev, fmt, scale):
    bits = np.arange(65536, dtype=np.uint32).astype(np.uint16).reshape(256, 256)
    x = bits_to_bf16(bits, dev)
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y, yT = ops.cast_amax(x, _f32(scale, dev), amax, fmt)
    q, a = O.quantize_delayed(bits, np.float32(scale), fmt)
    got = u8(y)
    bad = np.nonzero(got != q)
    assert bad[0].size == 0, f"{bad[0].size} byte mismatches, first: in={bits[bad][0]:#06x} got={got[bad][0]:#04x} want={q[bad][0]:#04x}"
    np.testing.assert_array_equal(u8(yT), q.T)
    assert amax.item() == a == np.inf


@pytest.mark.parametrize("shape", [(8, 8), (16, 136), (136, 72), (1000, 264), (1024, 3072), (4104, 3080)])  # the last: > 1 tile per wave of the persistent walk, ragged both ways
@pytest.mark.parametrize("fmt", [O.E4M3, O.E5M2])
def test_cast_ragged_shapes_and_amax(ops, dev, shape, fmt):
    g = torch.Generator().manual_seed(shape[0] * 7 + shape[1])
    x = (torch.randn(shape, generator=g) * 3).to(torch.bfloat16)
    x[0, 0] = float("nan")
    bits = bf16_bits(x)
    scale = np.float32(448.0 / 11.0)
    amax = torch.full((1,), 0.25, dtype=torch.float32, device=dev)
    y, yT = ops.cast_amax(x.to(dev), _f32(scale, dev), amax, fmt)
    q, a = O.quantize_delayed(bits, scale, fmt)
    np.testing.assert_array_equal(u8(y), q)
    np.testing.assert_array_equal(u8(yT), q.T)
    assert amax.item() == max(a, np.float32(0.25))
    # y-only and yT-only variants, amax optional
    y2, t2 = ops.cast_amax(x.to(dev), _f32(scale, dev), None, fmt, want_t=False)
    assert t2 is None
    np.testing.assert_array_equal(u8(y2), q)
    y3, t3 = ops.cast_amax(x.to(dev), _f32(scale, dev), None, fmt, want_y=False)
    assert y3 is None
    np.testing.assert_array_equal(u8(t3), q.T)


def test_cast_into_slices_of_fused_buffers(ops, dev):
    """q|k|v weights are cast into row-slices of ONE fused operand and share one amax slot
    (SURVEY App. A "Fused-QKV"; te_llama.py:194-217)."""
    g = torch.Generator().manual_seed(11)
    K = 64
    parts = [(torch.randn(n, K, generator=g) * (i + 1)).to(torch.bfloat16) for i, n in enumerate((48, 16, 16))]
    N = sum(p.shape[0] for p in parts)
    y = torch.zeros((N, K), dtype=torch.uint8, device=dev)
    yT = torch.zeros((K, N), dtype=torch.uint8, device=dev)
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    r = 0
    for p in parts:
        n = p.shape[0]
        ops.cast_amax(p.to(dev), _f32(2.0, dev), amax, O.E4M3, y=y[r:r + n], yT=yT[:, r:r + n])
        r += n
    full = torch.cat(parts, 0)
    q, a = O.quantize_delayed(bf16_bits(full), np.float32(2.0), O.E4M3)
    np.testing.assert_array_equal(u8(y), q)
    np.testing.assert_array_equal(u8(yT), q.T)
    assert amax.item() == a


def test_cast_rejects_bad_args(ops, dev):
    x = torch.zeros((12, 8), dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.cast_amax(x, _f32(1.0, dev), None, O.E4M3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cast_amax(x.cpu(), _f32(1.0, dev), None, O.E4M3)


@pytest.mark.parametrize("fmt", [O.E4M3, O.E5M2])
def test_cast_full_size_properties(ops, dev, fmt):
    """BASELINE size (M=8192 x K=3072): transpose consistency, amax, idempotent dequant bound."""
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((8192, 3072), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    scale = 448.0 / 8.0 if fmt == O.E4M3 else 57344.0 / 8.0
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y, yT = ops.cast_amax(x, _f32(scale, dev), amax, fmt)
    assert torch.equal(yT, y.t().contiguous())
    assert amax.item() == x.float().abs().max().item()
    deq = dequant_table(fmt, dev)[y.long()] / scale
    rel = ((deq - x.float()).abs() / x.float().abs().clamp_min(2.0 ** -6)).max().item()
    assert rel <= (2.0 ** -4 if fmt == O.E4M3 else 2.0 ** -3) * 1.01


# ----------------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("H,algo,margin", [(16, "max", 0), (1024, "most_recent", 0), (4, "max", 2), (1, "max", 0),
                                            (300, "max", 0)])
def test_scale_update_trajectory(ops, dev, H, algo, margin):
    S = 37
    rng = np.random.default_rng(H)
    hist = np.zeros((H, S), np.float32)
    scale = np.ones(S, np.float32)
    fmax = np.where(np.arange(S) % 3 == 0, 57344.0, 448.0).astype(np.float32)
    d_hist = torch.zeros((H, S), dtype=torch.float32, device=dev)
    d_scale = torch.ones(S, dtype=torch.float32, device=dev)
    d_inv = torch.ones(S, dtype=torch.float32, device=dev)
    d_fmax = torch.from_numpy(fmax).to(dev)
    for step in range(24):
        am = np.exp(rng.normal(size=S) * 4).astype(np.float32)
        am[rng.random(S) < 0.2] = 0.0
        if step == 5:
            am[1] = np.inf
        if step == 7:
            am[2] = 1e-42  # subnormal amax -> fp8_max/amax overflows -> FLT_MAX guard
        if step == 9:
            am[3] = 3e38
        hist[0] = am
        d_hist[0] = torch.from_numpy(am).to(dev)
        hist, scale, inv = O.scale_update(hist, scale, fmax, margin, algo)
        ops.scale_update(d_hist, d_scale, d_inv, d_fmax, margin, algo)
        np.testing.assert_array_equal(d_hist.cpu().numpy().view(np.uint32), hist.view(np.uint32), err_msg=f"hist step {step}")
        np.testing.assert_array_equal(d_scale.cpu().numpy().view(np.uint32), scale.view(np.uint32), err_msg=f"scale step {step}")
        np.testing.assert_array_equal(d_inv.cpu().numpy().view(np.uint32), inv.view(np.uint32), err_msg=f"inv step {step}")


# ----------------------------------------------------------------------------------------- K4-K6
def _rand_fp8(shape, fmt, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    v = (rng.normal(size=shape) * spread * np.exp(rng.normal(size=shape))).astype(np.float32)
    return O.fp8_encode_sat(v, fmt)


def assert_mfma_close(got, ref, a8, b8, fa, fb, alpha):
    """fp32-output bound from the measured MFMA accumulation behaviour (tools/probe_mfma.hip): inside one
    128-deep instruction, products are added in groups of 8 aligned to the group's largest product and
    anything 2^14 below it is dropped -> |err| <= 7 * 2^-14 * sum_k |a_k b_k| worst case (typical data is
    far below it); fp32 rounding of the running sum adds ~K/128 ulps."""
    mag = (np.abs(O.fp8_decode(a8, fa)).astype(np.float64) @ np.abs(O.fp8_decode(b8, fb)).astype(np.float64).T) * alpha
    tol = 7 * 2.0 ** -14 * mag + 1e-5 * np.abs(ref)
    diff = np.abs(got.astype(np.float64) - ref)
    assert (diff <= tol).all(), f"max diff/bound = {(diff / np.maximum(tol, 1e-300)).max():.3f}"
    # and the typical error is much smaller than the worst-case bound
    assert np.sqrt(np.mean(diff ** 2)) <= 2.0 ** -12 * np.sqrt(np.mean(mag ** 2))


GEMM_SHAPES = [(256, 256, 256), (512, 768, 640), (768, 384, 512), (1536, 1920, 256), (192, 192, 256), (2048, 2304, 512), (4096, 4352, 256), (64, 96, 128), (16, 16, 16), (8, 24, 48), (200, 136, 400), (256, 256, 128), (256, 512, 384),
               (512, 256, 3072), (256, 5120, 3072), (768, 1024, 256)]


@pytest.mark.parametrize("shape", GEMM_SHAPES)
@pytest.mark.parametrize("fa,fb", [(O.E4M3, O.E4M3), (O.E5M2, O.E4M3), (O.E4M3, O.E5M2), (O.E5M2, O.E5M2)])
@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 5, 6, 9, 41, 42, 43])
def test_gemm_fp8_vs_oracle(ops, dev, shape, fa, fb, algo):
    M, N, K = shape
    if algo in (6, 9) and (M % 256 or N % 256 or K % 256 or (algo == 9 and K < 512)):
        pytest.skip("four-wave kernels need 256-aligned M, N, K (the persistent one K >= 512)")
    if algo == 1 and M * N * K > 256 * 512 * 3072:
        pytest.skip("generic path covered at smaller sizes")
    if algo in (2, 3) and (M % 256 or N % 256 or K % 128):
        pytest.skip("fast kernels need 256/256/128-aligned shapes")
    if algo in (4, 5) and ((M % 256 and M % 192) or (N % 256 and N % 192) or K % 256):
        pytest.skip("persistent kernel needs 256- or 192-aligned M, N and 256-aligned K")
    if algo in (41, 42, 43):
        bm, bn = {41: (256, 192), 42: (192, 256), 43: (192, 192)}[algo]
        if M % bm or N % bn or K % 256:
            pytest.skip("tile shape does not divide the problem")
    a8 = _rand_fp8((M, K), fa, 1 + M, 4.0 if fa == O.E4M3 else 64.0)
    b8 = _rand_fp8((N, K), fb, 2 + N, 4.0 if fb == O.E4M3 else 64.0)
    sa, sb = np.float32(1 / 7.3), np.float32(1 / 0.011)
    rng = np.random.default_rng(3)
    bias = O.f32_to_bf16_bits(rng.normal(size=N).astype(np.float32) * 10)
    for use_bias in ((False,) if algo == 6 else (False, True)):  # of the four-wave kernels only the persistent one takes a bias
        ref = O.gemm_fp8_tn(a8, b8, fa, fb, sa, sb, bias if use_bias else None, out_f32=True)
        d = ops.gemm_fp8(torch.from_numpy(a8).to(dev), torch.from_numpy(b8).to(dev), _f32(sa, dev), _f32(sb, dev),
                         fa, fb, bias=bits_to_bf16(bias, dev) if use_bias else None, algo=algo)
        assert_gemm_close(d.float().cpu().numpy(), ref, f"gemm {shape} fmt({fa},{fb}) algo {algo} bias {use_bias}")
    if algo >= 4:
        return  # bf16 output only
    # fp32 output
    d32 = ops.gemm_fp8(torch.from_numpy(a8).to(dev), torch.from_numpy(b8).to(dev), _f32(sa, dev), _f32(sb, dev),
                       fa, fb, out_dtype=torch.float32, algo=algo)
    ref = O.gemm_fp8_tn(a8, b8, fa, fb, sa, sb, None, out_f32=True)
    assert_mfma_close(d32.cpu().numpy(), ref, a8, b8, fa, fb, float(sa) * float(sb))


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_gemm_identity_with_asymmetric_b(ops, dev, algo):
    """A = I (padded), B asymmetric small integers: exact result, catches row/col swaps and K-permutation
    mismatches between the A and B fragments."""
    M = N = K = 256
    a = np.zeros((M, K), np.float32)
    a[np.arange(M), np.arange(K)] = 1.0
    b = ((np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 5) % 17 - 8).astype(np.float32)
    a8, b8 = O.fp8_encode_sat(a, O.E4M3), O.fp8_encode_sat(b, O.E4M3)
    d = ops.gemm_fp8(torch.from_numpy(a8).to(dev), torch.from_numpy(b8).to(dev), _f32(1.0, dev), _f32(1.0, dev),
                     O.E4M3, O.E4M3, out_dtype=torch.float32, algo=algo)
    np.testing.assert_array_equal(d.cpu().numpy(), b.T)


def test_gemm_full_size_vs_device_fp32_matmul(ops, dev):
    """BASELINE 3B qkv shape (8192 x 5120 x 3072): compare with an fp32 matmul of the dequantised operands
    computed on the device by torch (independent of our kernel), plus linearity in alpha."""
    M, N, K = 8192, 5120, 3072
    g = torch.Generator(device=dev).manual_seed(9)
    a8 = torch.randint(0, 256, (M, K), generator=g, device=dev, dtype=torch.uint8)
    b8 = torch.randint(0, 256, (N, K), generator=g, device=dev, dtype=torch.uint8)
    a8[(a8 & 0x7F) == 0x7F] = 0  # no NaN bytes
    b8[(b8 & 0x7F) == 0x7F] = 0
    a8 &= 0xBF  # keep magnitudes < 2 so the fp32 reference matmul is well conditioned
    b8 &= 0xBF
    ta, tb = dequant_table(O.E4M3, dev), dequant_table(O.E4M3, dev)
    ref = (ta[a8.long()] @ tb[b8.long()].t())
    one, half = _f32(1.0, dev), _f32(0.5, dev)
    d = ops.gemm_fp8(a8, b8, one, one, O.E4M3, O.E4M3, out_dtype=torch.float32)
    rms = ref.pow(2).mean().sqrt().item()
    mag = ta[a8.long()].abs() @ tb[b8.long()].abs().t()
    assert ((d - ref).abs() <= 2.0 ** -12 * mag + 1e-4 * rms).all()  # MFMA group truncation + torch's own fp32 matmul error
    d2 = ops.gemm_fp8(a8, b8, half, one, O.E4M3, O.E4M3, out_dtype=torch.float32)
    assert torch.equal(d2, d * 0.5)
    dbf = ops.gemm_fp8(a8, b8, one, one, O.E4M3, O.E4M3)
    assert torch.equal(dbf, d.to(torch.bfloat16))


def test_gemm_rejects_bad_shapes(ops, dev):
    a = torch.zeros((16, 24), dtype=torch.uint8, device=dev)
    b = torch.zeros((16, 24), dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="K of 16"):
        ops.gemm_fp8(a, b, _f32(1, dev), _f32(1, dev), 0, 0)
    a = torch.zeros((64, 128), dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="needs M,N"):
        ops.gemm_fp8(a, a, _f32(1, dev), _f32(1, dev), 0, 0, algo=2)


# ----------------------------------------------------------------------------------------- K7/K8
@pytest.mark.parametrize("shape", [(32, 32), (64, 96), (160, 416), (1024, 3072)])
@pytest.mark.parametrize("fmt", [O.E4M3, O.E5M2])
def test_mxfp8_quantize_bitexact(ops, dev, shape, fmt):
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g) * torch.exp(torch.randn(shape[0], 1, generator=g) * 5)
    x[:, :32] = 0  # all-zero blocks
    if shape[1] >= 64:
        x[1, 40] = float("inf")
        x[2, 41] = float("nan")
        x[3, 33:64] = 1e-30
    x[5, 3] = float("nan")  # NaN inside an otherwise-zero block
    x[6, 7] = 3e38
    x = x.to(torch.bfloat16)
    bits = bf16_bits(x)
    y_row, s_row, y_colT, s_colT = ops.mxfp8_quantize(x.to(dev), fmt)
    q, e = O.mxfp8_quantize_rowwise(bits, fmt)
    qc, ec = O.mxfp8_quantize_colwise(bits, fmt)
    np.testing.assert_array_equal(u8(s_row), e.T)  # device scales are block-major [C/32, R]
    np.testing.assert_array_equal(u8(y_row), q)
    np.testing.assert_array_equal(u8(s_colT), ec.T)
    np.testing.assert_array_equal(u8(y_colT), qc)
    yr, sr, yc, sc = ops.mxfp8_quantize(x.to(dev), fmt, colwise=False)
    assert yc is None and sc is None
    np.testing.assert_array_equal(u8(yr), q)
    yr, sr, yc, sc = ops.mxfp8_quantize(x.to(dev), fmt, rowwise=False)
    assert yr is None
    np.testing.assert_array_equal(u8(yc), qc)


@pytest.mark.parametrize("shape", [(32, 32, 32), (64, 96, 128), (96, 160, 320), (256, 512, 1024), (256, 256, 256),
                                   (768, 512, 512), (2048, 2304, 768), (768, 576, 512), (384, 1536, 256)])
@pytest.mark.parametrize("algo", [1, 4, 41, 42, 43])
def test_gemm_mxfp8_vs_oracle(ops, dev, shape, algo):
    M, N, K = shape
    if algo in (4, 41, 42, 43):
        bm, bn = {4: (256, 256), 41: (256, 192), 42: (192, 256), 43: (192, 192)}[algo]
        if M % bm or N % bn or K % 256:
            pytest.skip("tile shape does not divide the problem")
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K // 32, generator=g).repeat_interleave(32, 1) * 2)).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * torch.exp(torch.randn(N, K // 32, generator=g).repeat_interleave(32, 1) * 2)).to(torch.bfloat16)
    a8, ae = O.mxfp8_quantize_rowwise(bf16_bits(a))
    b8, be = O.mxfp8_quantize_rowwise(bf16_bits(b))
    ref = O.gemm_mxfp8_tn(a8, ae, b8, be, out_f32=True)
    t = lambda v: torch.from_numpy(v).to(dev)
    tT = lambda v: torch.from_numpy(np.ascontiguousarray(v.T)).to(dev)  # block-major scales
    d = ops.gemm_mxfp8(t(a8), tT(ae), t(b8), tT(be), out_dtype=torch.float32, algo=1)
    sa = np.repeat(O.e8m0_to_f32(ae).astype(np.float64), 32, axis=1)
    sb = np.repeat(O.e8m0_to_f32(be).astype(np.float64), 32, axis=1)
    mag = (np.abs(O.fp8_decode(a8, O.E4M3)) * sa) @ (np.abs(O.fp8_decode(b8, O.E4M3)) * sb).T
    diff = np.abs(d.cpu().numpy().astype(np.float64) - ref)
    assert (diff <= 7 * 2.0 ** -14 * mag + 1e-5 * np.abs(ref)).all()
    dbf = ops.gemm_mxfp8(t(a8), tT(ae), t(b8), tT(be), algo=algo)
    assert_gemm_close(dbf.float().cpu().numpy(), ref, f"mx gemm {shape}")
    bias = O.f32_to_bf16_bits(np.random.default_rng(5).normal(size=N).astype(np.float32) * np.abs(ref).mean())
    refb = O.gemm_mxfp8_tn(a8, ae, b8, be, bias_bf16_bits=bias, out_f32=True)
    dbb = ops.gemm_mxfp8(t(a8), tT(ae), t(b8), tT(be), bias=bits_to_bf16(bias, dev), algo=algo)
    assert_gemm_close(dbb.float().cpu().numpy(), refb, f"mx gemm + bias {shape}")


# ----------------------------------------------------------------------------------------- fused neighbours (RoPE, K10)
def _ulp_close_fp8(got, want, fmt, frac_exact=0.995):
    """fp8 bytes equal, except where the fp32 value sits on a rounding boundary (device exp vs numpy exp): those may
    differ by one code; at least `frac_exact` of the bytes must be identical."""
    got, want = got.astype(np.int16), want.astype(np.int16)
    diff = np.abs((got & 0x7F) - (want & 0x7F))
    same_sign = ((got ^ want) & 0x80) == 0
    ok = (diff == 0) & same_sign | (diff == 1) & same_sign | ((got & 0x7F) + (want & 0x7F) <= 1)
    assert ok.all(), f"{(~ok).sum()} bytes differ by more than one fp8 code"
    assert (diff == 0).mean() >= frac_exact, f"only {(diff == 0).mean():.4f} exact"


def _assert_matches_fp32_restatement(got_bytes, v_scaled_f32, fmt, what, max_frac=1e-3, abs_slack=None):
    """The device computed the same float32 operations in the same order as the oracle's *_device_order restatement; only its
    transcendental (v_exp_f32 / v_rsq_f32, a few ulps) can differ from numpy's.  So the FP8 bytes are identical EXCEPT where the
    float32 value lies within 2^-17 (relative) of the rounding boundary between two neighbouring codes -- every mismatch must be
    explained that way, and there can only be a handful (a boundary band of 2^-17 against a code spacing of 2^-4 .. 2^-2)."""
    n_mis, n_bad = O.fp8_mismatches_near_boundary(got_bytes, v_scaled_f32, fmt, abs_slack=abs_slack)
    assert n_bad == 0, f"{what}: {n_bad} of {n_mis} mismatching bytes are NOT within 2^-17 of a rounding boundary"
    assert n_mis <= max(2, max_frac * got_bytes.size), f"{what}: {n_mis} of {got_bytes.size} bytes differ"


@pytest.mark.parametrize("shape", [(8, 8), (136, 72), (1024, 3072)])
@pytest.mark.parametrize("fmt", [O.E4M3, O.E5M2])
def test_swiglu_cast_vs_oracle(ops, dev, shape, fmt):
    R, F = shape
    g = torch.Generator().manual_seed(R + F)
    h = (torch.randn(R, 2 * F, generator=g) * 2).to(torch.bfloat16)
    scale = np.float32(16.0)
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y, yT = ops.swiglu_cast(h.to(dev), _f32(scale, dev), amax, fmt)
    act = O.swiglu_f32(bf16_bits(h))
    want = O.fp8_encode_sat((act * scale).astype(np.float32), fmt)
    _ulp_close_fp8(u8(y), want, fmt)  # float64 reference: one code at most, >= 99.5 % identical
    act32 = O.swiglu_f32_device_order(bf16_bits(h))  # float32, the kernel's operation order: mismatches only at rounding boundaries
    _assert_matches_fp32_restatement(u8(y), (act32 * scale).astype(np.float32), fmt, "swiglu_cast")
    np.testing.assert_array_equal(u8(yT), u8(y).T)
    np.testing.assert_allclose(amax.item(), np.abs(act).max(), rtol=1e-5)
    np.testing.assert_allclose(amax.item(), np.abs(act32).max(), rtol=2e-6)


@pytest.mark.parametrize("shape", [(8, 8), (136, 72), (1024, 3072)])
def test_swiglu_kernels_with_fused_bias_vs_oracle(ops, dev, shape):
    """mi_swiglu_cast_bias / mi_dswiglu_cast_bias / mi_add_bias_rmsnorm_stats: the MLP biases added inside the consumer kernels
    (TE's bias + activation fusion; LayerNormMLP keeps TE's default bias=True at te_llama.py:58-63).  Held to the float32
    restatement in the device's operation order (one fp32 add on the unpacked values first): every mismatching byte must sit
    within 2^-17 of a rounding boundary; with a ZERO bias the outputs are bit for bit those of the kernels without bias; the
    residual add is exact (no transcendental): bytes identical to bf16(a + (b + bias))."""
    R, F = shape
    g = torch.Generator().manual_seed(R * 7 + F)
    h = (torch.randn(R, 2 * F, generator=g) * 2).to(torch.bfloat16)
    d = (torch.randn(R, F, generator=g) / 8).to(torch.bfloat16)
    bias = (torch.randn(2 * F, generator=g) * 0.5).to(torch.bfloat16)
    scale = np.float32(16.0)
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y, yT = ops.swiglu_cast(h.to(dev), _f32(scale, dev), amax, O.E4M3, bias=bias.to(dev))
    act32 = O.swiglu_f32_device_order(bf16_bits(h), bf16_bits(bias))
    _assert_matches_fp32_restatement(u8(y), (act32 * scale).astype(np.float32), O.E4M3, "swiglu_cast + bias")
    np.testing.assert_array_equal(u8(yT), u8(y).T)
    np.testing.assert_allclose(amax.item(), np.abs(act32).max(), rtol=2e-6)
    s2 = np.float32(64.0)
    y2, y2T, cs = ops.dswiglu_cast(h.to(dev), d.to(dev), _f32(s2, dev), None, O.E5M2, want_colsum=True, bias=bias.to(dev))
    dh32 = O.dswiglu_f32_device_order(bf16_bits(h), bf16_bits(d), bf16_bits(bias))
    hb = (O.bf16_bits_to_f32(bf16_bits(h)) + O.bf16_bits_to_f32(bf16_bits(bias))[None, :]).astype(np.float32)
    df = O.bf16_bits_to_f32(bf16_bits(d))
    slack = np.concatenate([np.abs(df * hb[:, F:]) * s2 * 2.0 ** -21, np.zeros_like(df)], axis=1)
    _assert_matches_fp32_restatement(u8(y2), (dh32 * s2).astype(np.float32), O.E5M2, "dswiglu_cast + bias", abs_slack=slack)
    np.testing.assert_array_equal(u8(y2T), u8(y2).T)
    np.testing.assert_allclose(cs.sum(0).cpu().numpy(), dh32.astype(np.float64).sum(0), rtol=1e-4, atol=1e-4 * np.abs(dh32).max() * np.sqrt(R))
    zero = torch.zeros(2 * F, dtype=torch.bfloat16, device=dev)
    a0, a0T = ops.swiglu_cast(h.to(dev), _f32(scale, dev), None, O.E4M3)
    a1, a1T = ops.swiglu_cast(h.to(dev), _f32(scale, dev), None, O.E4M3, bias=zero)
    assert torch.equal(a0, a1) and torch.equal(a0T, a1T)
    # residual add with the fc2 bias: exact arithmetic
    C = 2 * F if (2 * F) % 8 == 0 else 8
    a = torch.randn(R, C, generator=g).to(torch.bfloat16)
    b = torch.randn(R, C, generator=g).to(torch.bfloat16)
    bb = (torch.randn(C, generator=g) * 0.1).to(torch.bfloat16)
    out, rstd = ops.add_rmsnorm_stats(a.to(dev), b.to(dev), 1e-5, bias=bb.to(dev))
    want = (a.float() + (b.float() + bb.float()[None, :])).to(torch.bfloat16)
    assert torch.equal(out.cpu(), want)
    out0, rstd0 = ops.add_rmsnorm_stats(a.to(dev), b.to(dev), 1e-5)
    out1, rstd1 = ops.add_rmsnorm_stats(a.to(dev), b.to(dev), 1e-5, bias=torch.zeros(C, dtype=torch.bfloat16, device=dev))
    assert torch.equal(out0, out1) and torch.equal(rstd0, rstd1)
    np.testing.assert_allclose(rstd.cpu().numpy(), 1.0 / np.sqrt((want.float().numpy().astype(np.float64) ** 2).mean(1) + 1e-5), rtol=1e-5)


@pytest.mark.parametrize("shape", [(8, 8), (136, 72), (1024, 3072)])
def test_dswiglu_cast_vs_oracle(ops, dev, shape):
    R, F = shape
    g = torch.Generator().manual_seed(R * 3 + F)
    h = (torch.randn(R, 2 * F, generator=g) * 2).to(torch.bfloat16)
    d = (torch.randn(R, F, generator=g) / 8).to(torch.bfloat16)
    scale = np.float32(64.0)
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y, yT, cs = ops.dswiglu_cast(h.to(dev), d.to(dev), _f32(scale, dev), amax, O.E5M2, want_colsum=True)
    dh = O.dswiglu_f32(bf16_bits(h), bf16_bits(d))
    want = O.fp8_encode_sat((dh * scale).astype(np.float32), O.E5M2)
    _ulp_close_fp8(u8(y), want, O.E5M2)
    dh32 = O.dswiglu_f32_device_order(bf16_bits(h), bf16_bits(d))
    # dsilu(g) = s (1 + g (1 - s)) vanishes near g = -1.2785: around that zero the result is the difference of two O(1) terms and
    # carries a few float32 ulps OF THOSE TERMS (times |d u|), however small it is itself -> absolute slack 2^-21 |d u| scale there
    hf, df = O.bf16_bits_to_f32(bf16_bits(h)), O.bf16_bits_to_f32(bf16_bits(d))
    slack = np.concatenate([np.abs(df * hf[:, F:]) * scale * 2.0 ** -21, np.zeros_like(df)], axis=1)
    _assert_matches_fp32_restatement(u8(y), (dh32 * scale).astype(np.float32), O.E5M2, "dswiglu_cast", abs_slack=slack)
    np.testing.assert_array_equal(u8(yT), u8(y).T)
    np.testing.assert_allclose(amax.item(), np.abs(dh).max(), rtol=1e-5)
    np.testing.assert_allclose(cs.sum(0).cpu().numpy(), dh.astype(np.float64).sum(0), rtol=1e-4, atol=1e-4 * np.abs(dh).max() * np.sqrt(R))
    # the bias-gradient partial sums are bitwise reproducible
    _, _, cs2 = ops.dswiglu_cast(h.to(dev), d.to(dev), _f32(scale, dev), None, O.E5M2, want_colsum=True)
    assert torch.equal(cs, cs2)


@pytest.mark.parametrize("B,S,nq,nkv,D", [(2, 16, 4, 2, 64), (1, 128, 24, 8, 128), (3, 40, 8, 8, 32)])
def test_rope_qkv_split_and_merge_vs_oracle(ops, dev, B, S, nq, nkv, D):
    import llm_fp8_amd.pytorch as te
    g = torch.Generator().manual_seed(S)
    W = (nq + 2 * nkv) * D
    qkv = torch.randn(B * S, W, generator=g).to(torch.bfloat16)
    freqs = te.attention.RotaryPositionEmbedding(D)(max_seq_len=256).to(dev)
    from llm_fp8_amd.pytorch.attention import _cos_sin_tables
    cos, sin = _cos_sin_tables(freqs, S)
    q, k, v = ops.rope_qkv_forward(qkv.to(dev), cos, sin, nq, nkv, D, S)
    pos = np.tile(np.arange(S), B)
    bits = bf16_bits(qkv)
    q_ref = O.rope_f32(bits[:, :nq * D], pos, D)
    k_ref = O.rope_f32(bits[:, nq * D:(nq + nkv) * D], pos, D)
    # device cos/sin come from torch.cos on the GPU: allow 1 bf16 ulp on a handful of elements
    for got, ref in ((q, q_ref), (k, k_ref)):
        gf, rf = got.float().cpu().numpy(), O.bf16_bits_to_f32(ref)
        assert np.all(np.abs(gf - rf) <= 2.0 ** -7 * np.abs(rf) + 1e-5 * np.abs(qkv.float().numpy()).max())
        assert (bf16_bits(got) == ref).mean() > 0.98
    np.testing.assert_array_equal(bf16_bits(v), bits[:, (nq + nkv) * D:])
    # backward = conjugate rotation merged back into the fused layout; rotation is orthogonal: merge(split(x)) ~ x
    back = ops.rope_qkv_backward(q, k, v, cos, sin, nq, nkv, D, S)
    bf, xf = back.float().cpu().numpy(), qkv.float().numpy()
    assert np.all(np.abs(bf - xf) <= 2.0 ** -6 * np.abs(xf) + 2.0 ** -7 * np.abs(xf).max())
    dq_ref = O.rope_f32(bf16_bits(q), pos, D, conj=True)
    assert np.all(np.abs(back[:, :nq * D].float().cpu().numpy() - O.bf16_bits_to_f32(dq_ref)) <= 2.0 ** -7 * np.abs(O.bf16_bits_to_f32(dq_ref)) + 1e-5 * np.abs(xf).max())


@pytest.mark.parametrize("fmt", [O.E5M2, O.E4M3])
@pytest.mark.parametrize("shape", [(2, 128, 24, 8), (1, 72, 4, 2), (3, 64, 8, 8)])
def test_rope_backward_cast_is_bitwise_the_two_kernel_sequence(ops, dev, fmt, shape):
    """mi_rope_qkv_bwd_cast == mi_rope_qkv(backward) followed by mi_cast_amax: FP8 bytes both ways and the amax, bit for bit
    (ragged 64-row blocks, GQA and MHA head counts, NaN / inf in the gradients)."""
    B, S, nq, nkv = shape
    D, T = 128, shape[0] * shape[1]
    g = torch.Generator().manual_seed(S + nq)
    dq = (torch.randn(T, nq * D, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    dk = (torch.randn(T, nkv * D, generator=g) * 3.0).to(torch.bfloat16).to(dev)
    dv = torch.randn(T, nkv * D, generator=g).to(torch.bfloat16).to(dev)
    dq[1, 5] = float("nan")
    dv[3, 7] = float("inf")
    ang = torch.outer(torch.arange(S, dtype=torch.float32), 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
    cos, sin = torch.cos(ang).contiguous().to(dev), torch.sin(ang).contiguous().to(dev)
    scale = torch.tensor([37.0], device=dev)
    ref = ops.rope_qkv_backward(dq, dk, dv, cos, sin, nq, nkv, D, S)
    a_ref = torch.zeros(1, device=dev)
    y_ref, t_ref = ops.cast_amax(ref, scale, a_ref, fmt)
    a = torch.zeros(1, device=dev)
    y, yt = ops.rope_qkv_backward_cast(dq, dk, dv, cos, sin, nq, nkv, D, S, scale, a, fmt)
    assert torch.equal(y, y_ref) and torch.equal(yt, t_ref) and torch.equal(a, a_ref) and a.item() > 0
    y2, none = ops.rope_qkv_backward_cast(dq, dk, dv, cos, sin, nq, nkv, D, S, scale, None, fmt, want_t=False)
    assert none is None and torch.equal(y2, y_ref)


@pytest.mark.parametrize("fmt", [O.E5M2, O.E4M3])
def test_ce_backward_cast_is_bitwise_the_two_kernel_sequence(dev, ops, fmt):
    """mi_ce_backward_cast == mi_ce_backward followed by mi_cast_amax (FP8 bytes both ways, amax), ragged tiles, ignored rows."""
    from llm_fp8_amd import _lib
    lib = _lib.load()
    T, V = 200, 1160  # neither a multiple of 128
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(T, V, generator=g) * 3).to(torch.bfloat16).to(dev)
    labels = torch.randint(0, V, (T,), generator=g).to(dev)
    labels[::7] = -100
    st = torch.cuda.current_stream().cuda_stream
    lse = torch.empty(T, dtype=torch.float32, device=dev)
    rows = torch.empty(T, dtype=torch.float32, device=dev)
    _lib.check(lib.mi_ce_forward(logits.data_ptr(), labels.data_ptr(), lse.data_ptr(), rows.data_ptr(), T, V, st), "fwd")
    gscale = torch.tensor([0.37], device=dev)
    d = torch.empty_like(logits)
    _lib.check(lib.mi_ce_backward(logits.data_ptr(), labels.data_ptr(), lse.data_ptr(), gscale.data_ptr(), d.data_ptr(), T, V, st), "bwd")
    scale = torch.tensor([4096.0], device=dev)
    a_ref = torch.zeros(1, device=dev)
    y_ref, t_ref = ops.cast_amax(d, scale, a_ref, fmt)
    y = torch.zeros((T, V), dtype=torch.uint8, device=dev)
    yt = torch.zeros((V, T), dtype=torch.uint8, device=dev)
    a = torch.zeros(1, device=dev)
    _lib.check(lib.mi_ce_backward_cast(logits.data_ptr(), labels.data_ptr(), lse.data_ptr(), gscale.data_ptr(), y.data_ptr(),
                                       yt.data_ptr(), scale.data_ptr(), a.data_ptr(), T, V, fmt, st), "fused")
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(yt, t_ref) and torch.equal(a, a_ref) and a.item() > 0
    assert int((y_ref != 0).sum()) > T  # the comparison is not vacuous


def test_embedding_grad_add_in_place(ops, dev):
    """mi_embedding_grad_add: grad[id] += alpha * sum of the rows of dY with that id -- repeated ids, untouched rows left alone,
    padding / out-of-range ids skipped, result within one bf16 ulp of an fp64 scatter, bitwise reproducible."""
    g = torch.Generator().manual_seed(12)
    V, H, T = 300, 264, 1000
    grad0 = torch.randn(V, H, generator=g).to(torch.bfloat16)
    dy = (torch.randn(T, H, generator=g) * 0.7).to(torch.bfloat16)
    ids = torch.randint(0, 40, (T,), generator=g)            # few distinct ids: long runs of repeats
    ids[::13] = 299
    ids[5] = 7_000                                            # out of range: ignored
    ids[6] = 17                                               # padding_idx below
    alpha = 0.5
    ok = (ids < V) & (ids != 17)
    want = grad0.double().index_add_(0, ids[ok], alpha * dy[ok].double())
    outs = []
    for _ in range(2):
        grad = grad0.clone().to(dev)
        ops.embedding_grad_add_(grad, dy.to(dev), ids.to(dev), alpha=alpha, padding_idx=17)
        outs.append(grad.cpu())
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    got = outs[0].double()
    ulp = torch.maximum(want.abs(), got.abs()).clamp_min(1e-30).log2().floor().exp2() * 2.0 ** -7
    assert float(((got - want).abs() / ulp).max()) <= 1.0
    untouched = torch.ones(V, dtype=torch.bool); untouched[ids[ok]] = False
    assert untouched.sum() > 100 and torch.equal(outs[0][untouched].view(torch.int16), grad0[untouched].view(torch.int16))


@pytest.mark.parametrize("fmt", [O.E5M2, O.E4M3])
@pytest.mark.parametrize("shape", [(2, 128, 24, 8), (1, 96, 4, 2)])
def test_mxfp8_rope_backward_quantize_is_bitwise_the_two_kernel_sequence(ops, dev, fmt, shape):
    """mi_mxfp8_rope_bwd_quantize == mi_rope_qkv(backward) + mi_mxfp8_quantize: data and E8M0 scales, both orientations."""
    B, S, nq, nkv = shape
    D, T = 128, shape[0] * shape[1]
    g = torch.Generator().manual_seed(S * 3 + nq)
    dq = (torch.randn(T, nq * D, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    dk = (torch.randn(T, nkv * D, generator=g) * 30.0).to(torch.bfloat16).to(dev)
    dv = (torch.randn(T, nkv * D, generator=g) * 1e-3).to(torch.bfloat16).to(dev)
    dq[1, 5] = float("nan")
    ang = torch.outer(torch.arange(S, dtype=torch.float32), 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
    cos, sin = torch.cos(ang).contiguous().to(dev), torch.sin(ang).contiguous().to(dev)
    ref = ops.mxfp8_quantize(ops.rope_qkv_backward(dq, dk, dv, cos, sin, nq, nkv, D, S), fmt)
    got = ops.mxfp8_rope_bwd_quantize(dq, dk, dv, cos, sin, nq, nkv, D, S, fmt)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    row_only = ops.mxfp8_rope_bwd_quantize(dq, dk, dv, cos, sin, nq, nkv, D, S, fmt, colwise=False)
    assert row_only[2] is None and torch.equal(row_only[0], ref[0]) and torch.equal(row_only[1], ref[1])


def test_colsum_finish_multi_equals_single_finishes(ops, dev):
    g = torch.Generator().manual_seed(5)
    items = [(torch.randn(64, 3072, generator=g).to(dev), torch.bfloat16), (torch.randn(64, 16384, generator=g).to(dev), torch.bfloat16),
             (torch.randn(512, 3072, generator=g).to(dev), torch.float32), (torch.randn(3, 40, generator=g).to(dev), torch.bfloat16),
             (torch.randn(9, 33, generator=g).to(dev), torch.float32), (torch.randn(7, 64, generator=g).to(dev), torch.float16)]
    multi = ops.colsum_finish_multi(items)
    for (part, dt), got in zip(items, multi):
        want = ops.colsum_finish(part, dt)
        assert got.dtype == dt and torch.equal(got, want)


# ----------------------------------------------------------------------------------------- K9 RMSNorm -> FP8
@pytest.mark.parametrize("shape", [(8, 512), (137, 1024), (8192, 3072), (3, 8192), (64, 4104)])
def test_add_rmsnorm_stats_matches_add_then_stats(ops, dev, shape):
    """Residual add folded into the statistics pass: the sum is the bf16 add bit for bit, rstd that of the ROUNDED sum."""
    R, C = shape
    g = torch.Generator().manual_seed(R * 7 + C)
    a = (torch.randn(R, C, generator=g) * torch.exp(torch.randn(R, 1, generator=g))).to(torch.bfloat16).to(dev)
    b = torch.randn(R, C, generator=g).to(torch.bfloat16).to(dev)
    out, rstd = ops.add_rmsnorm_stats(a, b, 1e-5)
    want = a + b
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    _, rstd_ref = O.rmsnorm_f32(bf16_bits(want.cpu()), bf16_bits(torch.ones(C, dtype=torch.bfloat16)), 1e-5)
    np.testing.assert_allclose(rstd.cpu().numpy(), rstd_ref, rtol=2e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), ops.rmsnorm_stats(want, 1e-5).cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("shape", [(8, 512), (136, 1024), (1024, 3072), (64, 4096), (4104, 3072)])  # the last: > 1 tile per wave of the persistent walk
def test_rmsnorm_cast_and_backward_vs_oracle(ops, dev, shape):
    R, C = shape
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g) * torch.exp(torch.randn(R, 1, generator=g))).to(torch.bfloat16)
    gamma = (torch.rand(C, generator=g) + 0.5).to(torch.bfloat16)
    eps = 1e-5
    y_ref, rstd_ref = O.rmsnorm_f32(bf16_bits(x), bf16_bits(gamma), eps)
    rstd = ops.rmsnorm_stats(x.to(dev), eps)
    np.testing.assert_allclose(rstd.cpu().numpy(), rstd_ref, rtol=2e-6)
    scale = np.float32(32.0)
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y8, y8t = ops.norm_cast(x.to(dev), rstd, gamma.to(dev), _f32(scale, dev), amax, O.E4M3)
    want = O.fp8_encode_sat((y_ref * scale).astype(np.float32), O.E4M3)
    _ulp_close_fp8(u8(y8), want, O.E4M3)
    # float32 restatements in the kernels' operation order.  Statistics: same summation order, so only v_rsq_f32's
    # approximation separates the two (<= 2 ulp).  Cast: no transcendental at all -- with the device's own rstd as input
    # the bytes must be IDENTICAL to the bytes of (x * rstd) * gamma * scale evaluated in float32.
    rstd32 = O.rmsnorm_rstd_device_order(bf16_bits(x), eps)
    ulp = np.abs(rstd.cpu().numpy().view(np.int32) - rstd32.view(np.int32))
    assert ulp.max() <= 2, f"rstd differs from the float32 restatement by {ulp.max()} ulp"
    y32 = O.norm_apply_f32_device_order(bf16_bits(x), rstd.cpu().numpy(), bf16_bits(gamma))
    np.testing.assert_array_equal(u8(y8), O.fp8_encode_sat((y32 * scale).astype(np.float32), O.E4M3))
    assert amax.item() == np.abs(y32).max()
    np.testing.assert_array_equal(u8(y8t), u8(y8).T)
    np.testing.assert_allclose(amax.item(), np.abs(y_ref).max(), rtol=1e-5)
    dy = (torch.randn(R, C, generator=g) / 8).to(torch.bfloat16)
    dx, dgam = ops.rmsnorm_bwd(dy.to(dev), x.to(dev), rstd, gamma.to(dev))
    dx_ref, dg_ref = O.rmsnorm_bwd_f32(bf16_bits(dy), bf16_bits(x), bf16_bits(gamma), eps)
    assert np.all(np.abs(dx.float().cpu().numpy() - dx_ref) <= 2.0 ** -7 * np.abs(dx_ref) + 1e-5 * np.abs(dx_ref).max())
    np.testing.assert_allclose(dgam.cpu().numpy(), dg_ref, rtol=1e-4, atol=1e-4 * np.abs(dg_ref).max())
    dx2, dgam2 = ops.rmsnorm_bwd(dy.to(dev), x.to(dev), rstd, gamma.to(dev))
    assert torch.equal(dgam, dgam2) and torch.equal(dx, dx2)  # fixed-order partial sums: reproducible


@pytest.mark.parametrize("shape", [(32, 512), (160, 1024), (1024, 3072)])
def test_mxfp8_fused_front_ends_vs_oracle(ops, dev, shape):
    """MXFP8 quantiser fused with RMSNorm / SwiGLU / dSwiGLU (config #3 runs on these: /root/reference/te_llama_mxfp8.py:28-29), held
    to the same criterion as the delayed-scaling twins: the kernels evaluate the float32 expressions of the oracle's *_device_order
    restatements, so
      * RMSNorm (no transcendental; the device's own rstd as input): E8M0 scales AND bytes identical, both orientations;
      * SwiGLU / dSwiGLU (device v_exp_f32 vs numpy exp, a few ulps): a block scale may differ only by one step and only where the
        block amax / 448 lies within 2^-17 of a power of two; with the device's scale every mismatching byte must belong to a value
        within 2^-17 (+ the dsilu cancellation slack) of an FP8 rounding boundary."""
    R, C = shape
    g = torch.Generator().manual_seed(R + 3 * C)
    rcp = np.float32(1.0 / 448.0)

    def check(got, v32, name, exact, abs_slack=None):
        y_row, s_row, y_colT, s_colT = got
        v32 = np.ascontiguousarray(v32, dtype=np.float32)
        for data, scales, mat, slk in ((y_row, s_row, v32, abs_slack),
                                       (y_colT, s_colT, np.ascontiguousarray(v32.T), None if abs_slack is None else np.ascontiguousarray(abs_slack.T))):
            r, c = mat.shape
            blk = np.abs(mat).reshape(r, c // 32, 32).max(-1).astype(np.float32)
            want_e = O.float_to_e8m0_roundup((blk * rcp).astype(np.float32))
            got_e = u8(scales).T
            mism = got_e != want_e
            if exact:
                assert not mism.any(), f"{name}: {mism.sum()} block scales differ from the float32 restatement"
            else:
                v = (blk * rcp).astype(np.float64)[mism]
                step = np.abs(got_e.astype(np.int32) - want_e.astype(np.int32))[mism]
                p2 = np.exp2(np.round(np.log2(np.maximum(v, 1e-300))))
                assert (step == 1).all() and (np.abs(v / p2 - 1.0) <= 2.0 ** -17).all(), f"{name}: a block scale differs away from a power-of-two boundary"
                assert mism.sum() <= max(2, 1e-4 * mism.size), f"{name}: {mism.sum()} of {mism.size} block scales differ"
            inv = np.ldexp(np.float32(1.0), 127 - got_e.astype(np.int64)).astype(np.float32)   # the DEVICE's scale
            v_scaled = (mat.reshape(r, c // 32, 32) * inv[:, :, None]).astype(np.float32).reshape(r, c)
            if exact:
                np.testing.assert_array_equal(u8(data), O.fp8_encode_sat(v_scaled, O.E4M3), err_msg=name)
            else:
                slack = None if slk is None else (slk.reshape(r, c // 32, 32) * inv[:, :, None]).reshape(r, c)
                _assert_matches_fp32_restatement(u8(data), v_scaled, O.E4M3, name, abs_slack=slack)

    x = (torch.randn(R, C, generator=g) * torch.exp(torch.randn(R, 1, generator=g))).to(torch.bfloat16)
    gamma = (torch.rand(C, generator=g) + 0.5).to(torch.bfloat16)
    rstd = ops.rmsnorm_stats(x.to(dev), 1e-5)
    y32 = O.norm_apply_f32_device_order(bf16_bits(x), rstd.cpu().numpy(), bf16_bits(gamma))
    check(ops.mxfp8_norm_quantize(x.to(dev), rstd, gamma.to(dev)), y32, "mx norm", exact=True)
    y_ref, _ = O.rmsnorm_f32(bf16_bits(x), bf16_bits(gamma), 1e-5)   # and the float64-derived value: the restatement is the same function
    np.testing.assert_allclose(y32, y_ref, rtol=2e-6, atol=0)
    h = (torch.randn(R, 2 * C, generator=g) * 2).to(torch.bfloat16)
    act32 = O.swiglu_f32_device_order(bf16_bits(h))
    check(ops.mxfp8_swiglu_quantize(h.to(dev)), act32, "mx swiglu", exact=False)
    np.testing.assert_allclose(act32, O.swiglu_f32(bf16_bits(h)), rtol=1e-5, atol=1e-30)
    d = (torch.randn(R, C, generator=g) / 8).to(torch.bfloat16)
    out = ops.mxfp8_dswiglu_quantize(h.to(dev), d.to(dev), want_colsum=True)
    dh32 = O.dswiglu_f32_device_order(bf16_bits(h), bf16_bits(d))
    hf, df = O.bf16_bits_to_f32(bf16_bits(h)), O.bf16_bits_to_f32(bf16_bits(d))
    slack = np.concatenate([np.abs(df * hf[:, C:]) * 2.0 ** -21, np.zeros_like(df)], axis=1).astype(np.float32)
    check(out[:4], dh32, "mx dswiglu", exact=False, abs_slack=slack)
    dh = O.dswiglu_f32(bf16_bits(h), bf16_bits(d))
    np.testing.assert_allclose(out[4].sum(0).cpu().numpy(), dh.astype(np.float64).sum(0), rtol=1e-4, atol=1e-4 * np.abs(dh).max() * np.sqrt(R))


def _torch_attn_ref(q, k, v, scale, causal=True):
    """fp32 torch reference of the attention core on the device (full sizes); q [B,S,H,D], k/v [B,S,G,D]."""
    B, S, H, D = q.shape
    G = k.shape[2]
    qf, kf, vf = (t.float().transpose(1, 2) for t in (q, k, v))
    kf, vf = kf.repeat_interleave(H // G, 1), vf.repeat_interleave(H // G, 1)
    s = (qf @ kf.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=q.device), 1), float("-inf"))
    lse2 = torch.logsumexp(s, -1) / np.log(2.0)
    return (torch.softmax(s, -1) @ vf).transpose(1, 2), lse2


@pytest.mark.parametrize("B,S,H,G", [(1, 128, 2, 1), (2, 256, 6, 2), (1, 512, 3, 3)])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("D", [128, 64])
def test_attn_fwd_vs_oracle(ops, dev, B, S, H, G, causal, D):
    g = torch.Generator().manual_seed(S + H)
    q, k, v = (torch.randn(B, S, n, D, generator=g).to(torch.bfloat16) for n in (H, G, G))
    q = q * 2.0  # sharper softmax
    scale = D ** -0.5
    o_ref, lse_ref = O.attention_f64(bf16_bits(q), bf16_bits(k), bf16_bits(v), scale, causal)
    o, lse = ops.attn_fwd(q.to(dev), k.to(dev), v.to(dev), scale, causal)
    # P is rounded to bf16 before P.V and O to bf16 at the end: 2^-8 relative each, on values bounded by max|v|
    np.testing.assert_allclose(o.float().cpu().numpy(), o_ref, rtol=2 ** -6, atol=2 ** -7 * float(v.abs().max()))
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref, rtol=0, atol=2e-3)


def test_attn_fwd_strided_views_and_full_size(ops, dev):
    """Operands as column slices of one fused [tokens, (H + 2G) D] buffer, BASELINE config sizes (3B: B16 S512 H24 G8 D128)."""
    B, S, H, G, D = 16, 512, 24, 8, 128
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(B, S, (H + 2 * G) * D, device=dev, dtype=torch.bfloat16, generator=g)
    q = qkv[..., :H * D].view(B, S, H, D)
    k = qkv[..., H * D:(H + G) * D].view(B, S, G, D)
    v = qkv[..., (H + G) * D:].view(B, S, G, D)
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale, True)
    o_ref, lse_ref = _torch_attn_ref(q, k, v, scale, True)
    assert torch.isfinite(o).all()
    err = (o.float() - o_ref).abs().max().item()
    assert err < 2 ** -6 * float(v.abs().max()), err
    assert (lse - lse_ref).abs().max().item() < 2e-3


@pytest.mark.parametrize("B,S,H,G", [(1, 128, 2, 1), (2, 256, 6, 2), (1, 512, 3, 3)])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("D", [128, 64])
def test_attn_bwd_vs_oracle(ops, dev, B, S, H, G, causal, D):
    g = torch.Generator().manual_seed(S + H + 1)
    q, k, v = (torch.randn(B, S, n, D, generator=g).to(torch.bfloat16) for n in (H, G, G))
    do = (torch.randn(B, S, H, D, generator=g) / 4).to(torch.bfloat16)
    scale = D ** -0.5
    dq_r, dk_r, dv_r = O.attention_bwd_f64(bf16_bits(q), bf16_bits(k), bf16_bits(v), bf16_bits(do), scale, causal)
    qd, kd, vd, dod = (t.to(dev) for t in (q, k, v, do))
    o, lse = ops.attn_fwd(qd, kd, vd, scale, causal)
    dq, dk, dv = ops.attn_bwd(dod, qd, kd, vd, o, lse, scale, causal)
    for got, ref, name in ((dq, dq_r, "dq"), (dk, dk_r, "dk"), (dv, dv_r, "dv")):
        got = got.float().cpu().numpy()
        # P, dS and the outputs are rounded to bf16 (2^-8 each); errors add over the contraction like a random walk
        tol = 2 ** -6 * np.abs(ref) + 2 ** -7 * np.sqrt(np.mean(ref.astype(np.float64) ** 2))
        bad = np.abs(got - ref) > tol
        assert bad.mean() < 1e-3, f"{name}: {bad.sum()} / {bad.size} outside tolerance, max diff {np.abs(got - ref).max():.4g}"
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 6e-3, f"{name}: relative Frobenius error {rel:.4g}"


def test_attn_bwd_full_size_vs_torch_autograd_and_reproducible(ops, dev):
    B, S, H, G, D = 4, 512, 24, 8, 128
    g = torch.Generator(device=dev).manual_seed(2)
    q, k, v = (torch.randn(B, S, n, D, device=dev, dtype=torch.bfloat16, generator=g) for n in (H, G, G))
    do = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16, generator=g) / 4
    scale = D ** -0.5
    o, lse = ops.attn_fwd(q, k, v, scale, True)
    dq, dk, dv = ops.attn_bwd(do, q, k, v, o, lse, scale, True)
    dq2, dk2, dv2 = ops.attn_bwd(do, q, k, v, o, lse, scale, True)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)  # no atomics: bitwise reproducible
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _torch_attn_ref(qr, kr, vr, scale, True)
    o_ref.backward(do.float())
    for got, ref, name in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
        rel = ((got.float() - ref).norm() / ref.norm()).item()
        assert rel < 6e-3, f"{name}: {rel:.4g}"


@pytest.mark.parametrize("shape", [(8192, 3072, 512), (4352, 4096, 768), (8192, 5120, 256), (2560, 8192, 1024)])
@pytest.mark.parametrize("fa,fb", [(O.E4M3, O.E4M3), (O.E5M2, O.E4M3)])
def test_gemm_streamk_vs_oracle(ops, dev, shape, fa, fb):
    """Stream-K form of the persistent kernel (algo 44): tiles split between two workgroups are summed through the
    registered workspace; checked against the oracle, with and without bias, twice (flag epochs), and against whole tiles."""
    M, N, K = shape
    a8 = _rand_fp8((M, K), fa, 11 + M, 4.0 if fa == O.E4M3 else 64.0)
    b8 = _rand_fp8((N, K), fb, 12 + N, 4.0)
    sa, sb = np.float32(1 / 3.1), np.float32(1 / 0.02)
    bias = O.f32_to_bf16_bits(np.random.default_rng(4).normal(size=N).astype(np.float32) * 10)
    ta, tb = torch.from_numpy(a8).to(dev), torch.from_numpy(b8).to(dev)
    # 2^-7 |ref| + 1e-3 rms (bf16 output) plus the MFMA in-instruction truncation bound (see assert_mfma_close)
    mag = (np.abs(O.fp8_decode(a8, fa)).astype(np.float64) @ np.abs(O.fp8_decode(b8, fb)).astype(np.float64).T) * float(sa) * float(sb)
    for use_bias in (False, True, False):
        ref = O.gemm_fp8_tn(a8, b8, fa, fb, sa, sb, bias if use_bias else None, out_f32=True)
        d = ops.gemm_fp8(ta, tb, _f32(sa, dev), _f32(sb, dev), fa, fb, bias=bits_to_bf16(bias, dev) if use_bias else None, algo=44)
        diff = np.abs(d.float().cpu().numpy().astype(np.float64) - ref)
        bad = diff > O.gemm_tolerance(ref) + 7 * 2.0 ** -14 * mag
        assert not bad.any(), f"stream-K {shape} bias {use_bias}: {bad.sum()} outside tolerance, max diff {diff.max():.4g}"
    whole = ops.gemm_fp8(ta, tb, _f32(sa, dev), _f32(sb, dev), fa, fb, algo=45)
    diff = (d.float() - whole.float()).abs()
    assert (diff <= 2 ** -6 * whole.float().abs() + 1e-6).all()  # a bf16 ulp or two from the different summation order


def test_gemm_mxfp8_streamk_vs_oracle(ops, dev):
    M, N, K = 4352, 4096, 512
    g = torch.Generator().manual_seed(9)
    a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K // 32, generator=g).repeat_interleave(32, 1))).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * torch.exp(torch.randn(N, K // 32, generator=g).repeat_interleave(32, 1))).to(torch.bfloat16)
    a8, ae = O.mxfp8_quantize_rowwise(bf16_bits(a))
    b8, be = O.mxfp8_quantize_rowwise(bf16_bits(b))
    ref = O.gemm_mxfp8_tn(a8, ae, b8, be, out_f32=True)
    t = lambda v: torch.from_numpy(v).to(dev)
    tT = lambda v: torch.from_numpy(np.ascontiguousarray(v.T)).to(dev)
    d = ops.gemm_mxfp8(t(a8), tT(ae), t(b8), tT(be), algo=44)
    assert_gemm_close(d.float().cpu().numpy(), ref, "mx stream-K")


@pytest.mark.parametrize("shape", [(8, 16), (200, 136), (1024, 3072)])
def test_cast_colsum_and_finish(ops, dev, shape):
    """mi_cast_amax_colsum: same bytes / amax as mi_cast_amax plus fp32 column sums (the bias gradient)."""
    R, C = shape
    x = (torch.randn(R, C, generator=torch.Generator().manual_seed(R)) * 3).to(torch.bfloat16)
    sc = _f32(0.37, dev)
    a0, a1 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    y0, t0 = ops.cast_amax(x.to(dev), sc, a0, O.E5M2)
    y1, t1, part = ops.cast_amax(x.to(dev), sc, a1, O.E5M2, want_colsum=True)
    assert torch.equal(y0, y1) and torch.equal(t0, t1) and torch.equal(a0, a1)
    assert part.shape == ((R + 63) // 64, C)
    ref = O.bf16_bits_to_f32(bf16_bits(x)).astype(np.float64).sum(0)
    got32 = ops.colsum_finish(part, torch.float32).cpu().numpy()
    np.testing.assert_allclose(got32, ref, rtol=1e-5, atol=1e-4 * np.sqrt(R))
    got16 = ops.colsum_finish(part, torch.bfloat16).float().cpu().numpy()
    np.testing.assert_array_equal(got16, O.bf16_bits_to_f32(O.f32_to_bf16_bits(got32)))


@pytest.mark.parametrize("shape", [(8192, 5120, 3072), (8192, 16384, 512)])
@pytest.mark.parametrize("mx", [False, True])
def test_persistent_gemm_bitwise_reproducible_under_load(ops, dev, mx, shape):
    """Repeated launches of the persistent GEMM on a BASELINE-sized problem give bit-identical outputs (no data race between
    the two wave groups on LDS-staged operands / block scales; a race on the B scales once showed up as run-to-run
    differences of the mxfp8 loss; the second shape gives every workgroup 8 short tiles = many epilogue / first-K-tile seams)."""
    M, N, K = shape
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randint(0, 256, (M, K), generator=g, device=dev, dtype=torch.uint8)
    b = torch.randint(0, 256, (N, K), generator=g, device=dev, dtype=torch.uint8)
    for t in (a, b):
        t[(t & 0x7F) >= 0x78] &= 0x3F
    one = torch.ones(1, device=dev)
    sa = torch.randint(118, 134, (K // 32, M), generator=g, device=dev, dtype=torch.uint8)
    sb = torch.randint(118, 134, (K // 32, N), generator=g, device=dev, dtype=torch.uint8)
    run = (lambda: ops.gemm_mxfp8(a, sa, b, sb, algo=4)) if mx else (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, algo=4))
    ref = run()
    ref_generic = ops.gemm_mxfp8(a, sa, b, sb, algo=1) if mx else ops.gemm_fp8(a, b, one, one, 0, 0, algo=1)
    assert torch.equal(ref, ref_generic) or ((ref.float() - ref_generic.float()).abs() <= 2 ** -6 * ref_generic.float().abs() + 1e-3).all()
    for _ in range(12):
        assert torch.equal(run(), ref)


@pytest.mark.parametrize("shape", [(2048, 4096, 512), (8192, 3072, 1024), (3072, 3072, 768), (768, 3072, 1280), (8192, 8192, 256),
                                   (4352, 4096, 256), (256, 256, 256), (8192, 5120, 3072)])
@pytest.mark.parametrize("algo", [6, 9])
@pytest.mark.parametrize("fa,fb", [(O.E4M3, O.E4M3), (O.E5M2, O.E4M3), (O.E4M3, O.E5M2)])
def test_four_wave_kernels_are_bitwise_the_eight_wave_kernel(ops, dev, shape, algo, fa, fb):
    """mi_gemm_w4.hip (algo 6: one tile per workgroup, algo 9: persistent with the epilogue spread over the tile boundary) computes
    every output element with the same MFMA sequence over K as the eight-wave persistent kernel (algo 4): the results must be
    IDENTICAL bits.  Any difference means an accumulator was read after the next tile's zero-C MFMA overwrote it, a counted vmcnt
    wait let an LDS half or a store's data be reused early, or a cursor staged the wrong panel.  Shapes: 1 - 4 tiles per workgroup,
    K-tile counts 2 (the combined LAST2 K-tile), 4, 6, 8, 10 and 24, tile counts that are not a multiple of the CU count."""
    M, N, K = shape
    if M % 256 or N % 256 or K % 256 or (algo == 9 and K < 512):
        pytest.skip("the four-wave kernels take 256-multiples only (the persistent one K >= 512)")
    g = torch.Generator(device=dev).manual_seed(11 + fa)
    a = torch.randint(0, 256, (M, K), generator=g, device=dev, dtype=torch.uint8)
    b = torch.randint(0, 256, (N, K), generator=g, device=dev, dtype=torch.uint8)
    for t, f in ((a, fa), (b, fb)):
        if f == O.E4M3:
            t[(t & 0x7F) >= 0x78] &= 0x3F
        else:
            t[(t & 0x7F) >= 0x54] &= 0xCF
    sa, sb = torch.full((1,), 0.37, device=dev), torch.full((1,), 1.9, device=dev)
    ref = ops.gemm_fp8(a, b, sa, sb, fa, fb, algo=4 if (M % 256 == 0 and N % 256 == 0) else 0)
    ref40 = ops.gemm_fp8(a, b, sa, sb, fa, fb, algo=40)  # the 256 x 256 shape of the eight-wave kernel, whatever the picker takes
    assert torch.equal(ref, ref40)
    for _ in range(3):
        assert torch.equal(ops.gemm_fp8(a, b, sa, sb, fa, fb, algo=algo), ref)


def test_lab_timing_builds_are_bitwise_the_default(dev):
    """The lab library's epilogue-placement builds (27, 29, 46) and phase schedules of the four-wave kernel (54, 62, 66) move work
    around but keep every fp32 summation order: same bits as the product kernel.  Skipped when the lab library is not built
    (tools/bin/libmi_fp8_lab.so; `make -C llm_fp8_amd/csrc lab`).  The lab library is loaded directly -- the package never does."""
    import ctypes
    from llm_fp8_amd import _lib
    if not os.path.exists(_lib.LAB_LIB_PATH):
        pytest.skip("lab library not built")
    lab = ctypes.CDLL(_lib.LAB_LIB_PATH)
    lab.mi_gemm_fp8.argtypes = _lib.SIGNATURES["mi_gemm_fp8"]
    lab.mi_gemm_fp8.restype = ctypes.c_int
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    for (M, N, K) in ((2048, 4096, 512), (8192, 3072, 1024), (3072, 3072, 768), (768, 3072, 1280)):
        a = torch.randint(0, 256, (M, K), generator=g, device=dev, dtype=torch.uint8)
        b = torch.randint(0, 256, (N, K), generator=g, device=dev, dtype=torch.uint8)
        for t in (a, b):
            t[(t & 0x7F) >= 0x78] &= 0x3F
        sa, sb = torch.full((1,), 0.37, device=dev), torch.full((1,), 1.9, device=dev)
        outs = {}
        for algo in (4, 27, 29, 46) + ((54, 62, 66) if M % 256 == 0 and N % 256 == 0 and K % 256 == 0 else ()):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
            rc = lab.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), sa.data_ptr(), sb.data_ptr(), None, M, N, K, K, K, N, 0, 0, 0, algo, st)
            assert rc == 0, f"algo {algo}: {lab.mi_last_error()}"
            outs[algo] = out
        torch.cuda.synchronize()
        for algo, out in outs.items():
            assert torch.equal(out, outs[4]), f"lab algo {algo} differs from algo 4 on {M}x{N}x{K}"


@pytest.mark.parametrize("fmt", [O.E4M3, O.E5M2])
def test_mxfp8_quantize_row_blocks_and_colsum(ops, dev, fmt):
    """mi_mxfp8_quantize_ex: parts quantised into their row-blocks of a larger operand == quantising the concatenation
    (all four outputs, bit for bit); and the partial column sums it can emit reduce to the column sums of the input."""
    g = torch.Generator().manual_seed(31)
    K = 384
    parts = [(torch.randn(n, K, generator=g) * s).to(torch.bfloat16).to(dev) for n, s in ((256, 1.0), (64, 30.0), (96, 1e-3))]
    N = sum(p.shape[0] for p in parts)
    ref = ops.mxfp8_quantize(torch.cat(parts, 0), fmt)
    w8 = torch.zeros((N, K), dtype=torch.uint8, device=dev)
    sc = torch.zeros((K // 32, N), dtype=torch.uint8, device=dev)
    wt8 = torch.zeros((K, N), dtype=torch.uint8, device=dev)
    sct = torch.zeros((N // 32, K), dtype=torch.uint8, device=dev)
    r = 0
    for p in parts:
        n = p.shape[0]
        ops.mxfp8_quantize(p, fmt, out=(w8[r:r + n], sc[:, r:r + n], wt8[:, r:r + n], sct[r // 32:(r + n) // 32]))
        r += n
    for got, want in zip((w8, sc, wt8, sct), ref):
        assert torch.equal(got, want)
    # row-only / column-only blocks
    w8b = torch.zeros_like(w8); scb = torch.zeros_like(sc)
    ops.mxfp8_quantize(parts[1], fmt, rowwise=True, colwise=False, out=(w8b[256:320], scb[:, 256:320], None, None))
    assert torch.equal(w8b[256:320], ref[0][256:320]) and torch.equal(scb[:, 256:320], ref[1][:, 256:320])
    x = (torch.randn(392 // 8 * 8 * 4, 160, generator=g) * 3).to(torch.bfloat16).to(dev)  # 1568 rows: ragged last 128-row tile
    y_row, s_row, y_colT, s_colT, cs = ops.mxfp8_quantize(x, fmt, want_colsum=True)
    plain = ops.mxfp8_quantize(x, fmt)
    for got, want in zip((y_row, s_row, y_colT, s_colT), plain):
        assert torch.equal(got, want)
    assert cs.shape == ((x.shape[0] + 127) // 128, 160)
    np.testing.assert_allclose(ops.colsum_finish(cs, torch.float32).cpu().numpy(), x.float().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_mxfp8_full_size_properties(ops, dev):
    """fc1-sized activation (8192 x 16384): size-independent properties of the MXFP8 quantiser, checked on the device with
    torch ops only -- tight power-of-two scales, saturation-free bytes, round-trip error bound, the column-wise copy is the
    row-wise quantisation of x^T, and the block-scaled GEMM of a BASELINE shape agrees with an fp32 matmul of the dequantised
    operands."""
    R, C = 8192, 16384
    g = torch.Generator(device=dev).manual_seed(21)
    x = (torch.randn(R, C, device=dev, generator=g) * torch.exp(2 * torch.randn(R, C // 32, device=dev, generator=g)).repeat_interleave(32, 1)).to(torch.bfloat16)
    y, s, yT, sT = ops.mxfp8_quantize(x, O.E4M3)
    tab = dequant_table(O.E4M3, dev)

    def check(data, scales, src):  # data [r, c] u8, scales block-major [c/32, r], src [r, c] bf16
        r, c = src.shape
        amax = src.float().abs().view(r, c // 32, 32).amax(-1)                     # [r, c/32]
        e = scales.t().contiguous().to(torch.int32)                                # [r, c/32]
        pow2 = torch.ldexp(torch.ones_like(amax), e - 127)
        nz = amax > 0
        assert (amax[nz] <= 448.0 * pow2[nz]).all()                                # no saturation ...
        assert (amax[nz] > 224.0 * pow2[nz] * (1 - 2 ** -20)).all()                 # ... and the next smaller power of two would saturate
        assert ((data & 0x7F) != 0x7F).all()                                        # no NaN byte from finite input
        deq = tab[data.long()].view(r, c // 32, 32) * pow2[:, :, None]
        err = (deq - src.float().view(r, c // 32, 32)).abs()
        assert (err <= 2.0 ** -4 * src.float().abs().view(r, c // 32, 32) + 2.0 ** -10 * pow2[:, :, None]).all()

    check(y, s, x)
    check(yT, sT, x.t().contiguous())
    del y, s, yT, sT
    # block-scaled GEMM at a BASELINE shape vs fp32 matmul of the dequantised operands
    M, N, K = 8192, 3072, 3072
    a = (torch.randn(M, K, device=dev, generator=g)).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    a8, sa, _, _ = ops.mxfp8_quantize(a, O.E4M3, colwise=False)
    b8, sb, _, _ = ops.mxfp8_quantize(b, O.E4M3, colwise=False)
    da = tab[a8.long()].view(M, K // 32, 32) * torch.ldexp(torch.ones(M, K // 32, device=dev), sa.t().to(torch.int32) - 127)[:, :, None]
    db = tab[b8.long()].view(N, K // 32, 32) * torch.ldexp(torch.ones(N, K // 32, device=dev), sb.t().to(torch.int32) - 127)[:, :, None]
    ref = da.view(M, K) @ db.view(N, K).t()
    d = ops.gemm_mxfp8(a8, sa, b8, sb)
    rms = ref.pow(2).mean().sqrt().item()
    assert ((d.float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2e-3 * rms).all()


# ----------------------------------------------------------------------------------------- grouped GEMM (dgrad + wgrad in one launch)
GROUPS = [
    [(512, 768, 512), (768, 512, 512)],                       # small pair the four-wave kernel takes too (K = 512: FIRST, SECOND, THIRD, LAST only)
    [(2048, 1024, 768), (1024, 768, 2048), (512, 512, 1280)],  # three problems, K-tile counts 6, 16, 10, on both kernels
    [(512, 768, 256), (768, 256, 512)],                       # a Linear's dgrad [M,K_w] and wgrad [N_w,K_w]: K 256 vs 512
    [(2048, 768, 1024), (1024, 768, 2048)],                   # 192-column tiles
    [(4096, 3072, 3072), (3072, 3072, 4096)],                 # o-proj backward at M = 4096: > 1 round, mixed K
    [(1536, 1536, 256), (768, 384, 1024), (384, 1920, 512)],  # three problems, 192 x 192 tiles
    [(8192, 3072, 8192), (8192, 3072, 8192)],                 # fc2 backward of Llama-3.2-3B at full size
    [(256, 256, 256)],                                        # a single problem, a single tile
]


@pytest.mark.parametrize("group", GROUPS)
@pytest.mark.parametrize("fa,fb", [(O.E4M3, O.E4M3), (O.E5M2, O.E4M3)])
def test_grouped_gemm_is_bitwise_the_separate_launches(ops, dev, group, fa, fb):
    """mi_gemm_fp8_grouped: every problem's output must be BIT FOR BIT what mi_gemm_fp8 (persistent kernel) gives for it alone
    (same MFMA order per output element), whatever the tile shape chosen for the group; small problems also vs the oracle."""
    g = torch.Generator(device=dev).manual_seed(len(group) * 7 + fa)
    probs, refs = [], []
    for (M, N, K) in group:
        a8 = torch.randint(0, 256, (M, K), generator=g, device=dev, dtype=torch.uint8)
        b8 = torch.randint(0, 256, (N, K), generator=g, device=dev, dtype=torch.uint8)
        for t, f in ((a8, fa), (b8, fb)):
            if f == O.E4M3:
                t[(t & 0x7F) >= 0x68] &= 0xBF
            else:
                t[(t & 0x7F) >= 0x54] &= 0xCF
        sa = torch.rand(1, generator=g, device=dev) + 0.5
        sb = torch.rand(1, generator=g, device=dev) * 0.01 + 0.001
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        probs.append((a8, b8, sa, sb, out))
    ops.gemm_fp8_grouped(probs, fa, fb)
    for cfg in (-1, 3, 4):  # the chosen tile shape; where it divides everything the 192 x 192 one; the four-wave kernel (256 x 256, K >= 512)
        if cfg == 3 and any(M % 192 or N % 192 for M, N, K in group):
            continue
        if cfg == 4 and any(M % 256 or N % 256 or K < 512 for M, N, K in group):
            continue
        if cfg >= 3:
            for p in probs:
                p[4].fill_(float("nan"))
            ops.gemm_fp8_grouped(probs, fa, fb, tile_cfg=cfg)
        for (a8, b8, sa, sb, out), (M, N, K) in zip(probs, group):
            alone = ops.gemm_fp8(a8, b8, sa, sb, fa, fb, algo=4)
            assert torch.equal(out.view(torch.int16), alone.view(torch.int16)), f"problem {M}x{N}x{K} (tile cfg {cfg}) differs from its own launch"
            if M * N * K <= 768 * 1024 * 1024:
                ref = O.gemm_fp8_tn(a8.cpu().numpy(), b8.cpu().numpy(), fa, fb, np.float32(sa.item()), np.float32(sb.item()), None, out_f32=True)
                assert_gemm_close(out.float().cpu().numpy(), ref, f"grouped {M}x{N}x{K}")


def test_grouped_gemm_rejects_what_it_cannot_take(ops, dev):
    a = torch.zeros((256, 128), dtype=torch.uint8, device=dev)  # K = 128: not a multiple of 256
    o = torch.empty((256, 256), dtype=torch.bfloat16, device=dev)
    one = torch.ones(1, device=dev)
    with pytest.raises(RuntimeError, match="does not fit tile shape|no tile shape"):
        ops.gemm_fp8_grouped([(a, a, one, one, o)], 0, 0)
    assert not ops.grouped_gemm_ok([(256, 256, 128)]) and ops.grouped_gemm_ok([(8192, 3072, 3072), (3072, 3072, 8192)])


@pytest.mark.parametrize("shape,parts", [((256, 512), (256,)), ((1024, 384), (512, 256, 256)), ((160, 96), (96, 64))])
def test_adamw_mxcast_emits_the_bytes_of_the_quantiser(ops, dev, shape, parts):
    """mi_adamw_mxcast_bf16_multi: the updated bf16 weight is the one mi_adamw_bf16_multi produces, and the MXFP8 copies it emits
    (row blocks + E8M0, column blocks transposed + E8M0, each part a row-block of one operand) are bitwise mi_mxfp8_quantize of
    that updated weight."""
    from llm_fp8_amd import _lib
    lib = _lib.load()
    N, K = shape
    g = torch.Generator(device=dev).manual_seed(3)
    ws = [torch.randn((n, K), generator=g, device=dev).to(torch.bfloat16) for n in parts]
    gs = [(torch.randn((n, K), generator=g, device=dev) * 1e-2).to(torch.bfloat16) for n in parts]
    ms = [(torch.randn((n, K), generator=g, device=dev) * 1e-3).to(torch.bfloat16) for n in parts]
    vs = [(torch.rand((n, K), generator=g, device=dev) * 1e-4).to(torch.bfloat16) for n in parts]
    ref = [t.clone() for t in ws], [t.clone() for t in ms], [t.clone() for t in vs]
    st = torch.cuda.current_stream().cuda_stream
    hp = dict(lr=1e-2, b1=0.9, b2=0.999, eps=1e-8, wd=0.01, step=3)

    def table(rows):
        return torch.tensor(rows, dtype=torch.int64).to(dev)

    # reference: plain multi-tensor update, then the quantiser
    T = len(parts)
    chunks = torch.tensor([(t, c) for t in range(T) for c in range((ref[0][t].numel() + 65535) // 65536)], dtype=torch.int32).to(dev)
    tab = table([[t.data_ptr() for t in ref[0]], [t.data_ptr() for t in gs], [t.data_ptr() for t in ref[1]], [t.data_ptr() for t in ref[2]],
                 [t.numel() for t in ref[0]]])
    _lib.check(lib.mi_adamw_bf16_multi(tab.data_ptr(), T, chunks.data_ptr(), chunks.shape[0], 65536, None, hp["lr"], hp["b1"], hp["b2"],
                                        hp["eps"], hp["wd"], hp["step"], st), "ref")
    q_ref = ops.mxfp8_quantize(torch.cat(ref[0], 0), 0, rowwise=True, colwise=True)
    # under test
    w8 = torch.zeros((N, K), dtype=torch.uint8, device=dev)
    sc = torch.zeros((K // 32, N), dtype=torch.uint8, device=dev)
    wt8 = torch.zeros((K, N), dtype=torch.uint8, device=dev)
    sct = torch.zeros((N // 32, K), dtype=torch.uint8, device=dev)
    rows = [[t.data_ptr() for t in ws], [t.data_ptr() for t in gs], [t.data_ptr() for t in ms], [t.data_ptr() for t in vs],
            [t.numel() for t in ws], [K] * T, [], [], [], [], [N] * T, [0] * T]
    r0, cks = 0, []
    for t, n in enumerate(parts):
        rows[6].append(w8.data_ptr() + r0 * K)
        rows[7].append(sc.data_ptr() + r0)
        rows[8].append(wt8.data_ptr() + r0)
        rows[9].append(sct.data_ptr() + (r0 // 32) * K)
        cks += [(t, c) for c in range(((n + 127) // 128) * ((K + 127) // 128))]
        r0 += n
    tab2, chunks2 = table(rows), torch.tensor(cks, dtype=torch.int32).to(dev)
    _lib.check(lib.mi_adamw_mxcast_bf16_multi(tab2.data_ptr(), T, chunks2.data_ptr(), chunks2.shape[0], 65536, None, hp["lr"], hp["b1"],
                                               hp["b2"], hp["eps"], hp["wd"], hp["step"], st), "mxcast")
    torch.cuda.synchronize()
    for a, b in zip(ws + ms + vs, ref[0] + ref[1] + ref[2]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    for got, want, what in zip((w8, sc, wt8, sct), q_ref, ("row data", "row scales", "column data", "column scales")):
        assert torch.equal(got, want), what
