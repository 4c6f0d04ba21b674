"""CPU tests of the oracle itself (oracle/fp8_oracle.py).  PARITY UNPINNED w.r.t. the reference: it
ships no fixtures for this path; the oracle is pinned here against (a) hand-derived OCP FP8 / MX
known answers and (b) torch's independent float8 casts."""
import numpy as np
import pytest
import torch

from oracle import fp8_oracle as O


def _bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def test_decode_tables_match_torch():
    b = torch.arange(256, dtype=torch.uint8)
    for fmt, dt in ((O.E4M3, torch.float8_e4m3fn), (O.E5M2, torch.float8_e5m2)):
        ref = b.view(dt).float().numpy()
        got = O.fp8_decode(b.numpy(), fmt)
        np.testing.assert_array_equal(np.isnan(ref), np.isnan(got))
        np.testing.assert_array_equal(ref[~np.isnan(ref)], got[~np.isnan(got)])


def test_encode_known_answers_e4m3():
    # value -> byte (OCP E4M3FN, RNE, saturating)
    kat = {0.0: 0x00, 1.0: 0x38, 17.0: 0x58, 19.0: 0x5A, 447.0: 0x7E, 448.0: 0x7E, 449.0: 0x7E,
           480.0: 0x7E, 1e5: 0x7E, float("inf"): 0x7E, -float("inf"): 0xFE, 2.0 ** -9: 0x01, 2.0 ** -10: 0x00,
           1.5 * 2.0 ** -10: 0x01, 3 * 2.0 ** -10: 0x02, 2.0 ** -6: 0x08, 0.9375 * 2.0 ** -6: 0x08,
           float("nan"): 0x7F}
    v = np.array(list(kat.keys()), dtype=np.float32)
    np.testing.assert_array_equal(O.fp8_encode_sat(v, O.E4M3), np.array(list(kat.values()), dtype=np.uint8))
    assert O.fp8_encode_sat(np.array([-0.0], np.float32), O.E4M3)[0] == 0x80


def test_encode_known_answers_e5m2():
    kat = {0.0: 0x00, 1.0: 0x3C, 57344.0: 0x7B, 61440.0: 0x7B, 1e9: 0x7B, float("inf"): 0x7B,
           -float("inf"): 0xFB, 2.0 ** -16: 0x01, 2.0 ** -17: 0x00, 1.5 * 2.0 ** -17: 0x01, 2.0 ** -14: 0x04,
           float("nan"): 0x7F, 1.125: 0x3C, 1.375: 0x3E, 1.25: 0x3D}
    v = np.array(list(kat.keys()), dtype=np.float32)
    np.testing.assert_array_equal(O.fp8_encode_sat(v, O.E5M2), np.array(list(kat.values()), dtype=np.uint8))


@pytest.mark.parametrize("fmt,dt", [(O.E4M3, torch.float8_e4m3fn), (O.E5M2, torch.float8_e5m2)])
def test_encode_matches_torch_on_all_bf16_and_random_f32(fmt, dt):
    mx = float(O.FP8_MAX[fmt])
    allbf = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()
    g = torch.Generator().manual_seed(0)
    rnd = torch.randn(200000, generator=g) * torch.exp(torch.randn(200000, generator=g) * 6)
    for x in (allbf, rnd, rnd * 1e-3, rnd * 1e3):
        finite = ~torch.isnan(x)
        ref = x.clamp(-mx, mx).to(dt).view(torch.uint8).numpy()
        got = O.fp8_encode_sat(x.numpy(), fmt)
        np.testing.assert_array_equal(ref[finite.numpy()], got[finite.numpy()])
        assert np.all(got[~finite.numpy()] == 0x7F)


def test_bf16_round_matches_torch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(100000, generator=g) * 37
    np.testing.assert_array_equal(O.f32_to_bf16_bits(x.numpy()), _bf16_bits(x))


def test_quantize_delayed_amax_and_scale():
    x = torch.tensor([[0.5, -3.0, float("nan"), 2.0] * 2] * 8)
    q, amax = O.quantize_delayed(_bf16_bits(x), np.float32(2.0), O.E4M3)
    assert amax == np.float32(3.0)
    assert q[0, 0] == 0x38 and q[0, 1] == 0xCC and q[0, 2] == 0x7F and q[0, 3] == 0x48


def test_scale_update_known_trajectory():
    H = 4
    hist = np.zeros((H, 2), np.float32)
    scale = np.ones(2, np.float32)
    seq = [(2.0, 0.0), (0.0, 0.0), (8.0, np.inf), (1.0, 3.0), (0.5, 0.0), (0.25, 0.0), (0.125, 0.0), (0.0625, 0.0)]
    exp_max = []
    for a0, a1 in seq:
        hist[0] = (a0, a1)
        hist, scale, inv = O.scale_update(hist, scale, np.float32(448.0), 0, "max")
        exp_max.append(scale.copy())
        assert hist[0, 0] == 0 and hist[0, 1] == 0
        np.testing.assert_array_equal(inv, np.float32(1) / scale)
    # slot 0, hand-traced: roll(-1) moves row 0 to row H-1 and zeroes the new row 0 (drops old row 1)
    np.testing.assert_allclose([s[0] for s in exp_max], [224, 224, 56, 56, 56, 56, 448, 896])
    # slot 1: 0 -> keep 1; inf -> keep; then 3 in window for 3 more steps
    np.testing.assert_allclose([s[1] for s in exp_max][:4], [1, 1, 1, 1])


def test_scale_update_most_recent_and_margin():
    hist = np.zeros((8, 1), np.float32)
    hist[0, 0] = 7.0
    hist[3, 0] = 100.0
    _, s, _ = O.scale_update(hist, np.ones(1, np.float32), np.float32(448.0), 1, "most_recent")
    assert s[0] == np.float32(448.0 / 7.0) / 2


def test_e8m0_roundup():
    v = np.array([1.0, 1.0000001, 0.75, 2.0 ** -127, 1.5 * 2.0 ** -127, 0.0, np.inf, np.nan, 2.0 ** 127, 3e38],
                 dtype=np.float32)
    np.testing.assert_array_equal(O.float_to_e8m0_roundup(v), [127, 128, 127, 0, 1, 0, 0xFE, 0xFF, 254, 254])


def test_mxfp8_quantize_roundtrip_bound_and_layout():
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(64, 96, generator=g) * torch.exp(torch.randn(64, 1, generator=g) * 3)).to(torch.bfloat16)
    bits = _bf16_bits(x)
    q, e = O.mxfp8_quantize_rowwise(bits)
    assert q.shape == (64, 96) and e.shape == (64, 3)
    deq = O.fp8_decode(q, O.E4M3) * np.repeat(O.e8m0_to_f32(e), 32, axis=1)
    xf = x.float().numpy()
    amax_b = np.abs(xf).reshape(64, 3, 32).max(-1)
    # scaled values never exceed 448, and the block max uses the top binade (>= 224)
    scaled_max = amax_b / O.e8m0_to_f32(e)
    assert np.all(scaled_max <= 448.0) and np.all(scaled_max[amax_b > 0] > 224.0 * (1 - 2 ** -8))
    assert np.all(np.abs(deq - xf) <= np.repeat(O.e8m0_to_f32(e), 32, axis=1) * 16 + 1e-30)  # half-ulp at top binade = 16
    qc, ec = O.mxfp8_quantize_colwise(bits)
    q2, e2 = O.mxfp8_quantize_rowwise(np.ascontiguousarray(bits.T))
    np.testing.assert_array_equal(qc, q2)
    np.testing.assert_array_equal(ec, e2)


def test_gemm_oracle_against_float_matmul():
    g = torch.Generator().manual_seed(3)
    a = torch.randn(32, 64, generator=g)
    b = torch.randn(48, 64, generator=g)
    a8 = O.fp8_encode_sat(a.numpy(), O.E4M3)
    b8 = O.fp8_encode_sat(b.numpy(), O.E5M2)
    d = O.gemm_fp8_tn(a8, b8, O.E4M3, O.E5M2, 0.5, 2.0, out_f32=True)
    ref = O.fp8_decode(a8, O.E4M3).astype(np.float64) @ O.fp8_decode(b8, O.E5M2).astype(np.float64).T
    np.testing.assert_allclose(d, ref, rtol=1e-6)


def test_delayed_linear_oracle_runs_three_steps():
    g = torch.Generator().manual_seed(4)
    lin = O.DelayedLinearOracle(O.E4M3, O.E5M2, history_len=4)
    w = _bf16_bits(torch.randn(32, 64, generator=g) * 0.02)
    for step in range(3):
        x = _bf16_bits(torch.randn(16, 64, generator=g))
        dy = _bf16_bits(torch.randn(16, 32, generator=g) / 32)
        y = lin.forward(x, w)
        lin.end_forward()
        dx, dw, db = lin.backward(dy)
        lin.end_backward()
        assert y.shape == (16, 32) and dx.shape == (16, 64) and dw.shape == (32, 64) and db.shape == (32,)
    assert lin.s_fwd[0] > 1 and lin.s_fwd[1] > 1000 and lin.s_bwd[0] > 1000
    # relative error of the fp8 forward vs the float matmul stays at the few-percent level
    xf, wf = O.bf16_bits_to_f32(x), O.bf16_bits_to_f32(w)
    ref = xf @ wf.T
    err = np.linalg.norm(O.bf16_bits_to_f32(y) - ref) / np.linalg.norm(ref)
    assert err < 0.08, err


def test_device_order_restatements_agree_with_the_float64_ones():
    """The float32, kernel-order restatements of the fused front ends (used to pin the fused kernels byte for byte except at
    rounding boundaries) are the same functions as the float64 ones, to float32 accuracy."""
    rng = np.random.default_rng(5)
    h = O.f32_to_bf16_bits((rng.normal(size=(64, 256)) * 2).astype(np.float32))
    d = O.f32_to_bf16_bits((rng.normal(size=(64, 128)) / 8).astype(np.float32))
    a, b = O.swiglu_f32_device_order(h), O.swiglu_f32(h)
    assert np.all(np.abs(a - b) <= 4e-6 * np.maximum(np.abs(b), 1e-3))
    a, b = O.dswiglu_f32_device_order(h, d), O.dswiglu_f32(h, d)
    assert np.all(np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-3))
    x = O.f32_to_bf16_bits((rng.normal(size=(9, 1032)) * np.exp(rng.normal(size=(9, 1)))).astype(np.float32))  # 1032: a ragged last sweep
    gam = O.f32_to_bf16_bits((rng.random(1032) + 0.5).astype(np.float32))
    y64, rstd64 = O.rmsnorm_f32(x, gam, 1e-5)
    rstd32 = O.rmsnorm_rstd_device_order(x, 1e-5)
    np.testing.assert_allclose(rstd32, rstd64, rtol=1e-6)
    y32 = O.norm_apply_f32_device_order(x, rstd32, gam)
    np.testing.assert_allclose(y32, y64, rtol=2e-6, atol=1e-7)


def test_fp8_mismatch_classifier():
    v = np.array([1.0, 17.0, 17.0, 17.0, 98.0], np.float32)  # 17 is the midpoint of the E4M3 codes 16 and 18
    want = O.fp8_encode_sat(v, O.E4M3)
    got = want.copy()
    assert O.fp8_mismatches_near_boundary(got, v, O.E4M3) == (0, 0)
    got[1] = O.fp8_encode_sat(np.array([18.0], np.float32), O.E4M3)[0]   # neighbour code, value ON the boundary: explained
    assert O.fp8_mismatches_near_boundary(got, v, O.E4M3) == (1, 0)
    got[4] = O.fp8_encode_sat(np.array([104.0], np.float32), O.E4M3)[0]  # neighbour code of 96 (98 -> 96), but 98 is no boundary (100 is)
    assert O.fp8_mismatches_near_boundary(got, v, O.E4M3) == (2, 1)
    assert O.fp8_mismatches_near_boundary(got, v, O.E4M3, abs_slack=np.full(5, 10.0)) == (2, 0)
