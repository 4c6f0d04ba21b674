"""GPU: the HIP path (through the C ABI) against the committed golden fixtures of tests/golden/ (SURVEY.md 8c G1-G7).
Casts, scales and MXFP8 bytes are bit-exact; GEMM outputs are held to the stated tolerance
|d| <= 2^-7 |ref| + 1e-3 rms(ref) (+ the measured MFMA in-instruction truncation bound 7 * 2^-14 * sum|a b|)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fp8_oracle as O
from tests.util import bits_to_bf16

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FMT = {"e4m3": O.E4M3, "e5m2": O.E5M2}


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope="module")
def ops(dev):
    from llm_fp8_amd.pytorch import ops as ops_
    return ops_


def _f32(v, dev):
    return torch.tensor([float(v)], dtype=torch.float32, device=dev)


def _cast(ops, dev, xb, scale, fmt):
    amax = torch.zeros(1, dtype=torch.float32, device=dev)
    y, yT = ops.cast_amax(bits_to_bf16(xb, dev), _f32(scale, dev), amax, fmt)
    return y.cpu().numpy(), yT.cpu().numpy(), amax.item()


def test_g1_cast_kat(ops, dev):
    g = load("g1_cast_kat.npz")
    for name, fmt in FMT.items():
        for i, s in enumerate(g[f"scales_{name}"]):
            y, yT, amax = _cast(ops, dev, g["x_bits"], s, fmt)
            np.testing.assert_array_equal(y, g[f"y_{name}_{i}"], err_msg=f"{name} scale {s}")
            np.testing.assert_array_equal(yT, g[f"y_{name}_{i}"].T)
            assert amax == float("inf")


@pytest.mark.parametrize("tag", ["s1", "s1e-3", "s1e3"])
def test_g2_random_cast(ops, dev, tag):
    g = load("g2_random_cast.npz")
    for name, fmt in FMT.items():
        for sname in ("unit", "fit", "hot"):
            y, yT, amax = _cast(ops, dev, g[f"x_{tag}"], g[f"scale_{tag}_{name}_{sname}"][0], fmt)
            np.testing.assert_array_equal(y, g[f"y_{tag}_{name}_{sname}"])
            np.testing.assert_array_equal(yT, g[f"y_{tag}_{name}_{sname}"].T)
            assert np.float32(amax) == g[f"amax_{tag}"][0]


@pytest.mark.parametrize("H,algo,margin,fmax,tag", [(16, "max", 0, 448.0, "h16_max_e4m3"), (1024, "most_recent", 0, 57344.0, "h1024_recent_e5m2"),
                                                    (4, "max", 2, 448.0, "h4_max_margin2")])
def test_g3_scale_trajectory(ops, dev, H, algo, margin, fmax, tag):
    g = load("g3_scale_trajectory.npz")
    hist = torch.zeros((H, 3), dtype=torch.float32, device=dev)
    scale = torch.ones(3, dtype=torch.float32, device=dev)
    inv = torch.ones(3, dtype=torch.float32, device=dev)
    fm = torch.full((3,), fmax, dtype=torch.float32, device=dev)
    for i, a in enumerate(g["amax_seq"]):
        hist[0].copy_(torch.from_numpy(a))
        ops.scale_update(hist, scale, inv, fm, margin, algo)
        np.testing.assert_array_equal(scale.cpu().numpy(), g[f"scale_{tag}"][i], err_msg=f"step {i}")
        np.testing.assert_array_equal(inv.cpu().numpy(), g[f"scale_inv_{tag}"][i], err_msg=f"step {i}")
    np.testing.assert_array_equal(hist.cpu().numpy(), g[f"hist_final_{tag}"])


def _gemm_ok(got, ref, mag=None):
    tol = O.gemm_tolerance(ref) + (7 * 2.0 ** -14 * mag if mag is not None else 0.0)
    diff = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert (diff <= tol).all(), f"{(diff > tol).sum()} / {diff.size} outside tolerance, max diff {diff.max():.4g}"


@pytest.mark.parametrize("tag", ["small_e4m3", "small_hybrid", "k3072"])
@pytest.mark.parametrize("algo", [0, 1, 3, 4])
def test_g4_gemm(ops, dev, tag, algo):
    from tests.golden.make_golden import gemm_operands
    g = load("g4_gemm.npz")
    M, N, K, fa, fb, seed = (int(v) for v in g[f"shape_{tag}"])
    if algo in (3, 4) and (M % 256 or N % 256 or K % 256):
        pytest.skip("fast kernels need aligned shapes")
    a, b = gemm_operands(seed, M, N, K, fa, fb)
    sa, sb = g[f"scales_{tag}"]
    bias = bits_to_bf16(g[f"bias_{tag}"], dev) if f"bias_{tag}" in g.files else None
    d = ops.gemm_fp8(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), _f32(sa, dev), _f32(sb, dev), fa, fb, bias=bias, algo=algo)
    mag = (np.abs(O.fp8_decode(a, fa)).astype(np.float64) @ np.abs(O.fp8_decode(b, fb)).astype(np.float64).T) * float(sa) * float(sb)
    _gemm_ok(d.float().cpu().numpy(), g[f"d_{tag}"], mag)


@pytest.mark.parametrize("tag,fmt_name", [("hybrid", "HYBRID"), ("e4m3", "E4M3")])
def test_g5_linear_three_steps(dev, tag, fmt_name):
    import llm_fp8_amd.pytorch as te
    from llm_fp8_amd.common.recipe import DelayedScaling, Format
    from llm_fp8_amd.pytorch.fp8 import FP8GlobalStateManager
    FP8GlobalStateManager.reset()
    g = load("g5_linear_steps.npz")
    N, K = g["w"].shape
    recipe = DelayedScaling(fp8_format=getattr(Format, fmt_name), amax_history_len=16, amax_compute_algo="max")
    lin = te.Linear(K, N, bias=True, params_dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        lin.weight.copy_(bits_to_bf16(g["w"], dev))
        lin.bias.copy_(bits_to_bf16(g["bias"], dev))
    f = lambda t: t.detach().float().cpu().numpy()
    for i in range(3):
        x = bits_to_bf16(g["x"][i], dev).requires_grad_(True)
        with te.fp8_autocast(enabled=True, fp8_recipe=recipe):
            y = lin(x)
        y.backward(bits_to_bf16(g["dy"][i], dev))
        _gemm_ok(f(y), O.bf16_bits_to_f32(g[f"{tag}_y{i}"]))
        _gemm_ok(f(x.grad), O.bf16_bits_to_f32(g[f"{tag}_dx{i}"]))
        _gemm_ok(f(lin.weight.grad), O.bf16_bits_to_f32(g[f"{tag}_dw{i}"]))
        np.testing.assert_allclose(f(lin.bias.grad), O.bf16_bits_to_f32(g[f"{tag}_db{i}"]), rtol=2 ** -7, atol=1e-3)
        lin.weight.grad = lin.bias.grad = None
        np.testing.assert_array_equal(lin._meta_fwd.state()["scale"].cpu().numpy()[:2], g[f"{tag}_scale_fwd{i}"][:2])
        np.testing.assert_array_equal(lin._meta_bwd.state()["scale"].cpu().numpy()[:1], g[f"{tag}_scale_bwd{i}"][:1])
    np.testing.assert_array_equal(lin._meta_fwd.state()["amax_history"].cpu().numpy()[:, :2], g[f"{tag}_hist_fwd"][:, :2])
    np.testing.assert_array_equal(lin._meta_bwd.state()["amax_history"].cpu().numpy()[:, :1], g[f"{tag}_hist_bwd"][:, :1])
    FP8GlobalStateManager.reset()


@pytest.mark.parametrize("tag", ["s1", "s1e-3", "s1e3"])
def test_g6_mxfp8(ops, dev, tag):
    g, g2 = load("g6_mxfp8.npz"), load("g2_random_cast.npz")
    x = bits_to_bf16(g2[f"x_{tag}"], dev)
    y_row, s_row, y_colT, s_colT = ops.mxfp8_quantize(x, O.E4M3)
    np.testing.assert_array_equal(y_row.cpu().numpy(), g[f"row_y_{tag}"])
    np.testing.assert_array_equal(s_row.cpu().numpy().T, g[f"row_e_{tag}"])  # device scales are block-major [K/32, rows]
    np.testing.assert_array_equal(y_colT.cpu().numpy(), g[f"col_y_{tag}"])
    np.testing.assert_array_equal(s_colT.cpu().numpy().T, g[f"col_e_{tag}"])
    if tag == "s1":
        w8, ws, _, _ = ops.mxfp8_quantize(bits_to_bf16(g["gemm_w_bits"], dev), O.E4M3, colwise=False)
        for algo in (0, 1):
            d = ops.gemm_mxfp8(y_row, s_row, w8, ws, algo=algo)
            _gemm_ok(d.float().cpu().numpy(), g["gemm_d"])


@pytest.mark.parametrize("scenario", ["mxfp8", "default"])
def test_g7_fp8_model_tracks_hf_bf16_golden_loss(dev, scenario):
    """Same seeded HF weights (built on the CPU exactly as the fixture generator does), converted with the replace_params
    mapping, trained on the golden batch in FP8 on the device.  The first loss matches the HF bf16 CPU loss within 2 %.
    MXFP8 (stateless scales) then follows the golden curve 8.65 -> 2.47 -> 0.07 closely; delayed scaling starts from
    scale 1 / empty history (TE semantics, SURVEY 8b "Module state"), so its first backward underflows most E4M3 grads
    and the curve lags by about one step (measured 8.60 -> 5.44 -> 3.13 -> 0.83)."""
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    from llm_fp8_amd import llama, train
    from llm_fp8_amd.pytorch.fp8 import FP8GlobalStateManager
    FP8GlobalStateManager.reset()
    meta = json.load(open(os.path.join(GOLD, "meta.json")))["g7"]
    config = llama.llama_config("llama-3.2-1b", num_hidden_layers=2, vocab_size=4096)
    torch.manual_seed(42)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        hf = LlamaForCausalLM(config)
    finally:
        torch.set_default_dtype(prev)
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=1, max_seq_length=128, mixed_precision="fp8", use_te=True,
                               fp8_scenario=scenario, num_hidden_layers=2, vocab_size=4096, num_warmup_steps=0, learning_rate=1e-3)
    model = llama.TELlamaForCausalLM.from_hf_state_dict(hf.state_dict(), config, scenario).to(dev)
    model = train.prepare_model(model, cfg)
    model.train()
    ids = torch.tensor(meta["input_ids"], device=dev)
    batch = {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids.clone()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(4):
        out = model(**batch)
        out.loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(out.loss.detach()))
    assert abs(losses[0] - meta["loss"][0]) < 0.02 * meta["loss"][0], (losses, meta["loss"])
    if scenario == "mxfp8":
        assert abs(losses[1] - meta["loss"][1]) < 0.35 * meta["loss"][1] and losses[2] < 0.05 * losses[0], (losses, meta["loss"])
    else:
        assert losses[1] < losses[0] and losses[3] < 0.25 * losses[0], (losses, meta["loss"])
    FP8GlobalStateManager.reset()
