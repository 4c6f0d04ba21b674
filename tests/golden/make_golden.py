#!/usr/bin/env python3
"""Generates the golden fixtures G1-G7 of SURVEY.md 8(c) into tests/golden/*.npz|json.

    python tests/golden/make_golden.py            (run in the build container; CPU only, ~1 min)

The reference holds no fixture for this path and its FP8 arithmetic (Transformer Engine) is not importable
here, so every expected value below is produced by the numpy restatement in oracle/fp8_oracle.py AND, wherever an
independent implementation exists in the container, asserted against it before it is written:

  * casts (G1, G2)        torch.float8_e4m3fn / float8_e5m2 casts of the clamped product (independent RNE encoder)
  * GEMM (G4)             torch._scaled_mm on CPU (independent fp8 x fp8 -> bf16 with per-tensor scales)
  * E8M0 scales (G6)      math.frexp-based round-up of amax/448, written separately from the oracle's bit version
  * HF bf16 model (G7)    transformers' LlamaForCausalLM itself = the reference's no-TE path (train_fp8.py:118-124)

G3 (scale trajectory) is hand-traceable: the "max" column for slot 0 is checked against literal numbers.
Fixtures are data only (inputs, expected outputs); nothing here is copied from the reference.
"""
from __future__ import annotations

import hashlib
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import fp8_oracle as O  # noqa: E402

TORCH_DT = {O.E4M3: torch.float8_e4m3fn, O.E5M2: torch.float8_e5m2}


def bits(t: torch.Tensor) -> np.ndarray:
    return t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def torch_sat_cast(x_f32: np.ndarray, scale: np.float32, fmt: int) -> np.ndarray:
    """Independent encoder: clamp the fp32 product to +-max, let torch round it; NaN -> 0x7F."""
    with np.errstate(over="ignore", invalid="ignore"):
        y = (x_f32 * np.float32(scale)).astype(np.float32)
    fmax = float(O.FP8_MAX[fmt])
    t = torch.from_numpy(np.clip(y, -fmax, fmax))
    b = t.to(TORCH_DT[fmt]).view(torch.uint8).numpy().copy()
    b[np.isnan(y)] = 0x7F
    return b


def g1_cast_kat():
    vals = [0.0, -0.0, 2.0 ** -9, 2.0 ** -10, 2.0 ** -16, 2.0 ** -17, 1.0, -1.0, 17.0, 19.0, 447.0, 448.0, 449.0, 464.0, 480.0,
            57344.0, 61440.0, 1e5, -1e5, float("inf"), float("-inf"), float("nan"), 0.0019531, 3.0e-5, 1.5e-5, 7.0e-6, 240.0, -0.3]
    while len(vals) % 128:  # the cast kernel takes rows % 8 == 0
        vals.append(0.0)
    x = torch.tensor(vals, dtype=torch.float32).to(torch.bfloat16).view(-1, 16)
    xb = bits(x)
    xf = O.bf16_bits_to_f32(xb)
    finite_amax = np.float32(np.abs(xf[np.isfinite(xf)]).max())
    out = {"x_bits": xb}
    for fmt, name in ((O.E4M3, "e4m3"), (O.E5M2, "e5m2")):
        scales = np.array([1.0, 0.5, float(O.FP8_MAX[fmt] / finite_amax), 64.0], dtype=np.float32)
        out[f"scales_{name}"] = scales
        for i, s in enumerate(scales):
            q, amax = O.quantize_delayed(xb, s, fmt)
            np.testing.assert_array_equal(q, torch_sat_cast(xf, s, fmt), err_msg=f"G1 {name} scale {s}")
            out[f"y_{name}_{i}"] = q
        out["amax"] = np.array([amax], dtype=np.float32)
    assert np.isinf(out["amax"][0])  # +-inf inputs propagate into amax (NaN ignored)
    np.savez_compressed(os.path.join(HERE, "g1_cast_kat.npz"), **out)


def g2_inputs():
    torch.manual_seed(0)
    base = torch.randn(256, 512)
    return {"s1": bits(base), "s1e-3": bits(base[:64] * 1e-3), "s1e3": bits(base[:64] * 1e3)}


def g2_random_cast():
    out = {}
    for tag, xb in g2_inputs().items():
        xf = O.bf16_bits_to_f32(xb)
        out[f"x_{tag}"] = xb
        amax = O.amax_f32(xf)
        for fmt, name in ((O.E4M3, "e4m3"), (O.E5M2, "e5m2")):
            for sname, s in (("unit", np.float32(1.0)), ("fit", np.float32(O.FP8_MAX[fmt] / amax)), ("hot", np.float32(4.0) * O.FP8_MAX[fmt] / amax)):
                q, qt, a = O.quantize_delayed_transpose(xb, s, fmt)
                np.testing.assert_array_equal(q, torch_sat_cast(xf, s, fmt))
                assert a == amax and np.array_equal(qt, q.T)
                out[f"y_{tag}_{name}_{sname}"] = q
                out[f"scale_{tag}_{name}_{sname}"] = np.array([s], np.float32)
        out[f"amax_{tag}"] = np.array([amax], np.float32)
    np.savez_compressed(os.path.join(HERE, "g2_random_cast.npz"), **out)


def g3_amax_sequence(steps=24):
    seq = np.zeros((steps, 3), np.float32)
    seq[:, 0] = [2, 0, 8, 1, .5, .25, .125, .0625, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 1e-30, 5]           # zeros, decay
    seq[:, 1] = [1, 1, np.inf, 1, 1, 1000, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, np.nan]              # inf, spike, NaN
    seq[:, 2] = [3e38, 1e-38, 1e-45, 448, 57344, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]  # overflow / FLT_MAX guard
    return seq


def g3_scale_trajectory():
    seq = g3_amax_sequence()
    out = {"amax_seq": seq}
    for H, algo, margin, fmax, tag in ((16, "max", 0, 448.0, "h16_max_e4m3"), (1024, "most_recent", 0, 57344.0, "h1024_recent_e5m2"),
                                       (4, "max", 2, 448.0, "h4_max_margin2")):
        hist = np.zeros((H, 3), np.float32)
        scale = np.ones(3, np.float32)
        traj, inv_traj = [], []
        for a in seq:
            hist[0] = a
            hist, scale, inv = O.scale_update(hist, scale, np.float32(fmax), margin, algo)
            traj.append(scale.copy())
            inv_traj.append(inv.copy())
        out[f"scale_{tag}"] = np.stack(traj)
        out[f"scale_inv_{tag}"] = np.stack(inv_traj)
        out[f"hist_final_{tag}"] = hist
    t = out["scale_h16_max_e4m3"][:, 0]
    # hand trace, slot 0, window 16: max stays 8 while the 8 is within the last 16 rows (steps 2..17), then the window max
    assert list(t[:3]) == [224.0, 224.0, 56.0] and t[17] == 56.0 and t[18] == 448.0 and t[19] == np.float32(448.0 / 3.0)
    assert out["scale_h16_max_e4m3"][2, 1] == 448.0  # inf amax: keep previous scale (448/1)
    assert out["scale_h1024_recent_e5m2"][2, 2] == np.float32(O.FLT_MAX)  # 57344/1e-45 overflows -> FLT_MAX guard
    np.savez_compressed(os.path.join(HERE, "g3_scale_trajectory.npz"), **out)


def gemm_operands(seed, M, N, K, fmt_a, fmt_b):
    """Deterministic fp8 operand bytes (PCG64 stream is version-stable); NaN/inf encodings are replaced by finite bytes."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(M, K), dtype=np.uint8)
    b = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    for arr, fmt in ((a, fmt_a), (b, fmt_b)):
        if fmt == O.E4M3:
            arr[(arr & 0x7F) == 0x7F] = 0x3A
        else:
            arr[(arr & 0x7C) == 0x7C] = 0x3A
        arr[(arr & 0x7F) > 0x5F] &= 0xBF  # keep magnitudes moderate so the bf16 output stays finite
    return a, b


def g4_gemm():
    out, meta = {}, {}
    for tag, (M, N, K, fa, fb) in {"small_e4m3": (64, 96, 128, O.E4M3, O.E4M3), "small_hybrid": (64, 96, 128, O.E5M2, O.E4M3),
                                   "k3072": (256, 512, 3072, O.E4M3, O.E4M3)}.items():
        seed = {"small_e4m3": 11, "small_hybrid": 12, "k3072": 13}[tag]
        a, b = gemm_operands(seed, M, N, K, fa, fb)
        sa, sb = np.float32(0.0123), np.float32(1.0 / 37.0)
        bias = O.f32_to_bf16_bits(np.linspace(-1, 1, N, dtype=np.float32)) if tag != "k3072" else None
        ref = O.gemm_fp8_tn(a, b, fa, fb, sa, sb, bias, out_f32=True)
        # independent check: torch._scaled_mm on CPU (row-major A, column-major B)
        ta = torch.from_numpy(a).view(TORCH_DT[fa])
        tb = torch.from_numpy(b).view(TORCH_DT[fb])
        try:
            got = torch._scaled_mm(ta, tb.t(), scale_a=torch.tensor(float(sa)), scale_b=torch.tensor(float(sb)),
                                   out_dtype=torch.float32).numpy()
            if bias is not None:
                got = got + O.bf16_bits_to_f32(bias)[None, :]
            tol = O.gemm_tolerance(ref)
            assert (np.abs(got - ref) <= tol).all(), f"G4 {tag}: torch._scaled_mm disagrees with the oracle"
            meta[tag] = "oracle == torch._scaled_mm within gemm_tolerance"
        except (RuntimeError, NotImplementedError) as e:  # pragma: no cover
            meta[tag] = f"torch._scaled_mm unavailable for this combination: {type(e).__name__}"
        out[f"d_{tag}"] = ref
        out[f"shape_{tag}"] = np.array([M, N, K, fa, fb, seed], np.int64)
        out[f"scales_{tag}"] = np.array([sa, sb], np.float32)
        if bias is not None:
            out[f"bias_{tag}"] = bias
        meta[tag + "_sha256_a"] = hashlib.sha256(a.tobytes()).hexdigest()
        meta[tag + "_sha256_b"] = hashlib.sha256(b.tobytes()).hexdigest()
    np.savez_compressed(os.path.join(HERE, "g4_gemm.npz"), **out)
    return meta


def g5_linear_steps():
    out = {}
    M, K, N = 64, 128, 96
    g = torch.Generator().manual_seed(5)
    w = bits(torch.randn(N, K, generator=g) * 0.05)
    b = bits(torch.randn(N, generator=g) * 0.1)
    xs = [bits(torch.randn(M, K, generator=g) * (1.0 + i)) for i in range(3)]
    dys = [bits(torch.randn(M, N, generator=g) / 32 * (2.0 ** i)) for i in range(3)]
    out.update(w=w, bias=b, x=np.stack(xs), dy=np.stack(dys))
    for tag, (ff, fb) in {"hybrid": (O.E4M3, O.E5M2), "e4m3": (O.E4M3, O.E4M3)}.items():
        lin = O.DelayedLinearOracle(ff, fb, history_len=16, algo="max")
        for i in range(3):
            y = lin.forward(xs[i], w, b)
            lin.end_forward()
            dx, dw, db = lin.backward(dys[i])
            lin.end_backward()
            out[f"{tag}_y{i}"], out[f"{tag}_dx{i}"], out[f"{tag}_dw{i}"], out[f"{tag}_db{i}"] = y, dx, dw, db
            out[f"{tag}_scale_fwd{i}"], out[f"{tag}_scale_bwd{i}"] = lin.s_fwd.copy(), lin.s_bwd.copy()
        out[f"{tag}_hist_fwd"], out[f"{tag}_hist_bwd"] = lin.h_fwd.copy(), lin.h_bwd.copy()
    np.savez_compressed(os.path.join(HERE, "g5_linear_steps.npz"), **out)


def e8m0_roundup_independent(v: float) -> int:
    """Smallest e with 2^(e-127) >= v (v finite, > 0), via frexp; clamps to [0, 254]."""
    if v == 0.0:
        return 0
    m, ex = math.frexp(v)  # v = m * 2^ex, m in [0.5, 1)
    e = (ex - 1 if m == 0.5 else ex) + 127
    return min(max(e, 0), 254)


def g6_mxfp8():
    out = {}
    for tag, xb in g2_inputs().items():
        yr, er = O.mxfp8_quantize_rowwise(xb, O.E4M3)
        yc, ec = O.mxfp8_quantize_colwise(xb, O.E4M3)
        xf = O.bf16_bits_to_f32(xb)
        blk = np.abs(xf).reshape(xf.shape[0], -1, 32).max(-1)
        want = np.vectorize(e8m0_roundup_independent)((blk * (np.float32(1.0) / np.float32(448.0))).astype(np.float32).astype(np.float64))
        np.testing.assert_array_equal(er, want.astype(np.uint8), err_msg="G6 e8m0 (frexp cross-check)")
        # element check through torch's fp8 cast: |x| * 2^(127-e) <= 448 by construction, so no clamp is involved
        inv = np.ldexp(np.float32(1.0), 127 - np.repeat(er.astype(np.int64), 32, axis=1)).astype(np.float32)
        np.testing.assert_array_equal(yr, torch_sat_cast((xf * inv).astype(np.float32), np.float32(1.0), O.E4M3))
        out[f"row_y_{tag}"], out[f"row_e_{tag}"], out[f"col_y_{tag}"], out[f"col_e_{tag}"] = yr, er, yc, ec
    # block-scaled GEMM on the s1 input against a second seeded operand
    torch.manual_seed(6)
    wb = bits(torch.randn(128, 512) * 0.02)
    w8, we = O.mxfp8_quantize_rowwise(wb, O.E4M3)
    out["gemm_w_bits"] = wb
    out["gemm_d"] = O.gemm_mxfp8_tn(out["row_y_s1"], out["row_e_s1"], w8, we, out_f32=True)
    np.savez_compressed(os.path.join(HERE, "g6_mxfp8.npz"), **out)


def g7_hf_tiny_llama():
    """HF bf16 path (reference's no-TE model, train_fp8.py:118-124; step order :270-291) on CPU: Llama-3.2-1B widths,
    2 layers, vocab cut to 4096 to keep the fixture run short; B=1, S=128, seed 42, 3 AdamW steps at lr 1e-3."""
    from llm_fp8_amd import llama, train
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=1, max_seq_length=128, mixed_precision="bf16", use_te=False,
                               num_hidden_layers=2, vocab_size=4096, num_warmup_steps=0, learning_rate=1e-3)
    config = llama.llama_config(cfg.model_name, num_hidden_layers=2, vocab_size=4096)
    torch.manual_seed(42)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        model = LlamaForCausalLM(config)
    finally:
        torch.set_default_dtype(prev)
    model.train()
    gen = torch.Generator().manual_seed(42)
    batch = train.synthetic_batch(cfg, 4096, torch.device("cpu"), gen)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    losses, gnorms = [], []
    for _ in range(3):
        out = model(**batch)
        out.loss.backward()
        gnorms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)))
        opt.step()
        opt.zero_grad()
        losses.append(float(out.loss.detach()))
    return {"config": {"model": "llama-3.2-1b widths", "layers": 2, "vocab": 4096, "batch": 1, "seq": 128, "seed": 42, "lr": 1e-3,
                       "optimizer": "torch.optim.AdamW, clip 1.0, same batch every step"},
            "input_ids": batch["input_ids"].tolist(), "loss": losses, "grad_norm": gnorms,
            "ln_vocab": math.log(4096)}


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g1_cast_kat()
    g2_random_cast()
    g3_scale_trajectory()
    meta = {"g4": g4_gemm()}
    g5_linear_steps()
    g6_mxfp8()
    meta["g7"] = g7_hf_tiny_llama()
    meta["generator"] = {"torch": torch.__version__, "numpy": np.__version__}
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for fn in sorted(os.listdir(HERE)):
        print(f"{fn:32s} {os.path.getsize(os.path.join(HERE, fn)):9d} B")


if __name__ == "__main__":
    main()
