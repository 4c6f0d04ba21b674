"""CPU: host-side logic that needs no GPU -- recipes, state-dict mapping both ways, tile picking rules mirrored in Python."""
import pytest
import torch


def test_recipe_surface_matches_te_defaults():
    from llm_fp8_amd.common.recipe import DelayedScaling, Format, MXFP8BlockScaling, fmt_codes
    r = DelayedScaling()
    assert r.fp8_format is Format.HYBRID and r.amax_history_len == 1024 and r.amax_compute_algo == "max" and r.margin == 0
    assert Format.HYBRID.value.max_fwd == 448.0 and Format.HYBRID.value.max_bwd == 57344.0 and Format.E4M3.value.max_bwd == 448.0
    assert fmt_codes(Format.HYBRID) == (0, 1) and fmt_codes(Format.E4M3) == (0, 0) and fmt_codes(Format.E5M2) == (1, 1)
    assert MXFP8BlockScaling().mxfp8() and not r.mxfp8() and r.delayed()
    with pytest.raises(ValueError):
        DelayedScaling(amax_compute_algo="median")
    # accelerate's TERecipeKwargs call shape (utils/transformer_engine.py:156-177)
    DelayedScaling(margin=0, interval=1, fp8_format=Format.E4M3, amax_history_len=16, amax_compute_algo="max",
                   override_linear_precision=(False, False, False))


def test_scenario_recipes_follow_the_three_reference_files():
    from llm_fp8_amd import llama
    from llm_fp8_amd.common.recipe import Format
    a, m = llama.scenario_recipes("default")      # te_llama.py:39-40
    assert a.fp8_format is Format.HYBRID and m.fp8_format is Format.E4M3 and a.amax_history_len == m.amax_history_len == 16
    a, m = llama.scenario_recipes("hybrid")       # te_llama_hybrid.py:39
    assert a is m and a.fp8_format is Format.HYBRID
    a, m = llama.scenario_recipes("mxfp8")        # te_llama_mxfp8.py:28-29
    assert a is m and a.mxfp8() and a.fp8_format is Format.E4M3
    o = llama.outer_recipe_for_scenario("default")  # train_fp8.py:144-146 -> TERecipeKwargs() defaults
    assert o.fp8_format is Format.HYBRID and o.amax_history_len == 1024 and o.amax_compute_algo == "most_recent"
    o = llama.outer_recipe_for_scenario("mxfp8")    # train_fp8.py:158-165
    assert o.fp8_format is Format.E4M3 and o.amax_history_len == 16 and o.delayed()


def test_replace_params_and_its_inverse_roundtrip():
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    from llm_fp8_amd import llama
    cfg = llama.llama_config("llama-3.2-1b", num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4,
                             num_key_value_heads=2, head_dim=16, vocab_size=128, max_position_embeddings=64)
    torch.manual_seed(0)
    hf = LlamaForCausalLM(cfg)
    te_model = llama.TELlamaForCausalLM.from_hf_state_dict(hf.state_dict(), cfg, "default", torch_dtype=torch.float32)
    sd = te_model.state_dict()
    assert "model.layers.1.layernorm_mlp.fc1_weight" in sd and "model.layers.0.self_attention.layernorm_qkv.key_weight" in sd
    assert "model.layers.0.layernorm_mlp.fc1_bias" in sd  # TE-only parameters
    back = llama.to_hf_state_dict(sd, cfg)
    ref = hf.state_dict()
    assert set(back.keys()) == set(ref.keys())
    for k in ref:
        assert torch.equal(back[k].float(), ref[k].float()), k
    hf2 = LlamaForCausalLM(cfg)
    hf2.load_state_dict(back)  # vanilla HF can load what we save
    sd["model.layers.0.layernorm_mlp.fc2_bias"] = torch.ones_like(sd["model.layers.0.layernorm_mlp.fc2_bias"])
    with pytest.raises(ValueError, match="no MLP bias"):
        llama.to_hf_state_dict(sd, cfg)


def test_llama_config_table():
    from llm_fp8_amd import llama
    c = llama.llama_config("meta-llama/Llama-3.2-3B")
    assert (c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, c.head_dim) == (3072, 8192, 28, 24, 8, 128)
    assert c.vocab_size == 128256 and c.tie_word_embeddings
    c = llama.llama_config("llama-3.1-8b")
    assert (c.hidden_size, c.intermediate_size, c.num_hidden_layers) == (4096, 14336, 32) and not c.tie_word_embeddings
    with pytest.raises(KeyError):
        llama.llama_config("gpt-2")


def test_committed_bench_line_has_the_contract_fields():
    """The newest committed headline bench line (profiles/r01*_bench_3b_default.json, printed by bench.py on the GPU box)
    carries every field of the bench contract, incl. the `roofline` and `cpu_baseline` objects."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_3b_default.json")))
    assert files, "no committed bench line"
    d = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    tokens = 16 * 512 * d["n_gpus"]
    assert abs(d["value"] - tokens / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_padding_mask_never_reaches_hfs_mask_builder():
    """te_llama.py:68-82 forwards `attention_mask`, TE's causal core ignores it.  HF's mask builder would synchronise the host on
    it once per forward; the model hands it a 4-D placeholder instead (llama._drop_padding_mask).  Outputs must not change."""
    import torch
    from transformers import LlamaConfig
    from llm_fp8_amd import llama
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=256, max_position_embeddings=128)
    torch.manual_seed(0)
    model = llama.TELlamaForCausalLM(cfg)
    ids = torch.randint(0, 256, (2, 32))
    seen = []
    h = model.model.layers[0].register_forward_pre_hook(lambda m, a, kw: seen.append(kw.get("attention_mask")), with_kwargs=True)
    with torch.no_grad():
        with_mask = model(input_ids=ids, attention_mask=torch.ones_like(ids)).logits
        no_mask = model(input_ids=ids).logits
        four_d = torch.ones((2, 1, 32, 32), dtype=torch.bool).tril()
        model(input_ids=ids, attention_mask=four_d)
        h.remove()
        model.model._forward_pre_hooks.clear()  # HF's own path (2-D mask -> its builder)
        plain = model(input_ids=ids, attention_mask=torch.ones_like(ids)).logits
    assert torch.equal(with_mask, no_mask) and torch.equal(with_mask, plain)
    assert seen[0] is not None and seen[0].dim() == 4 and seen[0].numel() == 1  # the placeholder reached the decoder layers
    assert seen[2] is four_d or (seen[2].shape == four_d.shape)  # a caller's prepared 4-D mask is passed through untouched
