"""CPU: file-level checkpoint I/O in the HF <-> fused layout (te_llama.py:100-178, train_fp8.py:657-681)."""
import json
import os

import pytest
import torch


def _tiny_cfg(**kw):
    from llm_fp8_amd import llama
    base = dict(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2,
                head_dim=16, vocab_size=128, max_position_embeddings=64)
    base.update(kw)
    return llama.llama_config("llama-3.2-1b", **base)


def _hf_model(cfg, seed=0):
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        m = LlamaForCausalLM(cfg)
    finally:
        torch.set_default_dtype(prev)
    return m


@pytest.mark.parametrize("tied", [True, False])
@pytest.mark.parametrize("sharded", [False, True])
def test_hf_checkpoint_dir_loads_into_te_model_and_saves_back_for_vanilla_hf(tmp_path, tied, sharded):
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    from llm_fp8_amd import checkpoint, llama
    cfg = _tiny_cfg(tie_word_embeddings=tied)
    hf = _hf_model(cfg)
    src = tmp_path / "src"
    # the checkpoint a user would have on disk: vanilla HF save_pretrained (safetensors), optionally sharded
    hf.save_pretrained(str(src), safe_serialization=True, max_shard_size="40KB" if sharded else "5GB")
    assert os.path.isfile(src / ("model.safetensors.index.json" if sharded else "model.safetensors"))
    shards = checkpoint.resolve_shards(str(src))
    assert (len(shards) > 1) == sharded

    te_model = llama.TELlamaForCausalLM.from_pretrained_local(str(src), config=cfg, scenario="hybrid", torch_dtype=torch.bfloat16)
    assert torch.get_default_dtype() == torch.float32  # restored (the reference leaves bf16 behind: SURVEY.md Appendix C.5)
    sd, hsd = te_model.state_dict(), hf.state_dict()
    f = cfg.intermediate_size
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        assert torch.equal(sd[p + "self_attention.layernorm_qkv.query_weight"], hsd[p + "self_attn.q_proj.weight"])
        assert torch.equal(sd[p + "self_attention.layernorm_qkv.key_weight"], hsd[p + "self_attn.k_proj.weight"])
        assert torch.equal(sd[p + "self_attention.proj.weight"], hsd[p + "self_attn.o_proj.weight"])
        assert torch.equal(sd[p + "layernorm_mlp.fc1_weight"][:f], hsd[p + "mlp.gate_proj.weight"])
        assert torch.equal(sd[p + "layernorm_mlp.fc1_weight"][f:], hsd[p + "mlp.up_proj.weight"])
        assert torch.equal(sd[p + "layernorm_mlp.fc2_weight"], hsd[p + "mlp.down_proj.weight"])
        assert torch.equal(sd[p + "layernorm_mlp.layer_norm_weight"], hsd[p + "post_attention_layernorm.weight"])
    assert torch.equal(sd["model.embed_tokens.weight"], hsd["model.embed_tokens.weight"])
    assert torch.equal(sd["lm_head.weight"], hsd["lm_head.weight"])
    assert torch.equal(sd["model.norm.weight"], hsd["model.norm.weight"])

    # ... and back: the saved directory must load into a VANILLA LlamaForCausalLM with identical weights
    out = tmp_path / "out"
    files = checkpoint.save_pretrained(te_model, str(out), max_shard_bytes=(40 << 10) if sharded else (5 << 30))
    assert (len(files) > 1) == sharded and os.path.isfile(out / "config.json")
    if sharded:
        idx = json.load(open(out / "model.safetensors.index.json"))
        assert set(idx["weight_map"].values()) == {os.path.basename(fn) for fn in files}
    back = LlamaForCausalLM.from_pretrained(str(out), torch_dtype=torch.bfloat16)
    bsd = back.state_dict()
    assert set(bsd) == set(hsd)
    for k in hsd:
        assert torch.equal(bsd[k], hsd[k]), k


def test_te_layout_roundtrip_keeps_te_names_and_biases(tmp_path):
    from llm_fp8_amd import checkpoint, llama
    cfg = _tiny_cfg()
    torch.manual_seed(1)
    m = llama.TELlamaForCausalLM(cfg, "default")
    with torch.no_grad():
        m.model.layers[0].layernorm_mlp.fc1_bias.normal_()   # a trained TE-only bias: HF layout must refuse it, TE layout keeps it
    with pytest.raises(ValueError):
        checkpoint.save_pretrained(m, str(tmp_path / "hf"), layout="hf")
    checkpoint.save_pretrained(m, str(tmp_path / "te"), layout="te")
    m2 = llama.TELlamaForCausalLM.from_pretrained_local(str(tmp_path / "te"), config=cfg, scenario="default", torch_dtype=torch.float32)
    for (k, a), (k2, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k == k2
        if isinstance(a, torch.Tensor) and not k.endswith("_extra_state"):
            assert torch.equal(a.float(), b.float()), k


def test_missing_or_pickled_checkpoints_are_refused(tmp_path):
    from llm_fp8_amd import checkpoint
    (tmp_path / "pytorch_model.bin").write_bytes(b"not read")
    with pytest.raises(AssertionError, match="Only sharded PyTorch ckpt format"):  # te_llama.py:150
        checkpoint.resolve_shards(str(tmp_path))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps({"weight_map": {"a": "model-00001-of-00002.safetensors"}}))
    with pytest.raises(FileNotFoundError):
        checkpoint.resolve_shards(str(tmp_path))


def test_train_harness_saves_and_reloads(tmp_path):
    """--output_dir / --load_dir of the harness (train_fp8.py:657-681): CPU plumbing config, bf16, no TE."""
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    from llm_fp8_amd import train
    out = tmp_path / "ckpt"
    args = ["--model_name", "llama-3.2-1b", "--batch_size", "1", "--max_seq_length", "16", "--mixed_precision", "bf16",
            "--num_hidden_layers", "1", "--vocab_size", "256", "--num_steps", "2", "--sharding_mode", "none"]
    train.main(args + ["--output_dir", str(out)])
    assert os.path.isfile(out / "model.safetensors") and os.path.isfile(out / "config.json")
    m = LlamaForCausalLM.from_pretrained(str(out), torch_dtype=torch.bfloat16)
    assert m.config.num_hidden_layers == 1 and m.config.vocab_size == 256
    train.main(args + ["--load_dir", str(out), "--use_te"])  # TE-layer model started from the saved HF-layout checkpoint


def test_gradient_accumulation_matches_the_large_batch():
    """train_step with N micro-batches = one step on the N x batch (loss / N, one optimiser step; train_multi_gpu.py:661,714-737)."""
    from llm_fp8_amd import train
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=2, max_seq_length=16, mixed_precision="bf16", use_te=False,
                               num_hidden_layers=1, vocab_size=128, sharding_mode="none", num_warmup_steps=0, learning_rate=1e-3)
    dev = torch.device("cpu")

    def fresh():
        torch.manual_seed(7)
        m = train.create_model(cfg, dev).float()
        opt, sched = train.create_optimizer(m, cfg)
        m.train()
        grads = {}
        step = opt.step

        def spy(*a, **k):  # gradients as the optimiser sees them (AdamW's first update is ~lr * sign(g): compare g, not w)
            grads.update({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            return step(*a, **k)

        opt.step = spy
        return m, opt, sched, grads

    g = torch.Generator().manual_seed(3)
    mbs = [train.synthetic_batch(cfg, 128, dev, generator=g) for _ in range(4)]
    big = {k: torch.cat([mb[k] for mb in mbs]) for k in mbs[0]}
    m1, o1, s1, g1 = fresh()
    l1 = train.train_step(m1, mbs, o1, s1, cfg)
    m2, o2, s2, g2 = fresh()
    l2 = train.train_step(m2, big, o2, s2, cfg)
    assert abs(float(l1.detach()) - float(l2.detach())) < 1e-5
    assert g1.keys() == g2.keys() and len(g1) > 0
    for k in g1:
        assert torch.allclose(g1[k], g2[k], rtol=1e-4, atol=1e-7), k
    assert s1.last_epoch == s2.last_epoch == 1  # ONE scheduler step per accumulation window
