"""File-level checkpoint I/O in the HF <-> fused layout.

Loading mirrors `TELlamaForCausalLM.from_pretrained_local` (te_llama.py:100-178): resolve
`model.safetensors.index.json` (sharded) or `model.safetensors` in a LOCAL directory, then for every shard
`replace_params` (HF names -> fused TE names) followed by `load_state_dict(strict=False)` for the rest.
Saving mirrors `ModelSaver.save_model` (train_fp8.py:657-681), but writes HF parameter names through
`llama.to_hf_state_dict`, so the result loads into a vanilla `LlamaForCausalLM` (the reference's
`save_pretrained` writes TE names that HF cannot read back); `layout="te"` keeps the TE names and the FP8
`_extra_state` blobs instead.

Only safetensors files are read (weights only, nothing is unpickled); a `pytorch_model.bin[.index.json]`
checkpoint is refused with the reference's own message.
"""
from __future__ import annotations

import gc
import json
import os
from typing import Dict, Iterable, List, Optional

import torch

SAFE_INDEX = "model.safetensors.index.json"
SAFE_SINGLE = "model.safetensors"
EXTRA_SINGLE = "fp8_extra_state.safetensors"


def resolve_shards(path: str) -> List[str]:
    """Shard files of a local checkpoint directory, in the order te_llama.py:113-163 probes for them."""
    index = os.path.join(path, SAFE_INDEX)
    if os.path.isfile(index):
        with open(index) as f:
            weight_map = json.load(f)["weight_map"]
        files = sorted(set(weight_map.values()))
        missing = [fn for fn in files if not os.path.isfile(os.path.join(path, fn))]
        if missing:
            raise FileNotFoundError(f"{index} names shards that are not in {path}: {missing}")
        return [os.path.join(path, fn) for fn in files]
    single = os.path.join(path, SAFE_SINGLE)
    if os.path.isfile(single):
        return [single]
    raise AssertionError("Only sharded PyTorch ckpt format supported at the moment")  # te_llama.py:150


def load_shard(file: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return load_file(file, device="cpu")


def load_into(model: torch.nn.Module, path: str, config, strict: bool = False) -> List[str]:
    """Copy a local HF-layout (or TE-layout) safetensors checkpoint into `model` shard by shard.
    Returns the shard files read.  Like the reference (`strict=False` per shard, te_llama.py:171) a shard may hold any subset of
    the keys -- but what NO shard covered is reported: a warning lists the parameters that kept their initial values (a truncated
    or mismatched shard set would otherwise load "successfully") and the checkpoint keys nothing consumed; `strict=True` raises."""
    import warnings
    from .llama import replace_params
    shards = resolve_shards(path)
    own = model.state_dict()
    covered, fused, unexpected = set(), set(), set()
    hf_fused = ("input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.down_proj.weight", "mlp.gate_proj.weight",
                "mlp.up_proj.weight")
    for shard in shards:
        state = load_shard(shard)
        replace_params(state, own, config, written=fused)    # parameters that live under fused TE names
        res = model.load_state_dict(state, strict=False)     # everything else (embeddings, final norm, lm_head, TE-named keys)
        covered.update(k for k in state if k in own)
        unexpected.update(k for k in res.unexpected_keys if not k.endswith(hf_fused))
        del state
        gc.collect()                                         # te_llama.py:174-176
    for k in list(fused):
        if "#" in k:  # fc1 = gate | up: covered when both halves arrived (or the TE-named tensor itself)
            base = k.split("#")[0]
            if base + "#gate" in fused and base + "#up" in fused:
                covered.add(base)
        else:
            covered.add(k)
    tied = bool(getattr(config, "tie_word_embeddings", False))
    missing = sorted(k for k in own if k not in covered and not k.endswith("_extra_state")
                     and not (tied and k == "lm_head.weight" and "model.embed_tokens.weight" in covered)
                     and not k.endswith(("fc1_bias", "fc2_bias")))  # TE-only MLP biases: HF Llama has none (zero-initialised)
    if missing or unexpected:
        msg = (f"checkpoint {path}: {len(missing)} parameter(s) were in no shard and keep their initial values "
               f"({missing[:8]}{' ...' if len(missing) > 8 else ''}); {len(unexpected)} checkpoint key(s) matched nothing "
               f"({sorted(unexpected)[:8]}{' ...' if len(unexpected) > 8 else ''})")
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg)
    extra = os.path.join(path, EXTRA_SINGLE)
    if os.path.isfile(extra):
        model.load_state_dict(load_shard(extra), strict=False)
    return shards


def _tied(config) -> bool:
    return bool(getattr(config, "tie_word_embeddings", False))


def _dedupe_tied(state: Dict[str, torch.Tensor], config) -> Dict[str, torch.Tensor]:
    """safetensors refuses aliased tensors; HF's convention for tied models is to omit `lm_head.weight`."""
    if _tied(config) and "lm_head.weight" in state and "model.embed_tokens.weight" in state:
        if state["lm_head.weight"].data_ptr() == state["model.embed_tokens.weight"].data_ptr():
            state = {k: v for k, v in state.items() if k != "lm_head.weight"}
    return state


def _shard(state: Dict[str, torch.Tensor], max_shard_bytes: int) -> List[Dict[str, torch.Tensor]]:
    shards, cur, size = [], {}, 0
    for k, v in state.items():
        nb = v.numel() * v.element_size()
        if cur and size + nb > max_shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = v
        size += nb
    if cur:
        shards.append(cur)
    return shards


def save_pretrained(model: torch.nn.Module, out_dir: str, config=None, layout: str = "hf", max_shard_bytes: int = 5 << 30,
                    save_fp8_state: bool = False) -> List[str]:
    """Write `config.json` + `model.safetensors` (or `model-0000i-of-0000n.safetensors` + index) into out_dir.

    layout "hf": HF Llama parameter names (loads into vanilla `LlamaForCausalLM`); the FP8 metadata blobs are dropped, or
    written next to the weights as `fp8_extra_state.safetensors` when `save_fp8_state`.
    layout "te": the module tree's own names, `_extra_state` included (what the reference's save_pretrained stores)."""
    from safetensors.torch import save_file
    from .llama import to_hf_state_dict
    config = config if config is not None else model.config
    os.makedirs(out_dir, exist_ok=True)
    raw = model.state_dict()
    if layout == "hf":
        state = to_hf_state_dict(raw, config, keep_extra=False)
    elif layout == "te":
        state = dict(raw)
    else:
        raise ValueError(f"layout must be 'hf' or 'te', got {layout!r}")
    state = _dedupe_tied(state, config)
    bad = [k for k, v in state.items() if not isinstance(v, torch.Tensor)]
    if bad:  # safetensors holds tensors only; dropping an entry silently would save a checkpoint that restores differently
        raise TypeError(f"save_pretrained: non-tensor state-dict entries cannot be written to safetensors: {bad[:8]}")
    state = {k: v.detach().to("cpu").contiguous() for k, v in state.items()}
    shards = _shard(state, max_shard_bytes)
    written = []
    if len(shards) == 1:
        fn = os.path.join(out_dir, SAFE_SINGLE)
        save_file(shards[0], fn, metadata={"format": "pt"})
        written.append(fn)
    else:
        weight_map, total = {}, 0
        for i, sh in enumerate(shards):
            name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(out_dir, name), metadata={"format": "pt"})
            written.append(os.path.join(out_dir, name))
            for k, v in sh.items():
                weight_map[k] = name
                total += v.numel() * v.element_size()
        with open(os.path.join(out_dir, SAFE_INDEX), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
    if layout == "hf" and save_fp8_state:
        extra = {k: v.detach().to("cpu").contiguous() for k, v in raw.items() if k.endswith("_extra_state") and isinstance(v, torch.Tensor)}
        if extra:
            save_file(extra, os.path.join(out_dir, EXTRA_SINGLE), metadata={"format": "pt"})
    if hasattr(config, "save_pretrained"):
        config.save_pretrained(out_dir)
    return written


def unwrap(model: torch.nn.Module) -> torch.nn.Module:
    """accelerator.unwrap_model (train_fp8.py:667): the module under a DDP / arena wrapper."""
    return getattr(model, "module", model)
