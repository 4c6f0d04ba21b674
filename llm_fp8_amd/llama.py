"""Counterpart of the reference's te_llama.py / te_llama_hybrid.py / te_llama_mxfp8.py on top of
`llm_fp8_amd.pytorch` (same class names, same monkey-patch, same parameter names), plus the two
accelerate steps train_fp8.py relies on (`convert_model`, `apply_fp8_autowrap`).

The three reference files differ only in the recipe objects, so one
file serves all three scenarios:

  scenario   attention region                         MLP region                               reference
  default    DelayedScaling(HYBRID, 16, "max")        DelayedScaling(E4M3, 16, "max")          te_llama.py:39-40
  hybrid     DelayedScaling(HYBRID, 16, "max")        same recipe object                       te_llama_hybrid.py:39
  mxfp8      MXFP8BlockScaling(E4M3)                  same recipe object                       te_llama_mxfp8.py:28-29

No weights, tokenizer or dataset exist offline: configs are hard-coded from the public model cards
 and models are random-initialised.
"""
from __future__ import annotations

import os
import re
from contextlib import contextmanager
from typing import Dict, Optional

import torch

from . import pytorch as te
from .common.recipe import DelayedScaling, Format, MXFP8BlockScaling
from .pytorch.fp8 import FP8GlobalStateManager
from .pytorch.module import _can_fuse_norm, residual_add_stats

LLAMA_CONFIGS: Dict[str, dict] = {
    "llama-3.2-1b": dict(hidden_size=2048, intermediate_size=8192, num_hidden_layers=16, num_attention_heads=32,
                         num_key_value_heads=8, head_dim=64, tie_word_embeddings=True),
    "llama-3.2-3b": dict(hidden_size=3072, intermediate_size=8192, num_hidden_layers=28, num_attention_heads=24,
                         num_key_value_heads=8, head_dim=128, tie_word_embeddings=True),
    "llama-3.1-8b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                         num_key_value_heads=8, head_dim=128, tie_word_embeddings=False),
}
_ALIASES = {"meta-llama/llama-3.2-1b": "llama-3.2-1b", "meta-llama/llama-3.2-3b": "llama-3.2-3b",
            "meta-llama/llama-3.1-8b": "llama-3.1-8b", "meta-llama/meta-llama-3.1-8b": "llama-3.1-8b"}


def llama_config(name: str, **overrides):
    """Hard-coded LlamaConfig (there is no config.json offline; train_fp8.py:88 reads one from the hub cache)."""
    from transformers import LlamaConfig
    key = name.lower()
    key = _ALIASES.get(key, key)
    if key not in LLAMA_CONFIGS:
        raise KeyError(f"unknown model {name!r}; known: {sorted(LLAMA_CONFIGS)}")
    kw = dict(vocab_size=128256, rms_norm_eps=1e-5, max_position_embeddings=131072, rope_theta=500000.0,
              hidden_act="silu", attention_bias=False, mlp_bias=False, initializer_range=0.02, use_cache=False,
              bos_token_id=128000, eos_token_id=128001)
    kw.update(LLAMA_CONFIGS[key])
    kw.update(overrides)
    cfg = LlamaConfig(**kw)
    cfg._attn_implementation = "sdpa"  # the reference asks for flash_attention_2 (train_fp8.py:89): absent on ROCm here
    return cfg


def scenario_recipes(scenario: str):
    """(attention-region recipe, MLP-region recipe) for --fp8_scenario (train_fp8.py:103-116)."""
    if scenario == "default":
        return (DelayedScaling(fp8_format=Format.HYBRID, amax_history_len=16, amax_compute_algo="max"),
                DelayedScaling(fp8_format=Format.E4M3, amax_history_len=16, amax_compute_algo="max"))
    if scenario == "hybrid":
        r = DelayedScaling(fp8_format=Format.HYBRID, amax_history_len=16, amax_compute_algo="max")
        return r, r
    if scenario == "mxfp8":
        r = MXFP8BlockScaling(fp8_format=Format.E4M3)
        return r, r
    raise ValueError(f"unknown fp8 scenario {scenario!r}")


_ROPE_CACHE: Dict[tuple, torch.Tensor] = {}


def _rope_table(dim: int, max_len: int, device) -> torch.Tensor:
    """One shared RoPE angle table per (dim, length, device).  The reference builds one 131072-row table PER
    LAYER (te_llama.py:65-66); sharing it is value-identical and saves (L-1) x 64 MB of HBM on the 3B model."""
    key = (dim, max_len, str(device))
    if key not in _ROPE_CACHE:
        _ROPE_CACHE[key] = te.attention.RotaryPositionEmbedding(dim)(max_seq_len=max_len).to(device)
    return _ROPE_CACHE[key]


def _handed_rstd(h: torch.Tensor):
    """(rstd, eps) left on the tensor by the previous decoder layer, unless the tensor was modified in place since."""
    tag = getattr(h, "_mi_rstd", None)
    if tag is None or tag[2] != h._version:
        return None
    return tag[0], tag[1]


class TELlamaDecoderLayer(torch.nn.Module):
    """te_llama.py:41-82: MultiheadAttention (+input RMSNorm, GQA, bshd) and LayerNormMLP (RMSNorm, swiglu,
    TE-default bias=True), each under its own fp8_autocast region, residual adds in between."""

    scenario = "default"

    def __init__(self, config, *args, dropout_rate: float = 0.0, **kwargs):
        super().__init__()
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.attn_recipe, self.mlp_recipe = scenario_recipes(self.scenario)
        self.self_attention = te.MultiheadAttention(
            hidden_size=config.hidden_size,
            num_attention_heads=config.num_attention_heads,
            kv_channels=getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads,
            bias=False,
            layernorm_epsilon=config.rms_norm_eps,
            attention_dropout=dropout_rate,
            fuse_qkv_params=False,
            normalization="RMSNorm",
            num_gqa_groups=config.num_key_value_heads,
            qkv_format="bshd",
            input_layernorm=True,
            device=dev,
        )
        self.layernorm_mlp = te.LayerNormMLP(
            hidden_size=config.hidden_size,
            ffn_hidden_size=config.intermediate_size,
            normalization="RMSNorm",
            activation="swiglu",
            device=dev,
        )
        head_dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
        self.te_rope_emb = _rope_table(head_dim, config.max_position_embeddings, dev)

    # "fused": the build's own wiring (private `_with_skip` / `_rstd` kwargs: residual gradient folded into the RMSNorm
    # backward, residual add fused with the next norm's statistics).  "reference": the call pattern of te_llama.py:76-81
    # verbatim -- public kwargs only, plain `h + module(h)` residuals -- i.e. what an UNCHANGED te_llama.py gets when it
    # imports this package in place of transformer_engine (bench.py --route reference measures it).
    route = "fused"

    def forward(self, hidden_states, attention_mask=None, **kwargs):
        if not isinstance(hidden_states, torch.Tensor):
            raise TypeError("hidden_states must be a torch.Tensor")
        if attention_mask is not None and not isinstance(attention_mask, torch.Tensor):
            raise TypeError("attention_mask must be a torch.Tensor")
        fp8 = hidden_states.is_cuda  # FP8 is unconditionally on in the reference layer (te_llama.py:76,79)
        if self.route == "reference":
            with te.fp8_autocast(enabled=fp8, fp8_recipe=self.attn_recipe):
                hidden_states = hidden_states + self.self_attention(hidden_states, attention_mask=attention_mask,
                                                                    rotary_pos_emb=self.te_rope_emb)
            with te.fp8_autocast(enabled=fp8, fp8_recipe=self.mlp_recipe):
                hidden_states = hidden_states + self.layernorm_mlp(hidden_states)
            return hidden_states
        # `_with_skip`: the module hands the residual branch back so that its gradient is added inside the fused
        # RMSNorm-backward kernel (same values as `h + f(h)`; one elementwise pass less per residual in backward).
        # residual_add_stats: the add also yields the statistics of the RMSNorm that consumes the sum (the next module's, or
        # -- handed over on the tensor -- the next layer's input norm), saving that norm's own pass over the activations.
        with te.fp8_autocast(enabled=fp8, fp8_recipe=self.attn_recipe):
            attn_out, skip = self.self_attention(hidden_states, attention_mask=attention_mask, rotary_pos_emb=self.te_rope_emb,
                                                 _with_skip=True, _rstd=_handed_rstd(hidden_states))
        eps_mlp = self.layernorm_mlp.eps
        hidden_states, rstd = residual_add_stats(skip, attn_out, eps_mlp)
        # `_defer_bias`: the fc2 bias (TE's default bias=True on LayerNormMLP, te_llama.py:58-63) joins the residual add below
        # instead of the GEMM epilogue (TE's bias + activation fusion idea applied to both MLP biases)
        with te.fp8_autocast(enabled=fp8, fp8_recipe=self.mlp_recipe):
            ffn_out, skip, ffn_bias = self.layernorm_mlp(hidden_states, _with_skip=True, _rstd=(rstd, eps_mlp), _defer_bias=True)
        eps_in = self.self_attention.layernorm_qkv.eps  # every decoder layer of a model shares rms_norm_eps
        hidden_states, rstd = residual_add_stats(skip, ffn_out, eps_in, bias=ffn_bias)
        if rstd is not None:
            hidden_states._mi_rstd = (rstd, eps_in, hidden_states._version)
        return hidden_states


def decoder_layer_cls(scenario: str):
    scenario_recipes(scenario)  # validate
    return type(f"TELlamaDecoderLayer_{scenario}", (TELlamaDecoderLayer,), {"scenario": scenario})


@contextmanager
def replace_decoder(te_decoder_cls):
    """te_llama.py:28-38: swap HF's LlamaDecoderLayer while LlamaForCausalLM is being built."""
    import transformers
    mod = transformers.models.llama.modeling_llama
    original = mod.LlamaDecoderLayer
    mod.LlamaDecoderLayer = te_decoder_cls
    try:
        yield
    finally:
        mod.LlamaDecoderLayer = original


def _drop_padding_mask(module, args, kwargs):
    """Forward pre-hook of the HF `LlamaModel` that carries TE decoder layers.  Their attention core is causal and, as TE does
    with `attn_mask_type="causal"` (te_llama.py:45-56), ignores a padding mask, so the 4-D mask HF would build from
    `attention_mask` is read by nobody -- but building it costs a host synchronisation per forward (`padding_mask.all()` with a
    2-D mask, the packed-sequence check `(...).all()` on `position_ids` with no mask; transformers' masking_utils): the CPU
    waits for the previous step to drain and the GPU then idles ~0.5 ms while the first kernels of the step are enqueued.  A
    mask that is already 4-D is passed through untouched by HF ("already prepared"), so a 2-D / missing mask is replaced by a
    one-element 4-D placeholder.  The outputs do not depend on the mask either way."""
    am = kwargs.get("attention_mask", None)
    if am is None or (isinstance(am, torch.Tensor) and am.dim() != 4):
        ref = kwargs.get("input_ids", None)
        if ref is None:
            ref = kwargs.get("inputs_embeds", None)
        if ref is None and args:
            ref = args[0]
        if isinstance(ref, torch.Tensor):
            kwargs = dict(kwargs)
            kwargs["attention_mask"] = torch.ones((1, 1, 1, 1), dtype=torch.bool, device=ref.device)
    return args, kwargs


def _install_final_norm_fusion(model) -> None:
    """The HF model's final RMSNorm (`model.model.norm`, a chain of ~8 torch elementwise kernels forward and ~10 backward on the
    [tokens, hidden] activations) in front of the lm_head: once the head is one of our FP8 Linears (accelerate's convert_model,
    llama.convert_model), the norm is fused into the head's input cast exactly as LayerNormLinear fuses a decoder norm (K9: statistics
    from the last decoder layer's residual add, normalisation inside mi_norm_cast, mi_rmsnorm_bwd in backward).  The module, its
    parameter and its state-dict key stay HF's; only while `LlamaForCausalLM.forward` runs -- where the norm's output goes to
    `lm_head` and nowhere else -- does the norm hand its weight to the head instead of running.  Anything else (the bare
    `model.model(...)`, `output_hidden_states`, FP8 off, eval under no autocast, a head that is still nn.Linear) takes HF's forward.
    LLM_FP8_AMD_NO_FINAL_NORM_FUSION=1 turns it off."""
    norm = model.model.norm
    hf_forward = norm.forward

    def norm_forward(x):
        head = getattr(norm, "_defer_head", None)
        if (head is not None and isinstance(head, te.Linear) and x.is_cuda and x.dtype == torch.bfloat16 and norm.weight.dtype == torch.bfloat16
                and FP8GlobalStateManager.is_fp8_enabled()
                and _can_fuse_norm(_FinalNormView, FP8GlobalStateManager.get_fp8_recipe(), x)):
            head._pending_norm = (norm.weight, float(norm.variance_epsilon), _handed_rstd(x), x.data_ptr(), tuple(x.shape))
            return x
        return hf_forward(x)

    norm.forward = norm_forward
    lm_forward = model.forward

    def forward(*args, **kwargs):
        want_hidden = kwargs.get("output_hidden_states", None)
        if want_hidden is None:
            want_hidden = getattr(model.config, "output_hidden_states", False)
        keep = kwargs.get("logits_to_keep", 0)
        if os.environ.get("LLM_FP8_AMD_NO_FINAL_NORM_FUSION") == "1" or want_hidden or not (isinstance(keep, int) and keep == 0):
            return lm_forward(*args, **kwargs)
        norm._defer_head = model.lm_head
        try:
            return lm_forward(*args, **kwargs)
        finally:
            norm._defer_head = None
            if isinstance(model.lm_head, te.Linear):
                model.lm_head._pending_norm = None

    model.forward = forward


class _FinalNormView:
    """What module._can_fuse_norm asks of a module, for HF's LlamaRMSNorm."""
    fused_norm, normalization, zero_centered_gamma = True, "RMSNorm", False


class TELlamaForCausalLM:
    """te_llama.py:85-98: HF owns embeddings / final norm / lm_head / loss, our layers own the decoder."""

    def __new__(cls, config, scenario: str = "default"):
        from transformers.models.llama.modeling_llama import LlamaForCausalLM
        with replace_decoder(te_decoder_cls=decoder_layer_cls(scenario)):
            model = LlamaForCausalLM(config)
        model.model.register_forward_pre_hook(_drop_padding_mask, with_kwargs=True)
        _install_final_norm_fusion(model)
        return model

    @classmethod
    def from_pretrained_local(cls, pretrained_model_name_or_path, *args, config, scenario: str = "default",
                              torch_dtype=torch.bfloat16, **kwargs):
        """te_llama.py:100-178: build the TE-layer model, then copy a LOCAL safetensors checkpoint (single file or
        `model.safetensors.index.json` shards) into it shard by shard: `replace_params` for the fused TE names,
        `load_state_dict(strict=False)` for the rest.  Weights-only loading (safetensors); nothing is downloaded."""
        from . import checkpoint
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch_dtype)
        try:
            model = cls(config, scenario)
        finally:
            torch.set_default_dtype(prev)  # the reference never restores it
        checkpoint.load_into(model, str(pretrained_model_name_or_path), config)
        return model

    @classmethod
    def from_hf_state_dict(cls, hf_state_dict, config, scenario: str = "default", torch_dtype=torch.bfloat16):
        """Offline counterpart of from_pretrained_local (te_llama.py:100-178): same copy rules, the state dict is
        handed in instead of being read from a hub snapshot."""
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch_dtype)
        try:
            model = cls(config, scenario)
        finally:
            torch.set_default_dtype(prev)  # the reference never restores it
        sd = dict(hf_state_dict)
        replace_params(sd, model.state_dict(), config)
        model.load_state_dict(sd, strict=False)
        return model


def replace_params(hf_state_dict, te_state_dict, config, written: Optional[set] = None):
    """HF -> TE-named parameters, same mapping as te_llama.py:181-239 (q|k|v separate, gate|up stacked).  The copies go through
    `Tensor.copy_` on the state-dict tensors (which share the parameters' version counters) and not through `.data[:] = ...` as in
    the reference: a `.data` write leaves `Parameter._version` where it was, and the FP8 weight copies the optimiser keeps
    (module.WeightSink / MXWeightSink) are declared current by exactly that counter -- a checkpoint loaded into a model that has
    already stepped would otherwise keep running on the FP8 images of the old weights.  `written` collects the TE keys filled."""
    prefixes = set()
    for k in hf_state_dict.keys():
        m = re.match(r"model\.layers\.\d+\.", k)
        if m is not None:
            prefixes.add(m.group())
    simple = {
        "input_layernorm.weight": "self_attention.layernorm_qkv.layer_norm_weight",
        "self_attn.q_proj.weight": "self_attention.layernorm_qkv.query_weight",
        "self_attn.k_proj.weight": "self_attention.layernorm_qkv.key_weight",
        "self_attn.v_proj.weight": "self_attention.layernorm_qkv.value_weight",
        "self_attn.o_proj.weight": "self_attention.proj.weight",
        "post_attention_layernorm.weight": "layernorm_mlp.layer_norm_weight",
        "mlp.down_proj.weight": "layernorm_mlp.fc2_weight",
    }
    f = config.intermediate_size
    with torch.no_grad():
        for p in prefixes:
            for src, dst in simple.items():
                if p + src in hf_state_dict:
                    te_state_dict[p + dst].copy_(hf_state_dict[p + src])
                    if written is not None:
                        written.add(p + dst)
            if p + "mlp.gate_proj.weight" in hf_state_dict:
                te_state_dict[p + "layernorm_mlp.fc1_weight"][:f].copy_(hf_state_dict[p + "mlp.gate_proj.weight"])
                if written is not None:
                    written.add(p + "layernorm_mlp.fc1_weight#gate")
            if p + "mlp.up_proj.weight" in hf_state_dict:
                te_state_dict[p + "layernorm_mlp.fc1_weight"][f:].copy_(hf_state_dict[p + "mlp.up_proj.weight"])
                if written is not None:
                    written.add(p + "layernorm_mlp.fc1_weight#up")
    return prefixes


def to_hf_state_dict(te_state_dict, config, keep_extra: bool = False):
    """Inverse of `replace_params`: TE-named decoder parameters -> HF Llama names, so a model
    trained here saves a checkpoint vanilla `LlamaForCausalLM` can load (the reference's `save_pretrained` writes the
    TE names, train_fp8.py:657-681, which HF cannot read back).  The zero-initialised TE-only MLP biases are dropped if
    they are still all-zero and refused otherwise (HF Llama has `mlp_bias=False`); FP8 `_extra_state` blobs are
    dropped unless `keep_extra`."""
    out = {}
    f = config.intermediate_size
    simple = {
        "self_attention.layernorm_qkv.layer_norm_weight": "input_layernorm.weight",
        "self_attention.layernorm_qkv.query_weight": "self_attn.q_proj.weight",
        "self_attention.layernorm_qkv.key_weight": "self_attn.k_proj.weight",
        "self_attention.layernorm_qkv.value_weight": "self_attn.v_proj.weight",
        "self_attention.proj.weight": "self_attn.o_proj.weight",
        "layernorm_mlp.layer_norm_weight": "post_attention_layernorm.weight",
        "layernorm_mlp.fc2_weight": "mlp.down_proj.weight",
    }
    for k, v in te_state_dict.items():
        m = re.match(r"(model\.layers\.\d+\.)(.*)", k)
        if k.endswith("_extra_state"):
            if keep_extra:
                out[k] = v
            continue
        if m is None:
            out[k] = v
            continue
        prefix, name = m.group(1), m.group(2)
        if name in simple:
            out[prefix + simple[name]] = v
        elif name == "layernorm_mlp.fc1_weight":
            out[prefix + "mlp.gate_proj.weight"] = v[:f]
            out[prefix + "mlp.up_proj.weight"] = v[f:]
        elif name in ("layernorm_mlp.fc1_bias", "layernorm_mlp.fc2_bias"):
            if bool(torch.count_nonzero(v)):
                raise ValueError(f"{k} is non-zero: HF Llama has no MLP bias to hold it")
        else:
            out[k] = v
    return out


# ------------------------------------------------------------------------------------------------------------
# The two accelerate steps of Accelerator.prepare the reference depends on (accelerator.py:2098-2131,1818-1833).
# accelerate's own FP8 path needs an installed `transformer_engine` distribution, so the harness does them itself.

def convert_model(model: torch.nn.Module, to_fp8: bool = True, _convert_linear: bool = True) -> int:
    """nn.Linear -> llm_fp8_amd Linear for every Linear whose dims are multiples of 16
    (accelerate utils/transformer_engine.py:42-59).  The Parameter objects are re-used, so tied
    lm_head/embedding weights stay tied.  Unlike accelerate's version this one does
    not stop at the first non-convertible sibling (its early `return` at :50-51).  Returns #converted."""
    n = 0
    for name, child in list(model.named_children()):
        if isinstance(child, torch.nn.Linear) and to_fp8 and _convert_linear:
            if child.in_features % 16 or child.out_features % 16:
                continue
            new = te.Linear(child.in_features, child.out_features, bias=child.bias is not None,
                            params_dtype=child.weight.dtype, device=child.weight.device)
            new.weight = child.weight
            if child.bias is not None:
                new.bias = child.bias
            setattr(model, name, new)
            n += 1
        else:
            n += convert_model(child, to_fp8, _convert_linear)
    return n


def apply_fp8_autowrap(model: torch.nn.Module, fp8_recipe=None, use_during_eval: bool = False):
    """Wrap `model.forward` in an outer fp8_autocast that is active only in training mode
    (accelerate utils/transformer_engine.py:118-186)."""
    recipe = fp8_recipe if fp8_recipe is not None else DelayedScaling()  # accelerate default: HYBRID/1024/most_recent
    inner = model.forward

    def forward(*args, **kwargs):
        enabled = (use_during_eval or model.training) and torch.cuda.is_available()
        with te.fp8_autocast(enabled=enabled, fp8_recipe=recipe):
            return inner(*args, **kwargs)

    model.forward = forward
    model._fp8_outer_recipe = recipe
    return model


def outer_recipe_for_scenario(scenario: str):
    """FP8Handler.create_fp8_kwargs (train_fp8.py:126-165): the OUTER (accelerate) recipe -- used by lm_head."""
    if scenario == "default":
        return DelayedScaling(fp8_format=Format.HYBRID, amax_history_len=1024, amax_compute_algo="most_recent")
    # scenarios mxfp8 / hybrid: E4M3, history 16, "max", margin 0 (never MXFP8 for the outer recipe: Appendix C.9)
    return DelayedScaling(fp8_format=Format.E4M3, amax_history_len=16, amax_compute_algo="max", margin=0)
