"""Attention pieces with the `transformer_engine.pytorch` surface used at te_llama.py:45-56,65-66,77:
`MultiheadAttention(...)`(hidden_states, attention_mask=, rotary_pos_emb=) and
`attention.RotaryPositionEmbedding(dim)(max_seq_len)`.

The two projections (`layernorm_qkv`, `proj`) are on the FP8 hot path; the attention core is bf16 (the
reference's core is bf16 flash-attn through TE, README.md:27-28): the hand-written flash-style HIP kernels of
csrc/mi_attn.hip for the shapes of the reference's configs (causal, head_dim 64 / 128, seq % 128 == 0, no dropout),
torch `scaled_dot_product_attention` otherwise."""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from .module import DyHandoff, LayerNormLinear, Linear, handoff_readers

__all__ = ["RotaryPositionEmbedding", "apply_rotary_pos_emb", "DotProductAttention", "MultiheadAttention"]


class RotaryPositionEmbedding(torch.nn.Module):
    """TE-style RoPE table: forward(max_seq_len) -> [s, 1, 1, dim] float32 of angles (base 10000 by
    default, as the reference constructs it at te_llama.py:65)."""

    def __init__(self, dim: int, rotary_percent: float = 1.0, seq_len_interpolation_factor=None,
                 pretrained_max_position_embeddings=None, rotary_base: float = 10000.0):
        super().__init__()
        if rotary_percent < 1.0:
            dim = int(dim * rotary_percent)
        self.dim = dim
        self.seq_len_interpolation_factor = seq_len_interpolation_factor
        inv_freq = 1.0 / (rotary_base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)

    def forward(self, max_seq_len: int, offset: int = 0) -> torch.Tensor:
        seq = torch.arange(max_seq_len, device=self.inv_freq.device, dtype=torch.float32) + offset
        if self.seq_len_interpolation_factor is not None:
            seq = seq / self.seq_len_interpolation_factor
        freqs = torch.outer(seq, self.inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        return emb.reshape(emb.size(0), 1, 1, emb.size(1))


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(t: torch.Tensor, freqs: torch.Tensor, tensor_format: str = "sbhd") -> torch.Tensor:
    """t: [s,b,h,d] ("sbhd") or [b,s,h,d] ("bshd"); freqs: [s_max,1,1,d_rot] angles."""
    s = t.shape[0] if tensor_format == "sbhd" else t.shape[1]
    f = freqs[:s]
    if tensor_format == "bshd":
        f = f.transpose(0, 1)  # [1, s, 1, d]
    cos, sin = torch.cos(f).to(t.dtype), torch.sin(f).to(t.dtype)
    rot = f.shape[-1]
    tr, tp = t[..., :rot], t[..., rot:]
    out = tr * cos + _rotate_half(tr) * sin
    return out if tp.shape[-1] == 0 else torch.cat((out, tp), dim=-1)


_COS_SIN_CACHE = {}


def _cos_sin_tables(freqs: torch.Tensor, seq: int):
    """fp32 cos/sin [seq, D/2] of a TE angle table [s_max,1,1,D] (= cat(f, f)); cached per table and length."""
    key = (freqs.data_ptr(), freqs.shape, seq, str(freqs.device))
    hit = _COS_SIN_CACHE.get(key)
    if hit is None:
        half = freqs.shape[-1] // 2
        f = freqs[:seq, 0, 0, :half].float()
        hit = (torch.cos(f).contiguous(), torch.sin(f).contiguous())
        if len(_COS_SIN_CACHE) > 16:
            _COS_SIN_CACHE.clear()
        _COS_SIN_CACHE[key] = hit
    return hit


class _RoPESplitFn(torch.autograd.Function):
    """qkv [B,S,W] -> q [B,S,h,d], k, v [B,S,g,d] with RoPE on q, k: one HIP launch each way (mi_rope_qkv).
    `handoff` (module.DyHandoff, optional): the q|k|v projection that produced `qkv` has announced how its backward quantises
    grad_output; backward then emits that FP8 gradient directly (mi_rope_qkv_bwd_cast) and returns an unwritten placeholder."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, n_q, n_kv, d, handoff=None):
        B, S, W = qkv.shape
        x = qkv.reshape(B * S, W)
        x = x if x.is_contiguous() else x.contiguous()
        q, k, v = ops.rope_qkv_forward(x, cos, sin, n_q, n_kv, d, S)
        ctx.save_for_backward(cos, sin)
        ctx.meta = (B, S, n_q, n_kv, d)
        ctx.handoff = handoff
        ctx.act_ref = weakref.ref(qkv) if handoff is not None else None  # to see hooks / retain_grad put on it before backward
        return q.view(B, S, n_q, d), k.view(B, S, n_kv, d), v.view(B, S, n_kv, d)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        cos, sin = ctx.saved_tensors
        B, S, n_q, n_kv, d = ctx.meta
        T = B * S
        h = ctx.handoff
        if h is not None and h.offered() and handoff_readers(ctx.act_ref()):
            h = None  # somebody reads d(qkv): produce it for real
        if h is not None and h.offered() and d == 128 and T % (32 if h.mx else 8) == 0:
            args = (dq.reshape(T, n_q * d), dk.reshape(T, n_kv * d), dv.reshape(T, n_kv * d), cos, sin, n_q, n_kv, d, S)
            if h.mx:
                fp8 = ops.mxfp8_rope_bwd_quantize(*args, h.fmt, rowwise=h.want_y, colwise=h.want_t)
            else:
                fp8 = ops.rope_qkv_backward_cast(*args, h.scale, h.amax, h.fmt, want_y=h.want_y, want_t=h.want_t)
            g = torch.empty((B, S, (n_q + 2 * n_kv) * d), dtype=torch.bfloat16, device=dq.device)  # placeholder: never written
            h.put(fp8, g)
            return g, None, None, None, None, None, None
        g = ops.rope_qkv_backward(dq.reshape(T, n_q * d), dk.reshape(T, n_kv * d), dv.reshape(T, n_kv * d), cos, sin,
                                  n_q, n_kv, d, S)
        return g.view(B, S, -1), None, None, None, None, None, None


class _FlashAttnFn(torch.autograd.Function):
    """bshd attention core on mi_attn_fwd / mi_attn_bwd; saves q, k, v, o and the log-sum-exp, never the scores."""

    @staticmethod
    def forward(ctx, q, k, v, scale, causal):
        o, lse = ops.attn_fwd(q, k, v, scale, causal)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale, ctx.causal = scale, causal
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = do if (do.stride(3) == 1 and do.stride(2) == do.shape[3] and do.stride(0) == do.shape[1] * do.stride(1)) else do.contiguous()
        dq, dk, dv = ops.attn_bwd(do, q, k, v, o, lse, ctx.scale, ctx.causal)
        return dq, dk, dv, None, None


def _flash_ok(q, k, v, causal: bool, dropout: float) -> bool:
    if not (q.is_cuda and q.dtype == k.dtype == v.dtype == torch.bfloat16 and dropout == 0.0 and causal):
        return False
    B, S, H, D = q.shape
    if D not in (64, 128) or S % 128 or S < 128 or H % k.shape[2]:
        return False
    return all(t.stride(3) == 1 and t.stride(2) == D and t.stride(0) == S * t.stride(1) and t.stride(1) % 8 == 0
               and S * t.stride(1) * 2 < 2 ** 31 and t.data_ptr() % 16 == 0 for t in (q, k, v))


class DotProductAttention(torch.nn.Module):
    """bf16 attention core.  q [b,s,h,d], k/v [b,s,g,d] (bshd) -> [b,s,h*d]."""

    def __init__(self, num_attention_heads: int, kv_channels: int, num_gqa_groups: Optional[int] = None,
                 attention_dropout: float = 0.0, attn_mask_type: str = "causal", qkv_format: str = "bshd", **_ignored):
        super().__init__()
        self.h, self.d = num_attention_heads, kv_channels
        self.g = num_gqa_groups or num_attention_heads
        self.p, self.attn_mask_type, self.qkv_format = attention_dropout, attn_mask_type, qkv_format

    def forward(self, q, k, v, attention_mask=None):
        if self.qkv_format == "sbhd":
            q, k, v = (t.transpose(0, 1) for t in (q, k, v))
        causal = self.attn_mask_type in ("causal", "padding_causal")
        if _flash_ok(q, k, v, causal, self.p if self.training else 0.0):
            o = _FlashAttnFn.apply(q, k, v, self.d ** -0.5, True)
            if self.qkv_format == "sbhd":
                o = o.transpose(0, 1)
            return o.reshape(*o.shape[:2], self.h * self.d)
        q, k, v = (t.transpose(1, 2) for t in (q, k, v))  # [b, h, s, d]
        mask = None
        if not causal and attention_mask is not None:
            mask = ~attention_mask if attention_mask.dtype == torch.bool else attention_mask  # TE: True = masked out
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.p if self.training else 0.0,
                                           is_causal=causal, enable_gqa=(self.g != self.h))
        o = o.transpose(1, 2)  # [b, s, h, d]
        if self.qkv_format == "sbhd":
            o = o.transpose(0, 1)
        return o.reshape(*o.shape[:2], self.h * self.d)


class MultiheadAttention(torch.nn.Module):
    """`layernorm_qkv` (LayerNormLinear with query_/key_/value_weight) -> RoPE -> attention -> `proj` (Linear).

    Arguments follow te.pytorch.MultiheadAttention as called at te_llama.py:45-56.  The default mask type is
    causal and, as in TE, a padding `attention_mask` passed with the causal type is ignored
    (SURVEY.md 8a row a5)."""

    def __init__(self, hidden_size: int, num_attention_heads: int, kv_channels: Optional[int] = None,
                 attention_dropout: float = 0.1, layernorm_epsilon: float = 1e-5, init_method=None,
                 output_layer_init_method=None, attn_mask_type: str = "causal", num_gqa_groups: Optional[int] = None,
                 input_layernorm: bool = False, attention_type: str = "self", fuse_qkv_params: bool = False,
                 zero_centered_gamma: bool = False, qkv_weight_interleaved: bool = True, bias: bool = True,
                 normalization: str = "LayerNorm", qkv_format: str = "sbhd", params_dtype=None, device="cuda",
                 **_ignored):
        super().__init__()
        assert attention_type == "self", "only self-attention is on the reference path"
        assert qkv_format in ("sbhd", "bshd")
        self.hidden_size, self.h = hidden_size, num_attention_heads
        self.d = kv_channels or hidden_size // num_attention_heads
        self.g = num_gqa_groups or num_attention_heads
        self.qkv_format, self.input_layernorm = qkv_format, input_layernorm
        q_out, kv_out = self.h * self.d, self.g * self.d
        self.split = (q_out, kv_out, kv_out)
        if fuse_qkv_params:
            psplit = None
        else:
            psplit = {"query_": q_out, "key_": kv_out, "value_": kv_out}
        if input_layernorm:
            self.layernorm_qkv = LayerNormLinear(hidden_size, sum(self.split), eps=layernorm_epsilon, bias=bias,
                                                 normalization=normalization, parameters_split=psplit,
                                                 params_dtype=params_dtype, device=device,
                                                 zero_centered_gamma=zero_centered_gamma, init_method=init_method)
        else:
            self.qkv = Linear(hidden_size, sum(self.split), bias=bias, params_dtype=params_dtype, device=device,
                              init_method=init_method)
        self.core_attention = DotProductAttention(self.h, self.d, self.g, attention_dropout, attn_mask_type, qkv_format)
        self.proj = Linear(q_out, hidden_size, bias=bias, params_dtype=params_dtype, device=device,
                           init_method=output_layer_init_method or init_method)

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                rotary_pos_emb=None, _with_skip: bool = False, _rstd=None, **_ignored):
        """`_with_skip` (extension): returns (out, skip); `_rstd`: statistics hand-off, see LayerNormLinear.forward."""
        if _with_skip:
            if not self.input_layernorm:
                return self._attend(self.qkv(hidden_states), attention_mask, rotary_pos_emb), hidden_states
            # the rotary split's backward hands the projection its grad_output already in FP8 (module.DyHandoff): only
            # when the two are wired back to back right here
            handoff = DyHandoff() if (rotary_pos_emb is not None and hidden_states.is_cuda and torch.is_grad_enabled()
                                      and os.environ.get("LLM_FP8_AMD_NO_DY_HANDOFF") != "1") else None
            qkv, skip = self.layernorm_qkv(hidden_states, _with_skip=True, _rstd=_rstd, _dy_handoff=handoff)
            return self._attend(qkv, attention_mask, rotary_pos_emb, handoff), skip
        qkv = self.layernorm_qkv(hidden_states) if self.input_layernorm else self.qkv(hidden_states)
        return self._attend(qkv, attention_mask, rotary_pos_emb)

    def _attend(self, qkv: torch.Tensor, attention_mask, rotary_pos_emb, handoff=None) -> torch.Tensor:
        if (rotary_pos_emb is not None and not isinstance(rotary_pos_emb, (tuple, list)) and qkv.is_cuda
                and qkv.dtype == torch.bfloat16 and self.qkv_format == "bshd" and self.d % 16 == 0
                and rotary_pos_emb.shape[-1] == self.d):
            cos, sin = _cos_sin_tables(rotary_pos_emb, qkv.shape[1])
            q, k, v = _RoPESplitFn.apply(qkv, cos, sin, self.h, self.g, self.d, handoff)
            return self.proj(self.core_attention(q, k, v, attention_mask))
        if handoff is not None:
            handoff.withdraw()  # not the fused rotary path: the projection quantises its own grad_output
        q, k, v = torch.split(qkv, self.split, dim=-1)
        a, b = qkv.shape[0], qkv.shape[1]
        q = q.reshape(a, b, self.h, self.d)
        k = k.reshape(a, b, self.g, self.d)
        v = v.reshape(a, b, self.g, self.d)
        if rotary_pos_emb is not None:
            fq, fk = rotary_pos_emb if isinstance(rotary_pos_emb, (tuple, list)) else (rotary_pos_emb, rotary_pos_emb)
            q = apply_rotary_pos_emb(q, fq, self.qkv_format)
            k = apply_rotary_pos_emb(k, fk, self.qkv_format)
        ctx = self.core_attention(q, k, v, attention_mask)
        return self.proj(ctx)
