"""Fused causal-LM cross-entropy for the harness: same value as HF's `ForCausalLMLoss` (shift labels by one, mean over
tokens with label != -100) computed by two streaming HIP kernels directly on the bf16 logits of the FP8 lm_head."""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib


class _CEFn(torch.autograd.Function):
    """`handoff` (pytorch.module.DyHandoff of the lm_head that produced the logits, optional): backward then emits d(logits)
    as the FP8 copies that layer's backward needs (mi_ce_backward_cast) and returns an unwritten placeholder to autograd."""

    @staticmethod
    def forward(ctx, logits2d: torch.Tensor, labels1d: torch.Tensor, handoff=None):
        T, V = logits2d.shape
        st = torch.cuda.current_stream().cuda_stream
        lse = torch.empty(T, dtype=torch.float32, device=logits2d.device)
        rows = torch.empty(T, dtype=torch.float32, device=logits2d.device)
        _lib.check(_lib.load().mi_ce_forward(logits2d.data_ptr(), labels1d.data_ptr(), lse.data_ptr(), rows.data_ptr(), T, V, st),
                   "mi_ce_forward")
        n_valid = (labels1d != -100).sum().clamp(min=1).to(torch.float32)
        ctx.save_for_backward(logits2d, labels1d, lse, n_valid)
        ctx.handoff = handoff
        return rows.sum() / n_valid

    @staticmethod
    def backward(ctx, dloss):
        logits2d, labels1d, lse, n_valid = ctx.saved_tensors
        T, V = logits2d.shape
        gscale = (dloss.to(torch.float32) / n_valid).reshape(1).contiguous()
        st = torch.cuda.current_stream().cuda_stream
        h = ctx.handoff
        if h is not None and h.offered():
            from .pytorch.module import handoff_readers
            if handoff_readers(None if h.act_ref is None else h.act_ref()):
                h = None  # somebody reads d(logits): produce it for real
        if h is not None and h.offered() and not h.mx and T % 8 == 0:
            y = torch.empty((T, V), dtype=torch.uint8, device=logits2d.device) if h.want_y else None
            yt = torch.empty((V, T), dtype=torch.uint8, device=logits2d.device) if h.want_t else None
            _lib.check(_lib.load().mi_ce_backward_cast(logits2d.data_ptr(), labels1d.data_ptr(), lse.data_ptr(), gscale.data_ptr(),
                                                       None if y is None else y.data_ptr(), None if yt is None else yt.data_ptr(),
                                                       h.scale.data_ptr(), h.amax.data_ptr(), T, V, h.fmt, st), "mi_ce_backward_cast")
            d = torch.empty_like(logits2d)  # placeholder: never written, never read (the lm_head's backward takes the FP8 copies)
            h.put((y, yt), d)
            return d, None, None
        d = torch.empty_like(logits2d)
        _lib.check(_lib.load().mi_ce_backward(logits2d.data_ptr(), labels1d.data_ptr(), lse.data_ptr(), gscale.data_ptr(),
                                              d.data_ptr(), T, V, st), "mi_ce_backward")
        return d, None, None


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor, vocab_size: int = None, num_items_in_batch=None,
                   ignore_index: int = -100, shift_labels=None, **_unused) -> torch.Tensor:
    """Drop-in for `transformers.loss.loss_utils.ForCausalLMLoss` (installed as `model.loss_function` by
    llm_fp8_amd.train.prepare_model on the GPU path).  Falls back to HF's implementation for anything it does not cover."""
    ok = (logits.is_cuda and logits.dtype == torch.bfloat16 and ignore_index == -100 and num_items_in_batch is None
          and logits.shape[-1] % 8 == 0)
    if not ok:
        from transformers.loss.loss_utils import ForCausalLMLoss
        return ForCausalLMLoss(logits, labels, vocab_size, num_items_in_batch=num_items_in_batch, ignore_index=ignore_index,
                               shift_labels=shift_labels, **_unused)
    if shift_labels is None:
        shift_labels = torch.nn.functional.pad(labels, (0, 1), value=-100)[..., 1:]
    V = logits.shape[-1]
    l2 = logits.reshape(-1, V)
    handoff = getattr(logits, "_mi_dy_handoff", None)
    if not l2.is_contiguous() or os.environ.get("LLM_FP8_AMD_NO_DY_HANDOFF") == "1":
        l2, handoff = l2.contiguous(), None  # (a copy sits between the lm_head and the loss: no hand-off)
    if handoff is not None:
        handoff.act_ref = weakref.ref(logits)  # hooks / retain_grad() put on the logits before backward turn the hand-off off
    return _CEFn.apply(l2, shift_labels.reshape(-1).to(torch.int64).contiguous(), handoff)
