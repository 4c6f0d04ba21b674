"""`ClippedAdamW`: the reference's `clip_grad_norm_(params, max_norm)` + `AdamW(fused=True).step()` pair
(train_fp8.py:288-291) as two streaming HIP passes: one reproducible squared-norm reduction over the gradients and one
AdamW update with the clip coefficient folded in (the gradients are never rescaled in place).  bf16 parameters with bf16
`exp_avg` / `exp_avg_sq`, fp32 math -- the state layout of torch's fused AdamW, so `state_dict()` is interchangeable."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class ClippedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm: Optional[float] = 1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm
        self._partials = None
        self.last_grad_norm: Optional[torch.Tensor] = None

    def _grads(self):
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if not (p.is_cuda and p.dtype == torch.bfloat16 and p.grad.dtype == torch.bfloat16):
                        raise RuntimeError("ClippedAdamW handles bf16 parameters resident on the GPU only")
                    out.append((group, p))
        return out

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        items = self._grads()
        if not items:
            return None
        lib = _lib.load()
        dev = items[0][1].device
        st = torch.cuda.current_stream().cuda_stream
        coef_ptr = None
        if self.max_grad_norm is not None:
            # one partial per 8 Ki elements (at most 2048 per tensor, 8 workgroups per CU): tiny tensors cost one workgroup
            nblks = [max(1, min(2048, (p.numel() + 8191) // 8192)) for _, p in items]
            need = sum(nblks)
            if self._partials is None or self._partials.numel() < need or self._partials.device != dev:
                self._partials = torch.empty(need, dtype=torch.float32, device=dev)
            part = self._partials[:need]
            off = 0
            for (_, p), nb in zip(items, nblks):
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _lib.check(lib.mi_sumsq_bf16(g.data_ptr(), g.numel(), part[off:].data_ptr(), nb, st), "mi_sumsq_bf16")
                off += nb
            total = part.sum(dtype=torch.float32).sqrt()
            self.last_grad_norm = total
            coef = (self.max_grad_norm / (total + 1e-6)).clamp(max=1.0).reshape(1)  # clip_grad_norm_'s coefficient
            coef_ptr = coef.data_ptr()
        for group, p in items:
            state = self.state[p]
            if not state:
                state["step"] = 0
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["step"] += 1
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            assert p.is_contiguous()
            b1, b2 = group["betas"]
            rc = lib.mi_adamw_bf16(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                   p.numel(), coef_ptr, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                   int(state["step"]), st)
            _lib.check(rc, "mi_adamw_bf16")
        return None
