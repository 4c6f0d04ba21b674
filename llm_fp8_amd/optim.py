"""`ClippedAdamW`: the reference's `clip_grad_norm_(params, max_norm)` + `AdamW(fused=True).step()` pair
(train_fp8.py:288-291) as two streaming HIP passes: one reproducible squared-norm reduction over the gradients and one
AdamW update with the clip coefficient folded in (the gradients are never rescaled in place), each ONE multi-tensor launch
per parameter group (device tables of tensor addresses + a chunk list).  bf16 parameters with bf16
`exp_avg` / `exp_avg_sq`, fp32 math -- the state layout of torch's fused AdamW, so `state_dict()` is interchangeable."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib


class ClippedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm: Optional[float] = 1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm
        self._plans = {}
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

    CHUNK = 65536  # elements per workgroup of the multi-tensor kernels (128 KiB of bf16)

    TILE = 128  # a sink tensor's chunks are TILE x TILE tiles of the weight matrix (mi_adamw_cast_bf16_multi)

    @staticmethod
    def _sink_of(p):
        """(sink, row offset, rows) when the FP8 copies of this weight are kept current by the optimiser (module.WeightSink)."""
        if os.environ.get("LLM_FP8_AMD_NO_OPT_WCAST") == "1":  # the switch of module.weight_sinks_enabled: the forward casts
            return None
        s = getattr(p, "_mi_fp8_sink", None)
        sh = getattr(p, "_mi_shard_of", None)
        if sh is not None:  # a row shard of a master weight (distributed.ShardedFP8DP): rows [r0, r0 + n) of the full operand part
            full, sr0, sn = sh
            fs = getattr(full, "_mi_fp8_sink", None)
            s = None if fs is None else (fs[0], fs[1] + sr0, sn)
        if s is None or not p.is_contiguous() or p.dim() != 2 or not ClippedAdamW._aligned16(p):
            return None
        sink, r0, n = s
        if n != p.shape[0] or sink.w8.shape[1] != p.shape[1] or p.shape[0] % 8 or p.shape[1] % 8:
            return None
        return s

    @staticmethod
    def _aligned16(p) -> bool:
        """The tile path of the *_cast kernels issues 16-byte loads / stores on the parameter, its gradient and both moments
        unconditionally; a tensor at an odd storage offset (a loaded optimiser state, a `.grad` view) takes the flat path."""
        if p.data_ptr() % 16:
            return False
        g = p.grad
        return g is None or g.data_ptr() % 16 == 0

    @staticmethod
    def _mx_sink_of(p):
        """(sink, row offset, rows) when the MXFP8 copies of this weight are kept current by the optimiser (module.MXWeightSink)."""
        if os.environ.get("LLM_FP8_AMD_NO_OPT_WCAST") == "1":
            return None
        s = getattr(p, "_mi_mx_sink", None)
        sh = getattr(p, "_mi_shard_of", None)
        if sh is not None:  # a row shard of a master weight (distributed.ShardedFP8DP)
            full, sr0, sn = sh
            fs = getattr(full, "_mi_mx_sink", None)
            s = None if fs is None else (fs[0], fs[1] + sr0, sn)
        if s is None or not p.is_contiguous() or p.dim() != 2 or not ClippedAdamW._aligned16(p):
            return None
        sink, r0, n = s
        if n != p.shape[0] or sink.w8.shape[1] != p.shape[1] or p.shape[0] % 32 or p.shape[1] % 32 or r0 % 32:
            return None
        return s

    def _plan(self, group_items, dev, mx=False):
        """Device tables of one parameter group for the multi-tensor kernels.  The chunk lists only depend on the tensor sizes
        (and on which tensors have an FP8 sink) and are built once; the address table is re-uploaded when an address changed
        (gradients are new tensors every step, though the caching allocator usually hands back the same blocks)."""
        # one plan per PARTITION (the parameters of a group that share a step count), not per group: two partitions of one
        # group must not share address table / partial sums
        key = (id(group_items[0][0]), tuple(id(p) for _, p in group_items), mx)
        sizes = tuple(p.numel() for _, p in group_items)
        sinks = tuple((self._mx_sink_of(p) if mx else self._sink_of(p)) for _, p in group_items)
        sink_sig = tuple(None if s is None else (id(s[0]), s[1]) for s in sinks)
        plan = self._plans.get(key)
        if plan is None or plan["sizes"] != sizes or plan["sink_sig"] != sink_sig:
            refs, refs_cast = [], []
            for t, (n, (_, p), s) in enumerate(zip(sizes, group_items, sinks)):
                flat = [(t, c) for c in range((n + self.CHUNK - 1) // self.CHUNK)]
                refs.extend(flat)
                if s is None:
                    refs_cast.extend(flat)
                else:
                    tr, tc = (p.shape[0] + self.TILE - 1) // self.TILE, (p.shape[1] + self.TILE - 1) // self.TILE
                    refs_cast.extend((t, c) for c in range(tr * tc))
            chunks = torch.tensor(refs, dtype=torch.int32).to(dev)
            any_sink = any(s is not None for s in sinks)
            plan = {"sizes": sizes, "sink_sig": sink_sig, "chunks": chunks, "n_chunks": len(refs), "addr": None,
                    "chunks_cast": torch.tensor(refs_cast, dtype=torch.int32).to(dev) if any_sink else None, "n_chunks_cast": len(refs_cast),
                    "table": torch.empty((12, len(sizes)), dtype=torch.int64, device=dev),
                    "host": [torch.empty((12, len(sizes)), dtype=torch.int64).pin_memory() for _ in range(2)], "flip": 0,
                    "uploaded": [None, None],
                    "partials": torch.empty(len(refs), dtype=torch.float64, device=dev)}
            self._plans[key] = plan
        rows = [[], [], [], [], list(sizes), [], [], [], [], [], [], []]
        for (_, p), s in zip(group_items, sinks):
            st = self.state[p]
            rows[0].append(p.data_ptr())
            rows[1].append(p.grad.data_ptr())
            rows[2].append(st["exp_avg"].data_ptr())
            rows[3].append(st["exp_avg_sq"].data_ptr())
            if s is None:
                for r in range(5, 12):
                    rows[r].append(0)
            elif mx:  # mi_adamw_mxcast_bf16_multi: 6 y_row  7 s_row  8 y_colT  9 s_colT  10 ldr  11 unused
                sink, r0, _ = s
                K, N = p.shape[1], sink.w8.shape[0]
                rows[5].append(K)
                rows[6].append(sink.w8.data_ptr() + r0 * K)
                rows[7].append(sink.sc.data_ptr() + r0)
                rows[8].append(sink.wt8.data_ptr() + r0)
                rows[9].append(sink.sct.data_ptr() + (r0 // 32) * K)
                rows[10].append(N)
                rows[11].append(0)
            else:
                sink, r0, _ = s
                K, N = p.shape[1], sink.w8.shape[0]
                rows[5].append(K)
                dbg = os.environ.get("MI_DEBUG_SINK", "full")  # tools/bench_adamw.py: mask the FP8 outputs off (timing only)
                rows[6].append(0 if dbg == "none" else sink.w8.data_ptr() + r0 * K)
                rows[7].append(0 if dbg in ("none", "noT") else sink.w8t.data_ptr() + r0)
                rows[8].append(K)
                rows[9].append(N)
                rows[10].append(sink.scale.data_ptr())
                rows[11].append(sink.amax.data_ptr())
        plan["sinks"] = sinks
        plan["mx"] = mx
        if plan["addr"] != rows:
            # two pinned staging buffers, each guarded by an event recorded behind its last upload: the host may run several
            # steps ahead of the device, and rewriting a buffer whose async H2D copy has not run yet would hand the kernels
            # the addresses of a LATER step
            slot = plan["flip"]
            plan["flip"] ^= 1
            if plan["uploaded"][slot] is not None:
                plan["uploaded"][slot].synchronize()
            host = plan["host"][slot]
            host.copy_(torch.tensor(rows, dtype=torch.int64))
            plan["table"].copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            plan["uploaded"][slot] = ev
            plan["addr"] = rows
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        items = self._grads()
        if not items:
            return None
        lib = _lib.load()
        dev = items[0][1].device
        st = torch.cuda.current_stream().cuda_stream
        by_group = {}
        for group, p in items:
            state = self.state[p]
            if not state:
                state["step"] = 0
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if torch.is_tensor(state["step"]):  # a state_dict written by torch's AdamW keeps `step` as a tensor
                state["step"] = int(state["step"].item())
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            assert p.is_contiguous()
            by_group.setdefault((id(group), state["step"]), []).append((group, p))
        # The squared norm is taken over the partitions above (ONE fixed summation order, whatever sinks exist); the update of a
        # partition that holds weights with an MXFP8 sink is split in two launches (another kernel and another table layout).
        plans = [(gi, self._plan(gi, dev)) for gi in by_group.values()]
        updates = []
        for gi, plan in plans:
            gi_mx = [it for it in gi if self._mx_sink_of(it[1]) is not None]
            if gi_mx:
                gi_rest = [it for it in gi if self._mx_sink_of(it[1]) is None]
                updates.append((gi_mx, self._plan(gi_mx, dev, mx=True)))
                if gi_rest:
                    updates.append((gi_rest, self._plan(gi_rest, dev)))
            else:
                updates.append((gi, plan))
        coef_ptr = None
        if self.max_grad_norm is not None:
            # squared norm: one launch per parameter group, one FLOAT64 partial per 64 Ki-element chunk (exact squares, fp64 sums)
            for gi, plan in plans:
                _lib.check(lib.mi_sumsq_bf16_multi(plan["table"].data_ptr(), len(gi), plan["chunks"].data_ptr(), plan["n_chunks"],
                                                   self.CHUNK, plan["partials"].data_ptr(), st), "mi_sumsq_bf16_multi")
            # Partials and totals stay in FLOAT64 and the norm is rounded to fp32 once: the wrappers partition the parameters
            # differently (distributed.ShardedFP8DP: one group of row shards + the replicated rest, summed over ranks), and in
            # fp32 the association of those sums moves the last bit of the clip coefficient -- a handful of weights then round
            # differently and a sharded run stops being bitwise the replicated one (seen once the MLP bias fusion changed the
            # gradients' low bits).  In float64 the order only matters below 2^-52 of the total.
            total = shard_total = None
            for gi, plan in plans:
                t = plan["partials"].sum(dtype=torch.float64)
                if gi[0][0].get("sharded", False):  # row shards of ShardedFP8DP: every rank holds different rows
                    shard_total = t if shard_total is None else shard_total + t
                else:
                    total = t if total is None else total + t
            if shard_total is not None:
                import torch.distributed as dist
                grp = next((gi[0][0].get("dp_group") for gi, _ in plans if gi[0][0].get("sharded", False)), None)
                if dist.is_available() and dist.is_initialized() and dist.get_world_size(grp) > 1:
                    dist.all_reduce(shard_total, op=dist.ReduceOp.SUM, group=grp)
                total = shard_total if total is None else total + shard_total
            total = total.sqrt().to(torch.float32)
            self.last_grad_norm = total
            coef = (self.max_grad_norm / (total + 1e-6)).clamp(max=1.0).reshape(1)  # clip_grad_norm_'s coefficient
            coef_ptr = coef.data_ptr()
        touched, sinks_seen = set(), {}
        for gi, plan in updates:
            group = gi[0][0]
            for _, p in gi:
                self.state[p]["step"] += 1
            b1, b2 = group["betas"]
            if plan["chunks_cast"] is not None:  # some weights of this partition have FP8 sinks: update + cast in one pass
                fn = lib.mi_adamw_mxcast_bf16_multi if plan["mx"] else lib.mi_adamw_cast_bf16_multi
                rc = fn(plan["table"].data_ptr(), len(gi), plan["chunks_cast"].data_ptr(), plan["n_chunks_cast"],
                                                  self.CHUNK, coef_ptr, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                                  int(self.state[gi[0][1]]["step"]), st)
                _lib.check(rc, "mi_adamw_mxcast_bf16_multi" if plan["mx"] else "mi_adamw_cast_bf16_multi")
            else:
                rc = lib.mi_adamw_bf16_multi(plan["table"].data_ptr(), len(gi), plan["chunks"].data_ptr(), plan["n_chunks"], self.CHUNK,
                                             coef_ptr, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                             int(self.state[gi[0][1]]["step"]), st)
                _lib.check(rc, "mi_adamw_bf16_multi")
            # the kernels write through raw addresses: tell autograd (and every cache keyed on `_version`, wcast.py)
            torch.autograd.graph.increment_version([p for _, p in gi])
            touched.update(id(p) for _, p in gi)
            for s in plan["sinks"]:
                if s is not None:
                    sinks_seen[id(s[0])] = s[0]
        for sink in sinks_seen.values():  # copies are current only if EVERY part of the operand was rewritten in this step
            if all(id(w) in touched for w, _, _ in sink.parts):
                sink.mark()
            else:
                sink.stamp = None
        return None
