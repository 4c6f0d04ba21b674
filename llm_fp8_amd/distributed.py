"""Replicated data parallelism over a flat gradient arena -- the N > 1 path for every model whose whole training state
(weights + gradients + AdamW moments) fits one MI355X's 288 GB of HBM3E, i.e. all of BASELINE.json's configurations.

The reference wraps the model in FSDP FULL_SHARD whenever more than one GPU is visible (train_multi_gpu.py:137-146,
:433-445) because its target GPUs cannot hold the replicated state.  On MI355X they can, so nothing has to be sharded:
the only exchange step of the path is the gradient reduction (SURVEY.md 8e "only where the run actually shards").
FSDP / DDP stay available through `--sharding_mode fsdp_full | ddp` (train.wrap_distributed).

Layout: one contiguous bf16 (per dtype) arena holds every parameter's gradient, ordered by when backward produces it
(last module first; q|k|v of one projection adjacent, so the fused wgrad GEMM of `_FP8LinearFn` writes all three in one
launch -- module._wgrad_out).  The arena is cut into buckets of >= `bucket_mb` (a quarter of that for the last 15 % of the
arena, which backward finishes last); when the last gradient of a bucket has
been accumulated the bucket is all-reduced (AVG) on RCCL's stream while backward continues: few, large collectives, as
xGMI's point-to-point links want (7 x ~153 GB/s per GPU, ring collectives are per-link bound).  The optimiser
(optim.ClippedAdamW) then reads gradients at addresses that never change between steps.

Embedding tables are reduced row-sparsely.  An embedding's gradient touches at most batch x seq rows, but it is the LAST
gradient of backward and (tied to lm_head, or not) the largest parameter: as a dense all-reduce it would be the exposed tail
of every step (Llama-3.2-3B: 788 MB; over one xGMI link at N = 2 that is ~6 ms).  Instead the embedding's backward only
hands (token ids, dY rows) to the wrapper; the table's bucket therefore completes as soon as lm_head's wgrad has landed
(start of backward, fully overlapped), and at the end of backward the ranks all-gather ids + rows (50 MB per rank) and add
the same deterministic scatter (aten embedding_dense_backward: sort + segmented sum) to their copy of the averaged gradient.
"""
from __future__ import annotations

import functools
import os
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

_ALIGN = 64  # elements: slots start on 128-byte boundaries (bf16); sizes that are multiples of it stay gap-free


class _Bucket:
    __slots__ = ("arena", "start", "end", "params", "pending", "shard_of")

    def __init__(self, arena: int, start: int):
        self.arena, self.start, self.end, self.params, self.pending = arena, start, start, [], 0
        self.shard_of = None  # ShardedFP8DP: the one row-sharded weight of this bucket (reduce-scatter instead of all-reduce)


def _param_groups_in_backward_order(module: torch.nn.Module) -> List[List[torch.nn.Parameter]]:
    """Direct parameters of each sub-module (registration order inside a module, so query|key|value stay adjacent and in
    order), modules in reverse pre-order ~ the order in which backward finishes them.  Shared parameters appear once."""
    seen, groups = set(), []
    for m in module.modules():
        g = [p for p in m._parameters.values() if p is not None and p.requires_grad and id(p) not in seen]
        seen.update(id(p) for p in g)
        if g:
            groups.append(g)
    return groups[::-1]


class _DeferredEmbeddingGrad(torch.autograd.Function):
    """Embedding lookup whose weight gradient is not produced by autograd: backward leaves (ids, dY) with the wrapper, which
    reduces them row-sparsely at the end of the pass.  `anchor` is a dummy that requires grad so that backward runs."""

    @staticmethod
    def forward(ctx, anchor, ids, weight, dp, emb):
        ctx.dp, ctx.emb, ctx.ids = dp, emb, ids
        return F.embedding(ids, weight, emb.padding_idx)

    @staticmethod
    def backward(ctx, dy):
        ctx.dp._defer_embedding_grad(ctx.emb, ctx.ids, dy)
        return None, None, None, None, None


class GradArenaDP(torch.nn.Module):
    """`module` replicated on every rank of `process_group`; gradients averaged bucket by bucket during backward."""

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: float = 256.0, broadcast: bool = True,
                 sparse_embedding_grads: bool = True):
        super().__init__()
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradArenaDP needs an initialised torch.distributed process group")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        backend = dist.get_backend(process_group)
        # RCCL averages in the collective; gloo (CPU rehearsal) has no AVG: SUM, then one scaling pass over the arena
        self._avg_in_collective = backend == "nccl"
        self._sync = True
        self._works: list = []
        self._callback_queued = False
        self.arenas: List[torch.Tensor] = []
        self.buckets: List[_Bucket] = []
        self._bucket_of: Dict[int, _Bucket] = {}
        self._prepare_sharding()
        self._build(_param_groups_in_backward_order(module), int(bucket_mb * (1 << 20)))
        if broadcast and self.world > 1:
            self._broadcast_state()
        self._sparse: list = []          # (embedding module, ids, dY) left by this pass' embedding backwards
        self._anchor = None
        if sparse_embedding_grads and (self.world > 1 or _FORCE_COLLECTIVES):
            self._install_sparse_embeddings()

    # ------------------------------------------------------------------------------------------------ construction
    def _prepare_sharding(self):
        """Hook for ShardedFP8DP (decides which weights get a bucket of their own and a reduce-scatter)."""

    def _own_bucket(self, p) -> bool:
        return False

    def _build(self, groups, bucket_bytes: int):
        keys: Dict[tuple, int] = {}
        sizes: List[int] = []
        plan = []  # (param, arena index, offset)
        for g in groups:
            for p in g:
                key = (p.device, p.dtype)
                a = keys.setdefault(key, len(keys))
                if a == len(sizes):
                    sizes.append(0)
                plan.append((p, a, sizes[a]))
                sizes[a] += -(-p.numel() // _ALIGN) * _ALIGN
        self.arenas = [torch.zeros(max(n, _ALIGN), dtype=dt, device=dev) for (dev, dt), n in zip(keys, sizes)]
        open_bucket: Dict[int, _Bucket] = {}
        for p, a, off in plan:
            arena = self.arenas[a]
            p._mi_grad_buf = arena[off:off + p.numel()].view(p.shape)
            p._mi_grad_slot = (arena, off)
            b = open_bucket.get(a)
            own = self._own_bucket(p)
            if b is not None and (own or p.numel() * arena.element_size() >= bucket_bytes):
                # a parameter that fills a bucket by itself (the embedding table, whose gradient is complete only at the very
                # end of backward) does not hold back the reduction of what came before it
                del open_bucket[a]
                b = None
            if b is None:
                b = open_bucket[a] = _Bucket(a, off)
                self.buckets.append(b)
            b.params.append(p)
            b.end = off + -(-p.numel() // _ALIGN) * _ALIGN
            self._bucket_of[id(p)] = b
            # the buckets that close last (the first layers: the tail of backward) are a quarter of the size, so that the
            # all-reduce still in flight when backward ends is short
            cap = bucket_bytes if off < 0.85 * sizes[a] else max(bucket_bytes // 4, 1)
            if own:
                b.shard_of = p
            if own or (b.end - b.start) * arena.element_size() >= cap:
                del open_bucket[a]
            p.register_post_accumulate_grad_hook(self._on_grad)
        for b in self.buckets:
            b.pending = len(b.params)

    def _broadcast_state(self):
        """Rank 0's parameters and buffers everywhere (what DDP / FSDP `sync_module_states` do at wrap time)."""
        with torch.no_grad():
            for t in list(self.module.parameters()) + list(self.module.buffers()):
                if t.numel():
                    dist.broadcast(t.data, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                   group=self.group)

    # ------------------------------------------------------------------------------------------------ embeddings
    def _install_sparse_embeddings(self):
        for m in self.module.modules():
            if (type(m) is torch.nn.Embedding and m.weight.requires_grad and id(m.weight) in self._bucket_of
                    and m.max_norm is None and not m.sparse and not m.scale_grad_by_freq):
                m.forward = functools.partial(self._embedding_forward, m)  # instance attribute: shadows Embedding.forward

    def _embedding_forward(self, emb, ids):
        if not (torch.is_grad_enabled() and emb.weight.requires_grad):
            return F.embedding(ids, emb.weight, emb.padding_idx)
        if self._anchor is None or self._anchor.device != emb.weight.device:
            self._anchor = torch.zeros(1, device=emb.weight.device, requires_grad=True)
        return _DeferredEmbeddingGrad.apply(self._anchor, ids, emb.weight.detach(), self, emb)

    def _defer_embedding_grad(self, emb, ids, dy):
        self._sparse.append((emb, ids.reshape(-1), dy.reshape(-1, dy.shape[-1])))
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _reduce_sparse(self):
        """All ranks: gather every rank's (ids, rows), then the same deterministic scatter into the local copy of the averaged
        gradient -- identical inputs and a fixed summation order, so the replicas stay bit-identical."""
        by_emb: Dict[int, list] = {}
        for emb, ids, dy in self._sparse:
            by_emb.setdefault(id(emb), [emb, [], []])
            by_emb[id(emb)][1].append(ids)
            by_emb[id(emb)][2].append(dy)
        self._sparse = []
        with torch.no_grad():
            for emb, ids_l, dy_l in by_emb.values():
                p = emb.weight
                ids = torch.cat(ids_l) if len(ids_l) > 1 else ids_l[0]
                dy = (torch.cat(dy_l) if len(dy_l) > 1 else dy_l[0]).to(p.dtype).contiguous()
                if self.world > 1:
                    # Ranks may hold different token counts: the reference's collator pads every batch to ITS OWN longest
                    # sequence (data.py:59-63), and micro-batches accumulate under no_sync.  Exchange the counts first (one
                    # tiny all_gather), pad ids with -1 (skipped by the scatter) and dy with zero rows up to the largest.
                    cnt = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
                    cnts = torch.empty(self.world, dtype=torch.int64, device=ids.device)
                    dist.all_gather_into_tensor(cnts, cnt, group=self.group)
                    tmax = int(cnts.max().item())
                    if ids.numel() < tmax:
                        ids = torch.cat([ids, ids.new_full((tmax - ids.numel(),), -1)])
                        dy = torch.cat([dy, dy.new_zeros((tmax - dy.shape[0], dy.shape[1]))])
                    ids_all = torch.empty(self.world * tmax, dtype=ids.dtype, device=ids.device)
                    dy_all = torch.empty((self.world * tmax, dy.shape[1]), dtype=dy.dtype, device=dy.device)
                    dist.all_gather_into_tensor(ids_all, ids.contiguous(), group=self.group)
                    dist.all_gather_into_tensor(dy_all, dy.contiguous(), group=self.group)
                else:
                    ids_all, dy_all = ids, dy
                buf = p._mi_grad_buf
                if p.grad is None:          # untied table: nothing else wrote its slot in this pass
                    buf.zero_()
                    p.grad = buf
                elif p.grad.data_ptr() != buf.data_ptr():
                    buf.copy_(p.grad)
                    p.grad = buf
                # tied to lm_head: the averaged wgrad is already there; either way the rows are added in place
                add_embedding_rows_(buf, dy_all, ids_all, 1.0 / self.world, emb.padding_idx)

    # ------------------------------------------------------------------------------------------------ backward side
    def _on_grad(self, p: torch.nn.Parameter):
        buf = p._mi_grad_buf
        g = p.grad
        if g is not None and g.data_ptr() != buf.data_ptr():
            # produced outside the arena (embedding, norms, modules that are not ours): move it in and alias
            buf.copy_(g)
            p.grad = buf
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        b = self._bucket_of[id(p)]
        b.pending -= 1
        if b.pending == 0 and self._sync:
            self._launch(b)

    def _launch(self, b: _Bucket):
        if self.world == 1 and not _FORCE_COLLECTIVES:
            return
        flat = self.arenas[b.arena][b.start:b.end]
        if self._avg_in_collective:
            try:
                self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), flat, False))
                return
            except (RuntimeError, ValueError):  # a backend build without AVG for this dtype: SUM, then scale (as with gloo)
                self._avg_in_collective = False
        self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat, True))

    def _finalize(self):
        """End of a backward pass (autograd engine callback): reduce what is still open, then make the compute stream wait
        for the collectives -- stream-side, the host does not block."""
        try:
            if self._sync:
                for b in self.buckets:
                    if 0 < b.pending < len(b.params):  # same graph on every rank, so every rank closes the same buckets
                        self._close_partial(b)
                for w, flat, summed in self._works:
                    w.wait()
                    if summed:
                        flat.mul_(1.0 / self.world)
                if self._sparse:
                    self._reduce_sparse()
        finally:
            self._works.clear()
            self._callback_queued = False
            for b in self.buckets:
                b.pending = len(b.params)

    def _close_partial(self, b: _Bucket):
        """A bucket some of whose parameters got no gradient in this pass: they contribute zeros on this rank."""
        if b.pending == 0:
            return
        with torch.no_grad():
            for p in b.params:
                if p.grad is None:
                    p._mi_grad_buf.zero_()
                    p.grad = p._mi_grad_buf
        b.pending = 0
        self._launch(b)

    # ------------------------------------------------------------------------------------------------ module surface
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside accumulate into the arena without reducing (DDP.no_sync)."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def describe(self) -> dict:
        return {"arenas": [{"dtype": str(a.dtype), "bytes": a.numel() * a.element_size()} for a in self.arenas],
                "buckets": [(b.end - b.start) * self.arenas[b.arena].element_size() for b in self.buckets],
                "world": self.world}


class _ShardHandle:
    """What a module sees of a row-sharded weight (`Parameter._mi_sharded`): the wrapper, this rank's shard, its rows."""
    __slots__ = ("dp", "shard", "r0", "rows")

    def __init__(self, dp, shard, r0, rows):
        self.dp, self.shard, self.r0, self.rows = dp, shard, r0, rows


class ShardedFP8DP(GradArenaDP):
    """The FSDP FULL_SHARD counterpart with an FP8 all-gather (`--sharding_mode fsdp_fp8`).

    torch's FSDP, as the reference wraps it (train_multi_gpu.py:392-406, :414-445), keeps parameters, gradients and optimiser
    state at 1/world per rank, all-gathers every layer's bf16 flat parameter in the forward AND again in the backward and
    reduce-scatters bf16 gradients: 6 bytes per parameter and step over xGMI, plus a cast of every weight after every gather.
    The FP8 Linear needs none of the bf16 weights in its backward -- it saved w8T -- so here:

      * every GEMM weight is ROW-sharded over the ranks ([N, K] cut into `world` equal row blocks).  A rank holds the bf16
        MASTER ROWS it owns and nothing else of the master: the module's Parameter keeps its logical shape (an expanded view of
        one element, no storage) and carries `_mi_sharded`; AdamW moments and the persistent gradient exist for the shard only;
      * backward: the wgrad GEMM writes the operand's full dW into a TRANSIENT buffer; when autograd hands it over it is
        reduce-scattered (AVG) into the shard's gradient, several weights per NCCL group call (`bucket_mb`), and freed when the
        collective has completed -- at most two buckets of full-size gradients exist at a time.  Every other parameter (norms,
        biases, embedding tables) is small or needed in bf16 everywhere and goes through the bucketed all-reduce of the base class;
      * optimiser: ClippedAdamW on the shards; mi_adamw_cast_bf16_multi / mi_adamw_mxcast_bf16_multi quantise the updated rows on
        the fly into the rank's rows of the operand's FP8 copies (module.WeightSink / MXWeightSink);
      * after the step the FP8 rows (+ E8M0 scales for MXFP8: both orientations) are all-gathered ASYNCHRONOUSLY on RCCL's stream,
        operand by operand in next-forward order, one NCCL group call per module; the forward of a module waits (stream-side) for
        ITS operands only and rebuilds w8T locally (mi_transpose_u8), so layer 0 computes while layer 27 is still gathering.
        No gather in the backward, no cast after a gather: 3 bytes per parameter and step instead of 6 (delayed scaling);
      * whenever a sink is not current -- first step, a scale-arena generation bump (per-layer autocast without an outer one),
        a loaded checkpoint, an evaluation pass first -- the module asks `refresh_operand`: the local rows are quantised with the
        CURRENT scale and gathered, which is what the replicated run's forward cast computes.  There is no stale-master path.

    What stays at full size per rank: the FP8 operand copies the GEMMs read (w8 + w8T, 2 bytes per parameter; for 8B: 16 GB of
    288).  `describe()` reports the bytes per category; tests/fsdp_fp8_worker.py asserts masters + gradients + moments = 1/world.
    amax: a rank deposits the amax of ITS rows; the arenas' MAX all-reduce makes it global before the next scale update, so the
    recipe must have reduce_amax=True (checked).  `gather_master_weights()` materialises full bf16 masters (checkpoint,
    FP8-off evaluation), `reshard()` drops them again.  Unmeasured on multi-GPU hardware (the driver alone runs N > 1);
    rehearsed with 2 ranks sharing one GPU over gloo and at world size 1 over RCCL (tests/test_distributed_gpu.py)."""

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_mb: float = 256.0, broadcast: bool = True,
                 sparse_embedding_grads: bool = True):
        self._rs_bucket_bytes = int(bucket_mb * (1 << 20))
        super().__init__(module, process_group, bucket_mb, broadcast, sparse_embedding_grads)
        self._shard_now()

    # ------------------------------------------------------------------------------------------------ construction
    def _prepare_sharding(self):
        from .pytorch.module import Linear, LayerNormLinear, LayerNormMLP
        self.rank = dist.get_rank(self.group)
        self._rs_pending: list = []     # (shard parameter, full gradient) of the open reduce-scatter bucket
        self._rs_pending_bytes = 0
        self._rs_inflight: list = []    # (works, [tensors kept alive], [(shard parameter, full gradient or None)])
        self._gathers: Dict[int, tuple] = {}  # id(sink) -> (works, post-gather fix-ups, stamp as of quantisation)
        self._materialised = False
        tables = {id(m.weight) for m in self.module.modules() if isinstance(m, torch.nn.Embedding)}
        self._sharded: Dict[int, torch.nn.Parameter] = {}
        self._fp8_modules = []
        for m in self.module.modules():
            if isinstance(m, (Linear, LayerNormLinear, LayerNormMLP)):
                self._fp8_modules.append(m)
                for p in m._parameters.values():
                    if (p is not None and p.requires_grad and p.dim() == 2 and id(p) not in tables and p.dtype == torch.bfloat16
                            and p.is_contiguous() and p.shape[0] % (32 * self.world) == 0 and p.shape[1] % 32 == 0):
                        self._sharded[id(p)] = p
        self._shards: Dict[int, torch.nn.Parameter] = {}

    def _build(self, groups, bucket_bytes: int):
        # the gradient arena holds the REPLICATED parameters only: a sharded weight's gradient exists at shard size
        rest = [[p for p in g if id(p) not in self._sharded] for g in groups]
        super()._build([g for g in rest if g], bucket_bytes)
        for p in self._sharded.values():
            p.register_post_accumulate_grad_hook(self._on_sharded_grad)

    def _shard_now(self):
        """Cut every sharded weight: keep this rank's rows in a Parameter of its own, release the full master."""
        with torch.no_grad():
            for p in self._sharded.values():
                n = p.shape[0] // self.world
                r0 = self.rank * n
                sp = torch.nn.Parameter(p.data[r0:r0 + n].clone(), requires_grad=True)
                sp._mi_shard_grad = torch.zeros_like(sp)  # the reduce-scatter's output; becomes `.grad` when it has landed
                sp._mi_shard_of = (p, r0, n)
                self._shards[id(p)] = sp
                p._mi_sharded = _ShardHandle(self, sp, r0, n)
                p._mi_full_shape = tuple(p.shape)
                p.data = torch.zeros(1, dtype=p.dtype, device=p.device).expand(p.shape)  # logical shape, no storage

    # ------------------------------------------------------------------------------------------------ backward side
    def wgrad_buffer(self, weights, K: int) -> Optional[torch.Tensor]:
        """module._wgrad_out: destination of one operand's weight-gradient GEMM -- a transient [sum N_i, K] bf16 buffer (the
        caching allocator hands the blocks of finished reduce-scatters back)."""
        if any(w.grad is not None for w in weights) or any(w.dtype != torch.bfloat16 or w.shape[1] != K for w in weights):
            return None
        return torch.empty((sum(w.shape[0] for w in weights), K), dtype=torch.bfloat16, device=self._shards[id(weights[0])].device)

    def _on_sharded_grad(self, p: torch.nn.Parameter):
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        if not self._sync:
            return  # gradient accumulation: the full-size gradient stays in `.grad` until the synchronising pass (as FSDP.no_sync)
        g = p.grad
        if not g.is_contiguous():
            g = g.contiguous()
        sp = self._shards[id(p)]
        p.grad = None
        self._rs_pending.append((sp, g))
        self._rs_pending_bytes += g.numel() * g.element_size()
        if self._rs_pending_bytes >= self._rs_bucket_bytes:
            self._flush_reduce_scatters()

    def _flush_reduce_scatters(self):
        """One NCCL group call (or, on gloo / at world 1, the per-weight fallback) for the pending (shard, full gradient) pairs."""
        pend, self._rs_pending, self._rs_pending_bytes = self._rs_pending, [], 0
        if not pend:
            return
        # at most two buckets of transient full-size gradients: settle the bucket before the previous one
        while len(self._rs_inflight) >= 2:
            self._settle(self._rs_inflight.pop(0))
        if self.world == 1 and not _FORCE_COLLECTIVES:
            for sp, g in pend:
                sp._mi_shard_grad.copy_(g)
            self._rs_inflight.append((None, [], [(sp, None) for sp, _ in pend]))
            return
        if self._avg_in_collective:  # RCCL: reduce-scatter with the average taken in the collective, one group call per bucket
            works = None
            try:
                from torch.distributed.distributed_c10d import _coalescing_manager
                with _coalescing_manager(group=self.group, device=pend[0][1].device, async_ops=True) as cm:
                    for sp, g in pend:
                        dist.reduce_scatter_tensor(sp._mi_shard_grad, g, op=dist.ReduceOp.AVG, group=self.group)
                works = [cm]
            except (ImportError, RuntimeError, TypeError, AttributeError):
                works = [dist.reduce_scatter_tensor(sp._mi_shard_grad, g, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                         for sp, g in pend]
            self._rs_inflight.append((works, [g for _, g in pend], [(sp, None) for sp, _ in pend]))
            return
        # gloo rehearsal (no reduce-scatter on this backend): all-reduce, then keep the rank's rows x 1 / world
        works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for _, g in pend]
        self._rs_inflight.append((works, [], [(sp, g) for sp, g in pend]))

    def _settle(self, entry):
        works, _keep, items = entry
        for w in (works or []):
            w.wait()
        for sp, full in items:
            if full is not None:
                _, r0, n = sp._mi_shard_of
                sp._mi_shard_grad.copy_(full[r0:r0 + n]).mul_(1.0 / self.world)
            sp.grad = sp._mi_shard_grad

    def _finalize(self):
        try:
            if self._sync:
                # weights whose full gradient was accumulated under no_sync and got no new one in this pass cannot occur: the
                # synchronising pass produces every gradient again; whatever is pending goes out now
                for p in self._sharded.values():
                    if p.grad is not None:  # accumulated under no_sync and completed by this pass
                        g = p.grad
                        p.grad = None
                        self._rs_pending.append((self._shards[id(p)], g.contiguous()))
                self._flush_reduce_scatters()
                while self._rs_inflight:
                    self._settle(self._rs_inflight.pop(0))
        finally:
            super()._finalize()

    # ------------------------------------------------------------------------------------------------ optimiser side
    def optimizer_param_groups(self):
        """Two groups for optim.ClippedAdamW: replicated parameters (every rank updates them identically) and the row shards
        (`sharded=True`: their squared gradient norm is summed over the ranks of `dp_group` before the clip coefficient is formed)."""
        seen, rep_u = set(), []
        for p in self.module.parameters():
            if p.requires_grad and id(p) not in self._sharded and id(p) not in seen:
                seen.add(id(p))
                rep_u.append(p)
        return [{"params": rep_u}, {"params": list(self._shards.values()), "sharded": True, "dp_group": self.group}]

    def _check_recipe(self, sink):
        arena = getattr(sink, "arena", None)
        rec = getattr(arena, "recipe", None)
        if self.world > 1 and rec is not None and not getattr(rec, "reduce_amax", True):
            raise RuntimeError("ShardedFP8DP needs reduce_amax=True: every rank quantises ITS rows of a weight, so the ranks must "
                               "derive the same scale from the global amax")

    def _sinks_in_forward_order(self):
        """(module, [sinks]) in registration order ~ the order the next forward uses them."""
        out = []
        for m in self._fp8_modules:
            ss = [v for k, v in m._wcache.items() if isinstance(k, tuple) and k[0] in ("sink", "mxsink")
                  and all(id(w) in self._sharded for w, _, _ in v.parts)]
            if ss:
                out.append((m, ss))
        return out

    def _gather_ops(self, sink):
        """The all-gathers of one operand as (dst, src) pairs plus the fix-ups to run once they have landed."""
        ops_, fix = [], []
        mx = hasattr(sink, "sc")
        K = sink.w8.shape[1]
        for w, row_off, n in sink.parts:
            rows = n // self.world
            lo = row_off + self.rank * rows
            dst = sink.w8[row_off:row_off + n]
            ops_.append((dst.view(-1), dst[self.rank * rows:(self.rank + 1) * rows].clone().view(-1)))
            if mx:
                d2 = sink.sct[row_off // 32:(row_off + n) // 32]
                ops_.append((d2.view(-1), sink.sct[lo // 32:(lo + rows) // 32].clone().view(-1)))
                # column blocks of [K/32, N] and [K, N]: gathered rank-major into scratch, then laid out
                t_sc = torch.empty((self.world, sink.sc.shape[0], rows), dtype=torch.uint8, device=sink.sc.device)
                t_wt = torch.empty((self.world, K, rows), dtype=torch.uint8, device=sink.wt8.device)
                ops_.append((t_sc.view(-1), sink.sc[:, lo:lo + rows].contiguous().view(-1)))
                ops_.append((t_wt.view(-1), sink.wt8[:, lo:lo + rows].contiguous().view(-1)))
                fix.append((sink.sc[:, row_off:row_off + n], t_sc))
                fix.append((sink.wt8[:, row_off:row_off + n], t_wt))
        return ops_, fix

    def _launch_gather(self, sinks, blocking: bool):
        pairs, per_sink = [], []
        for sink in sinks:
            self._check_recipe(sink)
            o, f = self._gather_ops(sink)
            pairs += o
            per_sink.append((sink, f))
        works = []
        if self.world > 1 or _FORCE_COLLECTIVES:
            try:
                if blocking or not self._avg_in_collective:
                    raise TypeError
                from torch.distributed.distributed_c10d import _coalescing_manager
                with _coalescing_manager(group=self.group, device=pairs[0][0].device, async_ops=True) as cm:
                    for dst, src in pairs:
                        dist.all_gather_into_tensor(dst, src, group=self.group)
                works = [cm]
            except (ImportError, RuntimeError, TypeError, AttributeError):
                works = [dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True) for dst, src in pairs]
        for sink, f in per_sink:
            # the stamp the bytes in flight deserve: parameter versions and scale-arena generation AS OF THEIR QUANTISATION (now).
            # Stamping at wait time would declare them current under a scale that has moved on since (per-layer autocasts
            # without an outer one bump the generation between this launch and the forward that waits)
            sink.mark()
            stamp, sink.stamp = sink.stamp, None
            self._gathers[id(sink)] = (works, f, stamp)
        if blocking:
            for sink, _ in per_sink:
                self.wait_operand(sink)

    def wait_operand(self, sink):
        """Called by the module before it reads the sink: wait (stream-side) for the operand's all-gather, finish it locally."""
        ent = self._gathers.pop(id(sink), None)
        if ent is None:
            return
        from .pytorch import ops
        works, fix, stamp = ent
        for w in works:
            w.wait()
        for dst, tmp in fix:  # [world, R, rows] -> [R, world * rows]
            dst.view(dst.shape[0], self.world, tmp.shape[2]).copy_(tmp.permute(1, 0, 2))
        if hasattr(sink, "w8t"):
            ops.transpose_u8(sink.w8, out=sink.w8t)
        sink.stamp = stamp

    def after_optimizer_step(self):
        """Issue the FP8 all-gathers of the freshly quantised shards: asynchronously, module by module in next-forward order."""
        for _m, sinks in self._sinks_in_forward_order():
            self._launch_gather(sinks, blocking=False)

    def refresh_operand(self, sink, fmt: int):
        """Delayed scaling, sink not current: cast this rank's rows with the CURRENT scale (amax deposited as the forward cast
        would), gather, transpose.  Collective: every rank reaches it at the same point of its forward."""
        from .pytorch import ops
        for w, row_off, n in sink.parts:
            h = w._mi_sharded
            ops.cast_amax(h.shard.data, sink.scale, sink.amax, fmt, y=sink.w8[row_off + h.r0:row_off + h.r0 + h.rows], want_t=False)
        self._launch_gather([sink], blocking=True)

    def refresh_mx_operand(self, sink, fmt: int):
        from .pytorch import ops
        for w, row_off, n in sink.parts:
            h = w._mi_sharded
            lo = row_off + h.r0
            ops.mxfp8_quantize(h.shard.data, fmt, rowwise=True, colwise=True,
                               out=(sink.w8[lo:lo + h.rows], sink.sc[:, lo:lo + h.rows], sink.wt8[:, lo:lo + h.rows],
                                    sink.sct[lo // 32:(lo + h.rows) // 32]))
        self._launch_gather([sink], blocking=True)

    # ------------------------------------------------------------------------------------------------ full masters on request
    def gather_master_weights(self):
        """Materialise the full bf16 master of every sharded weight on every rank (checkpoint, evaluation with FP8 off).
        Collective.  `reshard()` releases them; training may continue either way (the shards stay the source of truth)."""
        with torch.no_grad():
            for p in self._sharded.values():
                sp = self._shards[id(p)]
                full = torch.empty(p._mi_full_shape, dtype=sp.dtype, device=sp.device)
                if self.world > 1 or _FORCE_COLLECTIVES:
                    dist.all_gather_into_tensor(full.view(-1), sp.data.contiguous().view(-1), group=self.group)
                else:
                    full.copy_(sp.data)
                p.data = full
        self._materialised = True

    def reshard(self):
        with torch.no_grad():
            for p in self._sharded.values():
                p.data = torch.zeros(1, dtype=p.dtype, device=p.device).expand(p._mi_full_shape)
        self._materialised = False

    def describe(self) -> dict:
        d = super().describe()
        nb = lambda t: t.numel() * t.element_size()
        shards = list(self._shards.values())
        d["sharded_weights"] = len(self._sharded)
        d["sharded_logical_bytes"] = sum(int(torch.tensor(p._mi_full_shape).prod()) * 2 for p in self._sharded.values())
        d["master_bytes"] = sum(nb(sp.data) for sp in shards) + (d["sharded_logical_bytes"] if self._materialised else 0)
        d["shard_grad_bytes"] = sum(nb(sp._mi_shard_grad) for sp in shards)
        d["module_param_storage_bytes"] = sum(p.untyped_storage().nbytes() for p in self._sharded.values())
        fp8 = 0
        for _m, sinks in self._sinks_in_forward_order():
            for s_ in sinks:
                fp8 += sum(nb(getattr(s_, a)) for a in ("w8", "w8t", "wt8", "sc", "sct") if hasattr(s_, a))
        d["fp8_operand_bytes"] = fp8
        return d


# debug / rehearsal: run the collectives even at world size 1 (exercises the RCCL stream hand-over on a one-GPU box)
_FORCE_COLLECTIVES = os.environ.get("LLM_FP8_AMD_FORCE_COLLECTIVES") == "1"


def add_embedding_rows_(grad: torch.Tensor, dy: torch.Tensor, ids: torch.Tensor, alpha: float, padding_idx) -> None:
    """grad [V, H] += alpha * (rows of dy scattered by ids), in place and deterministic.  On the GPU in bf16 this is
    mi_embedding_grad_add (touched rows only); otherwise aten's embedding_dense_backward + a dense add."""
    pad = -1 if padding_idx is None else int(padding_idx)
    if grad.is_cuda and grad.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and grad.is_contiguous() and grad.shape[1] % 8 == 0:
        from .pytorch import ops
        ops.embedding_grad_add_(grad, dy, ids, alpha, pad)
        return
    ids = ids.reshape(-1)
    ids = torch.where(ids < 0, torch.zeros_like(ids), ids)  # padding entries (-1) carry zero rows: route them to row 0
    dense = torch.ops.aten.embedding_dense_backward(dy.reshape(-1, dy.shape[-1]).to(grad.dtype).contiguous(), ids,
                                                    grad.shape[0], pad, False)
    grad.add_(dense, alpha=alpha)


class _LocalEmbeddingGrad:
    """Single-process counterpart of GradArenaDP's row-sparse embedding reduction: the embedding's backward adds its rows
    straight into the table's `.grad` (which, for a table tied to lm_head, already holds the lm_head wgrad by then) instead of
    materialising a dense [vocab, hidden] gradient that autograd then adds to the other one (Llama-3.2-3B: 788 MB each)."""

    def __init__(self):
        self._anchor = None

    def forward(self, emb, ids):
        if not (torch.is_grad_enabled() and emb.weight.requires_grad):
            return F.embedding(ids, emb.weight, emb.padding_idx)
        if self._anchor is None or self._anchor.device != emb.weight.device:
            self._anchor = torch.zeros(1, device=emb.weight.device, requires_grad=True)
        return _DeferredEmbeddingGrad.apply(self._anchor, ids, emb.weight.detach(), self, emb)

    def _defer_embedding_grad(self, emb, ids, dy):
        p = emb.weight
        with torch.no_grad():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            add_embedding_rows_(p.grad, dy.reshape(-1, dy.shape[-1]), ids.reshape(-1), 1.0, emb.padding_idx)


def install_local_embedding_grad(module: torch.nn.Module) -> int:
    """For a model trained WITHOUT a data-parallel wrapper (wrappers reduce `.grad` from hooks that fire before the embedding's
    backward has run): route every plain nn.Embedding's weight gradient through _LocalEmbeddingGrad.  Returns how many."""
    sink, n = _LocalEmbeddingGrad(), 0
    for m in module.modules():
        if (type(m) is torch.nn.Embedding and m.weight.requires_grad and m.weight.is_cuda and m.max_norm is None and not m.sparse
                and not m.scale_grad_by_freq and "forward" not in m.__dict__):
            m.forward = functools.partial(sink.forward, m)
            n += 1
    return n


def replicated_state_bytes(module: torch.nn.Module, moment_bytes: int = 4) -> int:
    """Weights + gradients + two AdamW moments per trainable parameter (moments counted at fp32 to be safe)."""
    n = sum(p.numel() * (2 * p.element_size() + 2 * moment_bytes) for p in module.parameters() if p.requires_grad)
    return n + sum(p.numel() * p.element_size() for p in module.parameters() if not p.requires_grad)


def fits_replicated(module: torch.nn.Module, device: torch.device, fraction: float = 0.5) -> bool:
    """True when the replicated training state takes at most `fraction` of the device's memory (the rest is for
    activations): 3B -> 39 GB, 8B -> 96 GB of 288 GB."""
    if device.type != "cuda":
        return True
    total = torch.cuda.get_device_properties(device).total_memory
    return replicated_state_bytes(module) <= fraction * total
