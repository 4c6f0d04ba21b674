"""FP8 modules with the `transformer_engine.pytorch` surface the reference consumes:
`Linear` (accelerate utils/transformer_engine.py:52-59), `LayerNormLinear` / `LayerNormMLP`
(te_llama.py:45-63 via MultiheadAttention / LayerNormMLP), same parameter names as
`replace_params` writes (te_llama.py:194-238).

Every GEMM site is ONE autograd Function (`_FP8LinearFn`) whose forward/backward run only HIP kernels
from libmi_fp8.so:

  fwd:  x  --cast+T+amax-->  x8, x8T     w  --cast+T+amax-->  w8, w8T      y  = gemm(x8, w8)  (+bias)
  bwd:  dy --cast+T+amax-->  g8, g8T     dx = gemm(g8, w8T)                dw = gemm(g8T, x8T)

(SURVEY.md 3.4).  Weights stay bf16 `nn.Parameter`s; nothing here falls back to PyTorch matmuls when
FP8 is enabled.  Outside `fp8_autocast` (or with enabled=False) the modules are plain bf16 layers.
"""
from __future__ import annotations

import io
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from ..common.recipe import DelayedScaling, Format, MXFP8BlockScaling, Recipe, fmt_codes
from . import ops
from .fp8 import FP8GlobalStateManager, ModuleMeta

__all__ = ["Linear", "LayerNormLinear", "LayerNormMLP", "LayerNorm", "RMSNorm"]


def _as_bf16_2d(t: torch.Tensor) -> torch.Tensor:
    t2 = t.reshape(-1, t.shape[-1])
    if t2.dtype != torch.bfloat16:
        t2 = t2.to(torch.bfloat16)
    return t2.contiguous()


class DyHandoff:
    """grad_output of an FP8 Linear delivered already quantised by the op that produces it.

    The Linear's forward `offer`s what its backward will quantise grad_output with (delayed scaling: scale and amax slot are
    known in advance); the producer's backward (attention._RoPESplitFn) then emits the FP8 copies itself, `put`s them here and
    returns an UNWRITTEN bf16 placeholder as the autograd gradient; the Linear's backward `take`s the copies and never reads
    the placeholder.  One object per forward call, private to the module that wires the two together: nothing else may sit
    between them in the graph.  `take` checks that the gradient it was handed is that placeholder and fails loudly otherwise.
    Guard: a producer asks `handoff_readers` first -- with a tensor hook or `retain_grad()` on the activation in between, or
    under anomaly mode, it writes the real gradient and the hand-off is skipped for that backward pass."""
    __slots__ = ("scale", "amax", "fmt", "want_y", "want_t", "fp8", "ptr", "mx", "act_ref")

    def __init__(self):
        self.scale = self.amax = self.fmt = self.fp8 = self.ptr = None
        self.act_ref = None  # weakref to the activation the user can reach (the logits): see handoff_readers
        self.want_y = self.want_t = self.mx = False

    def offer(self, scale, amax, fmt, want_y, want_t, mx: bool = False):
        """Delayed scaling: (scale, amax slot, format); MXFP8 (`mx`): the format only, the producer `put`s the four tensors of
        ops.mxfp8_quantize (rowwise data + scales, columnwise data + scales) instead of (y, yT)."""
        self.scale, self.amax, self.fmt, self.want_y, self.want_t, self.mx = scale, amax, fmt, want_y, want_t, mx

    def withdraw(self):
        self.scale, self.mx = None, False

    def offered(self) -> bool:
        return (self.scale is not None or self.mx) and (self.want_y or self.want_t)

    def put(self, fp8: tuple, placeholder: torch.Tensor):
        self.fp8, self.ptr = fp8, placeholder.untyped_storage().data_ptr()

    def take(self, dy: torch.Tensor):
        if dy.untyped_storage().data_ptr() != self.ptr:
            raise RuntimeError("DyHandoff: the gradient reaching the Linear is not the producer's placeholder "
                               "(an op was inserted between the q|k|v projection and the rotary split)")
        fp8, self.fp8, self.ptr = self.fp8, None, None
        return fp8


def handoff_readers(t) -> bool:
    """True when something may READ the bf16 gradient that flows from a DyHandoff producer to its Linear: a tensor hook or
    `retain_grad()` on the activation between them (`t`, the producer's input), or autograd's anomaly mode (it scans every
    gradient for NaNs).  The producer then takes its ordinary route -- it writes the real gradient, `put`s nothing, and the
    Linear quantises what it receives -- instead of handing an unwritten placeholder to code that would read garbage."""
    if torch.is_anomaly_enabled():
        return True
    if t is None:
        return False
    return bool(getattr(t, "retains_grad", False)) or bool(getattr(t, "_backward_hooks", None))


class _GemmSpec:
    """Everything non-tensor a GEMM site needs: recipe, meta windows, slot base, update trigger."""
    __slots__ = ("recipe", "meta_fwd", "meta_bwd", "g", "fmt_fwd", "fmt_bwd", "trigger_bwd_update", "training", "eps",
                 "wcache", "first_mb", "with_skip", "rstd", "dy_handoff", "defer_bias")

    def __init__(self, recipe, meta_fwd, meta_bwd, g, trigger_bwd_update, training, eps=1e-5, wcache=None, first_mb=None,
                 with_skip=False, rstd=None, dy_handoff=None, defer_bias=False):
        self.dy_handoff = dy_handoff
        # LayerNormMLP (delayed scaling, fused SwiGLU): the fc1 bias is added inside the SwiGLU kernels (always, when there is
        # one) and -- defer_bias -- the fc2 bias is left to the caller's residual add (residual_add_stats(..., bias=...)): TE's
        # bias + activation fusion.  The bias add costs 12 % of a K = 3072 GEMM in its epilogue and nothing in an HBM-bound kernel.
        self.defer_bias = defer_bias
        self.eps = eps
        # rstd: RMSNorm statistics of the input already computed by the producer of the input (residual_add_stats)
        self.rstd = rstd
        # with_skip: the Function also returns its input as a second output (the residual branch); the gradient arriving
        # over it is added inside the RMSNorm-backward kernel instead of by a separate autograd add
        self.with_skip = with_skip
        # FP8 weight caching across micro-batches (TE `is_first_microbatch`): None = cast every
        # forward (what the reference does), True = cast and keep, False = reuse the kept FP8 weights and their scale_inv
        self.wcache, self.first_mb = wcache, first_mb
        self.recipe, self.meta_fwd, self.meta_bwd, self.g = recipe, meta_fwd, meta_bwd, g
        self.fmt_fwd, self.fmt_bwd = fmt_codes(recipe.fp8_format)
        self.trigger_bwd_update = trigger_bwd_update
        self.training = training


class WeightSink:
    """FP8 copies (w8 [N,K], w8T [K,N]) of one GEMM's weight operand that the OPTIMISER keeps current (optim.ClippedAdamW ->
    mi_adamw_cast_bf16_multi): under delayed scaling the scale of the next forward's weight cast is final once this step's
    forward has ended, and the optimiser streams every weight anyway, so it emits the bytes (and the amax) that forward
    would have produced -- bitwise -- and the forward skips its weight casts (141 launches and one read of all weights per
    step on Llama-3.2-3B).  `stamp` = (versions of the parts, generation of the scale arena) at emission; the copies are used
    only while both still match (no other write to the weights, no scale update in between: an evaluation pass or a second
    micro-batch re-casts as before)."""
    __slots__ = ("w8", "w8t", "parts", "arena", "scale", "amax", "stamp")

    def __init__(self, weights, ns, N, K, dev, mf, slot):
        self.w8 = torch.empty((N, K), dtype=torch.uint8, device=dev)
        self.w8t = torch.empty((K, N), dtype=torch.uint8, device=dev)
        self.parts, r = [], 0
        for w, n in zip(weights, ns):
            self.parts.append((w, r, n))
            w._mi_fp8_sink = (self, r, n)
            r += n
        self.arena, self.scale, self.amax = mf.arena, mf.scale(slot), mf.amax(slot)
        self.stamp = None

    def fresh(self) -> bool:
        return self.stamp is not None and self.stamp == (tuple(w._version for w, _, _ in self.parts), self.arena.generation)

    def mark(self) -> None:  # called by the optimiser after it rewrote every part in this step
        self.stamp = (tuple(w._version for w, _, _ in self.parts), self.arena.generation)


# LLM_FP8_AMD_NO_MLP_BIAS_FUSION=1: keep both MLP biases in the GEMM epilogues (the round-2 behaviour; A/B switch)
_FUSE_MLP_BIAS = os.environ.get("LLM_FP8_AMD_NO_MLP_BIAS_FUSION") != "1"


def weight_sinks_enabled() -> bool:
    return os.environ.get("LLM_FP8_AMD_NO_OPT_WCAST") != "1"


def sharded_handle(weights):
    """distributed.ShardedFP8DP marks a weight whose bf16 master rows live 1/world per rank with `_mi_sharded` (the Parameter the
    module holds keeps its logical shape but no storage).  Such an operand has ONE source of FP8 bytes: its sink, filled by the
    optimiser's rows + an FP8 all-gather -- or, whenever the sink is not current (first step, scale arena moved on, checkpoint
    load, evaluation first), by `handle.dp.refresh_*`: quantise the local rows now and gather.  Returns the handle or None; a
    mix of sharded and replicated parts in one operand is refused."""
    hs = [getattr(w, "_mi_sharded", None) for w in weights]
    if all(h is None for h in hs):
        return None
    if any(h is None for h in hs):
        raise RuntimeError("an FP8 GEMM operand mixes row-sharded and replicated weight parts (distributed.ShardedFP8DP shards all "
                           "parts of an operand or none)")
    return hs[0]


def _master(w):
    """The bf16 master of a weight for the unquantised path (FP8 disabled): a row-sharded weight has none on this rank."""
    if getattr(w, "_mi_sharded", None) is not None and not w.is_contiguous():
        raise RuntimeError("this weight is row-sharded (distributed.ShardedFP8DP): its bf16 master exists 1/world per rank, so the "
                           "unquantised path cannot run -- keep FP8 enabled, or materialise the masters first with "
                           "dp.gather_master_weights() (and dp.reshard() afterwards)")
    return w


def _weight_ok_for_sink(w, K: int) -> bool:
    return (isinstance(w, torch.nn.Parameter) and w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] == K
            and (w.is_contiguous() or getattr(w, "_mi_sharded", None) is not None))


class MXWeightSink:
    """MXFP8 copies (row-wise w8 [N,K] + E8M0 [K/32,N]; column-wise, stored transposed, wt8 [K,N] + E8M0 [N/32,K]) of one GEMM's
    weight operand that the OPTIMISER keeps current (optim.ClippedAdamW -> mi_adamw_mxcast_bf16_multi).  Block scaling has no
    state, so the only condition for using them is that nobody wrote the weights since: `stamp` = versions of the parts."""
    __slots__ = ("w8", "sc", "wt8", "sct", "parts", "stamp")

    def __init__(self, weights, ns, N, K, dev):
        self.w8 = torch.empty((N, K), dtype=torch.uint8, device=dev)
        self.sc = torch.empty((K // 32, N), dtype=torch.uint8, device=dev)
        self.wt8 = torch.empty((K, N), dtype=torch.uint8, device=dev)
        self.sct = torch.empty((N // 32, K), dtype=torch.uint8, device=dev)
        self.parts, r = [], 0
        for w, n in zip(weights, ns):
            self.parts.append((w, r, n))
            w._mi_mx_sink = (self, r, n)
            r += n
        self.stamp = None

    def fresh(self) -> bool:
        return self.stamp is not None and self.stamp == tuple(w._version for w, _, _ in self.parts)

    def mark(self) -> None:  # called by the optimiser after it rewrote every part in this step
        self.stamp = tuple(w._version for w, _, _ in self.parts)


def _mx_sink_copies(spec: _GemmSpec, g: int, weights, ns, N: int, K: int, dev):
    """(w8, sc, wt8, sct) from the optimiser-maintained sink of GEMM `g` if they are current, else None (the caller quantises).
    A training pass creates the sink the first time round; the optimiser fills it at its next step.  Row-sharded weights
    (distributed.ShardedFP8DP) always go through the sink: a stale one is refreshed from the ranks' shards."""
    shard = sharded_handle(weights)
    if shard is None and (spec.wcache is None or spec.fmt_fwd != 0 or not weight_sinks_enabled()):
        return None
    if shard is not None and (spec.wcache is None or spec.fmt_fwd != 0):
        raise RuntimeError("row-sharded weights need the module's FP8 weight cache and an E4M3 forward format")
    sink = spec.wcache.get(("mxsink", g))
    stale = sink is None or len(sink.parts) != len(weights) or any(a is not b for (a, _, _), b in zip(sink.parts, weights))
    if stale:
        if not spec.training and shard is None:
            return None
        ok = (N % 32 == 0 and K % 32 == 0 and all(n % 32 == 0 for n in ns) and all(_weight_ok_for_sink(w, K) for w in weights))
        if not ok:
            if shard is not None:
                raise RuntimeError("row-sharded weights of this shape cannot take an MXFP8 sink")
            return None
        sink = spec.wcache[("mxsink", g)] = MXWeightSink(weights, ns, N, K, dev)
    if shard is not None:
        shard.dp.wait_operand(sink)            # an FP8 all-gather issued after the optimiser step may still be in flight
        if not sink.fresh():
            shard.dp.refresh_mx_operand(sink, spec.fmt_fwd)
    return (sink.w8, sink.sc, sink.wt8, sink.sct) if sink.fresh() else None


def _cast_weights(spec: _GemmSpec, g: int, weights, ns, N: int, K: int, dev, need_t: bool):
    """FP8 copies (w8 [N,K], w8T [K,N]) of the concatenated weight parts of GEMM `g` under delayed scaling, plus the
    scale_inv they were quantised with.  Honours the micro-batch cache of the spec (see _GemmSpec)."""
    mf = spec.meta_fwd
    ck = ("ds", g)
    if spec.first_mb is False and spec.wcache is not None:
        hit = spec.wcache.get(ck)
        if hit is not None and (hit[1] is not None or not need_t):
            return hit
    # (no grad-mode test here: inside an autograd Function's forward grad mode is always off; fresh copies are the bytes a
    # cast would produce now in any mode)
    shard = sharded_handle(weights)
    if shard is not None and (spec.wcache is None or spec.fmt_fwd != 0):
        raise RuntimeError("row-sharded weights need the module's FP8 weight cache and an E4M3 forward format")
    if spec.wcache is not None and spec.fmt_fwd == 0 and (weight_sinks_enabled() or shard is not None):
        sink = spec.wcache.get(("sink", g))
        stale = sink is None or sink.arena is not mf.arena or len(sink.parts) != len(weights) or any(a is not b for (a, _, _), b in zip(sink.parts, weights))
        if stale and not spec.training and shard is None:
            sink = None  # sinks are created by training passes only; an evaluation pass may USE a fresh one (same bytes)
        elif stale:
            if all(_weight_ok_for_sink(w, K) for w in weights) and N % 8 == 0 and K % 8 == 0:
                sink = spec.wcache[("sink", g)] = WeightSink(weights, ns, N, K, dev, mf, 3 * g + 1)
            else:
                sink = None
        if shard is not None:
            if sink is None:
                raise RuntimeError("row-sharded weights of this shape cannot take an FP8 sink")
            shard.dp.wait_operand(sink)        # an FP8 all-gather issued after the optimiser step may still be in flight
            if not sink.fresh():
                # the bytes in the sink were quantised with another scale generation (or never): the only bf16 source is the
                # ranks' shards -- cast the local rows with the CURRENT scale and gather (what a replicated run's forward cast does)
                shard.dp.refresh_operand(sink, spec.fmt_fwd)
        if sink is not None and sink.fresh():
            siw = mf.scale_inv_snapshot()[3 * g + 1:3 * g + 2]
            if spec.first_mb is True:
                spec.wcache[ck] = (sink.w8, sink.w8t, siw)
            return sink.w8, sink.w8t, siw
    keep = spec.first_mb is True and spec.wcache is not None
    want_t = need_t or keep
    w8 = torch.empty((N, K), dtype=torch.uint8, device=dev)
    w8t = torch.empty((K, N), dtype=torch.uint8, device=dev) if want_t else None
    r = 0
    for w, n in zip(weights, ns):
        wb = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        ops.cast_amax(wb.contiguous(), mf.scale(3 * g + 1), mf.amax(3 * g + 1), spec.fmt_fwd,
                      y=w8[r:r + n], yT=None if w8t is None else w8t[:, r:r + n], want_t=want_t)
        r += n
    siw = mf.scale_inv_snapshot()[3 * g + 1:3 * g + 2]  # the arena is rewritten at autocast exit; the snapshot is not
    if keep:
        spec.wcache[ck] = (w8, w8t, siw)
    return w8, w8t, siw


def _mx_quantize_weights(weights, ns, N: int, K: int, fmt: int, colwise: bool):
    """MXFP8 copies of the concatenated weight parts of one GEMM operand (query | key | value): every part is quantised
    straight into its row-block of the operand's buffers (mi_mxfp8_quantize_ex), no bf16 concatenation."""
    ws_ = [(w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)).contiguous() for w in weights]
    if len(ws_) == 1:
        return ops.mxfp8_quantize(ws_[0], fmt, rowwise=True, colwise=colwise)
    if any(n % 32 for n in ns):
        return ops.mxfp8_quantize(torch.cat(ws_, 0), fmt, rowwise=True, colwise=colwise)
    dev = ws_[0].device
    w8 = torch.empty((N, K), dtype=torch.uint8, device=dev)
    sc = torch.empty((K // 32, N), dtype=torch.uint8, device=dev)
    wt8 = torch.empty((K, N), dtype=torch.uint8, device=dev) if colwise else None
    sct = torch.empty((N // 32, K), dtype=torch.uint8, device=dev) if colwise else None
    r = 0
    for w, n in zip(ws_, ns):
        ops.mxfp8_quantize(w, fmt, rowwise=True, colwise=colwise,
                           out=(w8[r:r + n], sc[:, r:r + n], wt8[:, r:r + n] if colwise else None,
                                sct[r // 32:(r + n) // 32] if colwise else None))
        r += n
    return w8, sc, wt8, sct


class _AddStatsFn(torch.autograd.Function):
    """out = a + b (+ bias) and the RMSNorm statistics of `out` in one pass (mi_add_rmsnorm_stats / mi_add_bias_rmsnorm_stats).
    `bias`: a detached tensor -- the gradient of a deferred bias is produced by the module that owns it (column sums of dy)."""

    @staticmethod
    def forward(ctx, a, b, eps, bias=None):
        out, rstd = ops.add_rmsnorm_stats(a, b, eps, bias=bias)
        ctx.mark_non_differentiable(rstd)
        ctx.set_materialize_grads(False)  # autograd otherwise zero-fills a [tokens] fp32 gradient for rstd in every backward
        return out, rstd

    @staticmethod
    def backward(ctx, dout, _drstd):
        return dout, dout, None, None


def residual_add_stats(a: torch.Tensor, b: torch.Tensor, eps: float, bias: Optional[torch.Tensor] = None):
    """`a + b` for the residual stream, plus rstd (or None) for the RMSNorm that consumes the sum: (sum, rstd).
    `bias`: the deferred output bias of the module that produced `b` (LayerNormMLP(..., _defer_bias=True)): out = a + (b + bias)."""
    if (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape == b.shape and a.is_contiguous()
            and b.is_contiguous() and a.shape[-1] % 8 == 0 and (bias is None or a.shape[-1] == bias.numel())):
        bb = None if bias is None else bias.detach().to(torch.bfloat16).contiguous()
        return _AddStatsFn.apply(a, b, eps, bb)
    if bias is not None:
        b = b + bias.detach().to(b.dtype)
    return a + b, None


def _wgrad_out(weights, K: int) -> Optional[torch.Tensor]:
    """Destination of the weight-gradient GEMM inside the data-parallel gradient arena (distributed.GradArenaDP): the
    [sum N_i, K] block formed by the weights' slots when these are adjacent and in order, the parameters are bf16 and
    nothing has been accumulated into them yet (gradient accumulation adds into `.grad` instead).  None otherwise."""
    shard = getattr(weights[0], "_mi_sharded", None)
    if shard is not None:  # distributed.ShardedFP8DP: a transient [sum N_i, K] buffer that lives until its reduce-scatter has run
        return shard.dp.wgrad_buffer(weights, K)
    slot = getattr(weights[0], "_mi_grad_slot", None)
    if slot is None:
        return None
    arena, off = slot
    if arena.dtype != torch.bfloat16:
        return None
    end = off
    for w in weights:
        s = getattr(w, "_mi_grad_slot", None)
        if s is None or s[0] is not arena or s[1] != end or w.grad is not None or w.dtype != torch.bfloat16 or w.shape[1] != K:
            return None
        end += w.numel()
    return arena[off:end].view(-1, K)


def _dgrad_wgrad(g8, wt8, g8t, xt8, sig, si_w, si_x, fmt_b: int, fmt_f: int, dw_out, need_dgrad: bool, need_wgrad: bool):
    """A Linear's two backward GEMMs on one grad_output: dX [M, K] = G8 [M, N] . W8T [K, N]^T and dW [N, K] = G8T [N, M] . X8T [K, M]^T.
    ONE grouped persistent launch (ops.gemm_fp8_grouped) where that is faster (ops.grouped_gemm_choice: measured once per shape in
    a single-process run, the count model under torch.distributed; env LLM_FP8_AMD_GROUPED_GEMM = auto | plan | autotune | off) --
    one ramp, one exposed epilogue, and the short problem's tiles fill the idle part of the long one's last round -- else two
    launches.  Bitwise the
    same results either way.  (Under torch.distributed the GEMMs run one workgroup per tile so that RCCL's kernels get CUs:
    no persistent grouping there.)"""
    dx = dw = None
    if need_dgrad and need_wgrad and ops.default_gemm_algo() in (0, 4, 47) and os.environ.get("LLM_FP8_AMD_NO_GROUPED_GEMM") != "1":
        M, N = g8.shape
        K = wt8.shape[0]
        if (ops.grouped_gemm_ok(((M, K, N), (N, K, M))) and g8.stride(1) == 1 and wt8.stride(1) == 1 and g8t.stride(1) == 1
                and xt8.stride(1) == 1):
            dx = torch.empty((M, K), dtype=torch.bfloat16, device=g8.device)
            dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.bfloat16, device=g8.device)
            probs = [(g8, wt8, sig, si_w, dx), (g8t, xt8, sig, si_x, dw)]
            cfg = ops.grouped_gemm_choice(probs, fmt_b, fmt_f)  # cached per shape set: -1 = two launches are faster here
            if cfg >= 0:
                ops.gemm_fp8_grouped(probs, fmt_b, fmt_f, tile_cfg=cfg)
                return dx, dw
            dx = dw = None
    if need_dgrad:
        dx = ops.gemm_fp8(g8, wt8, sig, si_w, fmt_b, fmt_f)
    if need_wgrad:
        dw = ops.gemm_fp8(g8t, xt8, sig, si_x, fmt_b, fmt_f, out=dw_out)
    return dx, dw


def _skip_2d(dskip: Optional[torch.Tensor], like: torch.Tensor) -> Optional[torch.Tensor]:
    """Residual-branch gradient as a contiguous bf16 [tokens, features] matrix for mi_rmsnorm_bwd's `dres`."""
    if dskip is None:
        return None
    d = dskip.reshape(like.shape)
    return d if (d.dtype == torch.bfloat16 and d.is_contiguous()) else d.to(torch.bfloat16).contiguous()


class _FP8LinearFn(torch.autograd.Function):
    """y[M, sum N_i] = x[M,K] . cat(W_i)[N,K]^T (+ bias), FP8 operands, bf16 result."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, bias: Optional[torch.Tensor], spec: _GemmSpec, ln_w: Optional[torch.Tensor],
                *weights: torch.Tensor):
        """`ln_w` (with spec.eps): K9 -- x is the UN-normalised input and RMSNorm is fused into its FP8 cast."""
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        if M % 8 or K % 16:
            raise RuntimeError(f"FP8 Linear needs tokens % 8 == 0 and in_features % 16 == 0, got {M} x {K}")
        ns = [w.shape[0] for w in weights]
        N = sum(ns)
        dev = x2.device
        # (forward runs in no-grad mode; needs_input_grad is all-False when grad was disabled at apply time)
        need_dgrad = bool(ctx.needs_input_grad[0]) or bool(ctx.needs_input_grad[3])
        need_wgrad = any(ctx.needs_input_grad[4:])
        bias_bf16 = None if bias is None else bias.to(torch.bfloat16).contiguous()
        ctx.norm = None
        if spec.recipe.mxfp8():
            if ln_w is not None:
                gam = (ln_w if ln_w.dtype == torch.bfloat16 else ln_w.to(torch.bfloat16)).contiguous()
                rstd = spec.rstd if spec.rstd is not None else ops.rmsnorm_stats(x2, spec.eps)
                x8, xs, xt8, xts = ops.mxfp8_norm_quantize(x2, rstd, gam, spec.fmt_fwd, rowwise=True, colwise=need_wgrad)
                if need_dgrad:
                    ctx.norm = (x2, rstd, gam, ln_w.dtype)
            else:
                x8, xs, xt8, xts = ops.mxfp8_quantize(x2, spec.fmt_fwd, rowwise=True, colwise=need_wgrad)
            ck = ("mx", spec.g)
            hit = spec.wcache.get(ck) if (spec.first_mb is False and spec.wcache is not None) else None
            if hit is not None and (hit[2] is not None or not need_dgrad):
                w8, ws, wt8, wts = hit
            else:
                kept = _mx_sink_copies(spec, spec.g, weights, ns, N, K, x2.device)
                if kept is not None:
                    w8, ws, wt8, wts = kept
                else:
                    w8, ws, wt8, wts = _mx_quantize_weights(weights, ns, N, K, spec.fmt_fwd, need_dgrad or spec.first_mb is True)
                if spec.first_mb is True and spec.wcache is not None:
                    spec.wcache[ck] = (w8, ws, wt8, wts)
            y = ops.gemm_mxfp8(x8, xs, w8, ws, spec.fmt_fwd, spec.fmt_fwd, bias=bias_bf16)
            ctx.saved_fp8 = (xt8, xts, wt8, wts, None)
            if spec.dy_handoff is not None and bias is None and (need_wgrad or need_dgrad):
                spec.dy_handoff.offer(None, None, spec.fmt_bwd, need_dgrad, need_wgrad, mx=True)
        else:
            mf, g = spec.meta_fwd, spec.g
            if ln_w is not None:
                gam = (ln_w if ln_w.dtype == torch.bfloat16 else ln_w.to(torch.bfloat16)).contiguous()
                rstd = spec.rstd if spec.rstd is not None else ops.rmsnorm_stats(x2, spec.eps)
                x8, x8t = ops.norm_cast(x2, rstd, gam, mf.scale(3 * g), mf.amax(3 * g), spec.fmt_fwd, want_t=need_wgrad)
                if need_dgrad:
                    ctx.norm = (x2, rstd, gam, ln_w.dtype)
            else:
                x8, x8t = ops.cast_amax(x2, mf.scale(3 * g), mf.amax(3 * g), spec.fmt_fwd, want_t=need_wgrad)
            w8, w8t, siw = _cast_weights(spec, g, weights, ns, N, K, dev, need_dgrad)
            y = ops.gemm_fp8(x8, w8, mf.scale_inv(3 * g), siw, spec.fmt_fwd, spec.fmt_fwd, bias=bias_bf16)
            # scale_inv as of quantisation time: the arena is updated at autocast exit, before backward
            sinv = (mf.scale_inv_snapshot()[3 * g:3 * g + 1], siw) if (need_wgrad or need_dgrad) else None
            ctx.saved_fp8 = (x8t, None, w8t, None, sinv)
            if spec.dy_handoff is not None and bias is None and spec.meta_bwd is not None and (need_wgrad or need_dgrad):
                spec.dy_handoff.offer(spec.meta_bwd.scale(2 * g), spec.meta_bwd.amax(2 * g), spec.fmt_bwd, need_dgrad, need_wgrad)
        ctx.spec, ctx.ns, ctx.x_shape, ctx.x_dtype = spec, ns, x.shape, x.dtype
        ctx.w_dtypes = [w.dtype for w in weights]
        ctx.w_refs = weights if need_wgrad else None  # the Parameters themselves (not saved tensors): for _wgrad_out
        ctx.has_bias, ctx.bias_dtype = bias is not None, (None if bias is None else bias.dtype)
        ctx.need_wgrad, ctx.need_dgrad = need_wgrad, need_dgrad
        if spec.with_skip:
            ctx.set_materialize_grads(False)
            return y.view(*x.shape[:-1], N), x
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy: torch.Tensor, dskip: Optional[torch.Tensor] = None):
        spec = ctx.spec
        g2 = _as_bf16_2d(dy)
        M, N = g2.shape
        xt8, xts, wt8, wts, sinv = ctx.saved_fp8
        ctx.saved_fp8 = None
        dx = dw = db_fused = None
        if spec.recipe.mxfp8():
            if spec.dy_handoff is not None and spec.dy_handoff.fp8 is not None:
                g8, gs, gt8, gts = spec.dy_handoff.take(dy)  # already quantised by the op that produced it (dy is a placeholder)
            elif ctx.has_bias and ctx.bias_dtype in (torch.bfloat16, torch.float32):  # the bias gradient rides on the quantisation
                g8, gs, gt8, gts, cs = ops.mxfp8_quantize(g2, spec.fmt_bwd, rowwise=ctx.need_dgrad, colwise=ctx.need_wgrad,
                                                          want_colsum=True)
                db_fused = ops.colsum_finish(cs, ctx.bias_dtype)
            else:
                g8, gs, gt8, gts = ops.mxfp8_quantize(g2, spec.fmt_bwd, rowwise=ctx.need_dgrad, colwise=ctx.need_wgrad)
            if ctx.need_dgrad:
                dx = ops.gemm_mxfp8(g8, gs, wt8, wts, spec.fmt_bwd, spec.fmt_fwd)
            if ctx.need_wgrad:
                dw = ops.gemm_mxfp8(gt8, gts, xt8, xts, spec.fmt_bwd, spec.fmt_fwd, out=_wgrad_out(ctx.w_refs, xt8.shape[0]))
        else:
            mb, g = spec.meta_bwd, spec.g
            if spec.dy_handoff is not None and spec.dy_handoff.fp8 is not None:
                g8, g8t = spec.dy_handoff.take(dy)  # already quantised by the op that produced it (dy is a placeholder)
            elif ctx.has_bias:  # the bias gradient rides on the cast of dy
                g8, g8t, cs = ops.cast_amax(g2, mb.scale(2 * g), mb.amax(2 * g), spec.fmt_bwd,
                                            want_y=ctx.need_dgrad, want_t=ctx.need_wgrad, want_colsum=True)
                db_fused = ops.colsum_finish(cs, ctx.bias_dtype)
            else:
                g8, g8t = ops.cast_amax(g2, mb.scale(2 * g), mb.amax(2 * g), spec.fmt_bwd,
                                        want_y=ctx.need_dgrad, want_t=ctx.need_wgrad)
            sig = mb.scale_inv(2 * g)
            dx, dw = _dgrad_wgrad(g8, wt8, g8t, xt8, sig, sinv[1], sinv[0], spec.fmt_bwd, spec.fmt_fwd,
                                  _wgrad_out(ctx.w_refs, xt8.shape[0]) if ctx.need_wgrad else None, ctx.need_dgrad, ctx.need_wgrad)
        db = None
        if ctx.has_bias:
            db = db_fused if db_fused is not None else g2.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
        dln = None
        if ctx.norm is not None and dx is not None:
            xin, rstd, gam, ln_dtype = ctx.norm
            ctx.norm = None
            dx, dln = ops.rmsnorm_bwd(dx, xin, rstd, gam, dres=_skip_2d(dskip, dx), dgamma_dtype=ln_dtype)
            dskip = None
        if spec.trigger_bwd_update:
            # this GEMM belongs to the first FP8 module of the outermost autocast: its backward is the last
            FP8GlobalStateManager.reduce_and_update_fp8_tensors(forward=False)
        if dx is not None:
            dx = dx.view(ctx.x_shape).to(ctx.x_dtype)
        if dskip is not None:
            dx = dskip if dx is None else dx + dskip
        dws: List[Optional[torch.Tensor]] = [None] * len(ctx.ns)
        if dw is not None:
            parts = torch.split(dw, ctx.ns, dim=0)
            dws = [p if p.dtype == dt else p.to(dt) for p, dt in zip(parts, ctx.w_dtypes)]
        return (dx, db, None, dln, *dws)


class _FP8SwiGLUMLPFn(torch.autograd.Function):
    """fc1 (FP8) -> SwiGLU fused with the FP8 cast of fc2's input -> fc2 (FP8), delayed scaling: TE's LayerNormMLP
    structure (te_llama.py:58-63).  The bf16 activation is never materialised; backward fuses dSwiGLU with the cast of
    fc1's grad_output and returns the fc1 bias gradient from the same pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, spec: _GemmSpec, ln_w=None):
        x2 = _as_bf16_2d(x)
        M, K = x2.shape
        if M % 8 or K % 16:
            raise RuntimeError(f"FP8 LayerNormMLP needs tokens % 8 == 0 and hidden % 16 == 0, got {M} x {K}")
        mf, fmt = spec.meta_fwd, spec.fmt_fwd
        need_dgrad = bool(ctx.needs_input_grad[0]) or bool(ctx.needs_input_grad[6])
        need_w = bool(ctx.needs_input_grad[1]) or bool(ctx.needs_input_grad[3])
        bwd = need_dgrad or need_w
        ctx.norm = None
        ctx.spec, ctx.x_shape, ctx.x_dtype = spec, x.shape, x.dtype
        ctx.dtypes = (w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype)
        ctx.need_dgrad, ctx.need_w = need_dgrad, need_w
        ctx.w_refs = (w1, w2) if need_w else None  # the Parameters themselves: for _wgrad_out
        if spec.recipe.mxfp8():
            return _FP8SwiGLUMLPFn._forward_mx(ctx, x, x2, w1, b1, w2, b2, spec, ln_w, need_dgrad, need_w, bwd)
        if ln_w is not None:  # K9: x is the un-normalised input
            gam = (ln_w if ln_w.dtype == torch.bfloat16 else ln_w.to(torch.bfloat16)).contiguous()
            rstd = spec.rstd if spec.rstd is not None else ops.rmsnorm_stats(x2, spec.eps)
            x8, x8t = ops.norm_cast(x2, rstd, gam, mf.scale(0), mf.amax(0), fmt, want_t=need_w)
            if need_dgrad:
                ctx.norm = (x2, rstd, gam, ln_w.dtype)
        else:
            x8, x8t = ops.cast_amax(x2, mf.scale(0), mf.amax(0), fmt, want_t=need_w)
        dev = x2.device
        w1_8, w1_8t, si1 = _cast_weights(spec, 0, (w1,), [w1.shape[0]], w1.shape[0], K, dev, bwd)
        # fc1: the GEMM leaves the bias out, the SwiGLU kernels add it (fp32) to the gate / up values they unpack anyway
        b1_bf = None if b1 is None else b1.detach().to(torch.bfloat16).contiguous()
        fuse_b1 = b1_bf is not None and _FUSE_MLP_BIAS and (b1_bf.data_ptr() % 16 == 0)
        h = ops.gemm_fp8(x8, w1_8, mf.scale_inv(0), si1, fmt, fmt, bias=None if (b1_bf is None or fuse_b1) else b1_bf)
        a8, a8t = ops.swiglu_cast(h, mf.scale(3), mf.amax(3), fmt, want_t=need_w, bias=b1_bf if fuse_b1 else None)
        ctx.b1_fused = b1_bf if fuse_b1 else None
        w2_8, w2_8t, si2 = _cast_weights(spec, 1, (w2,), [w2.shape[0]], w2.shape[0], w2.shape[1], dev, bwd)
        # fc2: with defer_bias the caller adds the bias in its residual add (LayerNormMLP.forward hands it over)
        y = ops.gemm_fp8(a8, w2_8, mf.scale_inv(3), si2, fmt, fmt,
                         bias=None if (b2 is None or spec.defer_bias) else b2.to(torch.bfloat16).contiguous())
        snap = mf.scale_inv_snapshot() if bwd else None
        sinv = (snap[0:1], si1, snap[3:4], si2) if bwd else None  # scale_inv of x, w1, act, w2 as of quantisation time
        ctx.saved_fp8 = (x8t, w1_8t, a8t, w2_8t, h if bwd else None, sinv)
        ctx.spec, ctx.x_shape, ctx.x_dtype = spec, x.shape, x.dtype
        ctx.dtypes = (w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype)
        ctx.need_dgrad, ctx.need_w = need_dgrad, need_w
        if spec.with_skip:
            ctx.set_materialize_grads(False)
            return y.view(*x.shape[:-1], w2.shape[0]), x
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def _mx_weights(spec, g, w, need_t):
        ck = ("mx", g)
        hit = spec.wcache.get(ck) if (spec.first_mb is False and spec.wcache is not None) else None
        if hit is not None and (hit[2] is not None or not need_t):
            return hit
        q = _mx_sink_copies(spec, g, [w], [w.shape[0]], w.shape[0], w.shape[1], w.device)
        if q is None:
            wb = (w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)).contiguous()
            q = ops.mxfp8_quantize(wb, spec.fmt_fwd, rowwise=True, colwise=need_t or spec.first_mb is True)
        if spec.first_mb is True and spec.wcache is not None:
            spec.wcache[ck] = q
        return q

    @staticmethod
    def _forward_mx(ctx, x, x2, w1, b1, w2, b2, spec, ln_w, need_dgrad, need_w, bwd):
        fmt = spec.fmt_fwd
        if ln_w is not None:
            gam = (ln_w if ln_w.dtype == torch.bfloat16 else ln_w.to(torch.bfloat16)).contiguous()
            rstd = spec.rstd if spec.rstd is not None else ops.rmsnorm_stats(x2, spec.eps)
            x8, xs, xt8, xts = ops.mxfp8_norm_quantize(x2, rstd, gam, fmt, rowwise=True, colwise=need_w)
            if need_dgrad:
                ctx.norm = (x2, rstd, gam, ln_w.dtype)
        else:
            x8, xs, xt8, xts = ops.mxfp8_quantize(x2, fmt, rowwise=True, colwise=need_w)
        w1_8, w1s, w1t8, w1ts = _FP8SwiGLUMLPFn._mx_weights(spec, 0, w1, bwd)
        h = ops.gemm_mxfp8(x8, xs, w1_8, w1s, fmt, fmt, bias=None if b1 is None else b1.to(torch.bfloat16).contiguous())
        a8, as_, at8, ats = ops.mxfp8_swiglu_quantize(h, fmt, rowwise=True, colwise=need_w)
        w2_8, w2s, w2t8, w2ts = _FP8SwiGLUMLPFn._mx_weights(spec, 1, w2, bwd)
        y = ops.gemm_mxfp8(a8, as_, w2_8, w2s, fmt, fmt, bias=None if (b2 is None or spec.defer_bias) else b2.to(torch.bfloat16).contiguous())
        ctx.saved_fp8 = ((xt8, xts), (w1t8, w1ts), (at8, ats), (w2t8, w2ts), h if bwd else None, None)
        if spec.with_skip:
            ctx.set_materialize_grads(False)
            return y.view(*x.shape[:-1], w2.shape[0]), x
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def _backward_mx(ctx, dy):
        spec = ctx.spec
        fmt_f, fmt_b = spec.fmt_fwd, spec.fmt_bwd
        (xt8, xts), (w1t8, w1ts), (at8, ats), (w2t8, w2ts), h, _ = ctx.saved_fp8
        ctx.saved_fp8 = None
        g2 = _as_bf16_2d(dy)
        if ctx.dtypes[3] is not None:  # fc2 bias gradient rides on the quantisation of dy
            g8, gs, gt8, gts, cs2 = ops.mxfp8_quantize(g2, fmt_b, rowwise=True, colwise=ctx.need_w, want_colsum=True)
            db2 = (cs2, ctx.dtypes[3])  # partial sums: finished with the layer's other column sums in _finish_backward
        else:
            g8, gs, gt8, gts = ops.mxfp8_quantize(g2, fmt_b, rowwise=True, colwise=ctx.need_w)
            db2 = None
        dact = ops.gemm_mxfp8(g8, gs, w2t8, w2ts, fmt_b, fmt_f)
        dw2 = ops.gemm_mxfp8(gt8, gts, at8, ats, fmt_b, fmt_f, out=_wgrad_out(ctx.w_refs[1:], at8.shape[0])) if ctx.need_w else None
        want_b1 = ctx.dtypes[1] is not None
        dh8, dhs, dht8, dhts, colsum = ops.mxfp8_dswiglu_quantize(h, dact, fmt_b, rowwise=ctx.need_dgrad, colwise=ctx.need_w,
                                                                  want_colsum=want_b1)
        db1 = (colsum, ctx.dtypes[1]) if want_b1 else None
        dx = ops.gemm_mxfp8(dh8, dhs, w1t8, w1ts, fmt_b, fmt_f) if ctx.need_dgrad else None
        dw1 = ops.gemm_mxfp8(dht8, dhts, xt8, xts, fmt_b, fmt_f, out=_wgrad_out(ctx.w_refs[:1], xt8.shape[0])) if ctx.need_w else None
        return dx, dw1, db1, dw2, db2

    @staticmethod
    def backward(ctx, dy, dskip=None):
        spec = ctx.spec
        if spec.recipe.mxfp8():
            dx, dw1, db1, dw2, db2 = _FP8SwiGLUMLPFn._backward_mx(ctx, dy)
            return _FP8SwiGLUMLPFn._finish_backward(ctx, dx, dw1, db1, dw2, db2, dskip)
        mb, fmt_f, fmt_b = spec.meta_bwd, spec.fmt_fwd, spec.fmt_bwd
        x8t, w1_8t, a8t, w2_8t, h, sinv = ctx.saved_fp8
        ctx.saved_fp8 = None
        g2 = _as_bf16_2d(dy)
        # fc2 backward (GEMM index 1: bwd slot 2)
        if ctx.dtypes[3] is not None:  # fc2 bias gradient rides on the cast of dy
            g8, g8t, cs2 = ops.cast_amax(g2, mb.scale(2), mb.amax(2), fmt_b, want_t=ctx.need_w, want_colsum=True)
            db2 = (cs2, ctx.dtypes[3])  # partial sums: finished with the layer's other column sums in _finish_backward
        else:
            g8, g8t = ops.cast_amax(g2, mb.scale(2), mb.amax(2), fmt_b, want_t=ctx.need_w)
            db2 = None
        dact, dw2 = _dgrad_wgrad(g8, w2_8t, g8t, a8t, mb.scale_inv(2), sinv[3], sinv[2], fmt_b, fmt_f,
                                 _wgrad_out(ctx.w_refs[1:], a8t.shape[0]) if ctx.need_w else None, True, ctx.need_w)
        # dSwiGLU + cast of fc1's grad_output (GEMM index 0: bwd slot 0) + fc1 bias gradient
        want_b1 = ctx.dtypes[1] is not None
        dh8, dh8t, colsum = ops.dswiglu_cast(h, dact, mb.scale(0), mb.amax(0), fmt_b, want_y=ctx.need_dgrad,
                                             want_t=ctx.need_w, want_colsum=want_b1, bias=getattr(ctx, "b1_fused", None))
        db1 = (colsum, ctx.dtypes[1]) if want_b1 else None
        dx, dw1 = _dgrad_wgrad(dh8, w1_8t, dh8t, x8t, mb.scale_inv(0), sinv[1], sinv[0], fmt_b, fmt_f,
                                _wgrad_out(ctx.w_refs[:1], x8t.shape[0]) if ctx.need_w else None, ctx.need_dgrad, ctx.need_w)
        return _FP8SwiGLUMLPFn._finish_backward(ctx, dx, dw1, db1, dw2, db2, dskip)

    @staticmethod
    def _finish_backward(ctx, dx, dw1, db1, dw2, db2, dskip=None):
        """db1 / db2 arrive as (partial column sums, dtype): one launch finishes them together with the RMSNorm weight gradient."""
        spec = ctx.spec
        dln = None
        if ctx.norm is not None and dx is not None:
            xin, rstd, gam, ln_dtype = ctx.norm
            ctx.norm = None
            dx, dln = ops.rmsnorm_bwd(dx, xin, rstd, gam, dres=_skip_2d(dskip, dx), dgamma_dtype=ln_dtype, finish=False)
            dln = (dln, ln_dtype)
            dskip = None
        pend = [t for t in (db1, db2, dln) if t is not None]
        if pend:
            done = iter(ops.colsum_finish_multi(pend))
            db1 = next(done) if db1 is not None else None
            db2 = next(done) if db2 is not None else None
            dln = next(done) if dln is not None else None
        if spec.trigger_bwd_update:
            FP8GlobalStateManager.reduce_and_update_fp8_tensors(forward=False)
        if dx is not None:
            dx = dx.view(ctx.x_shape).to(ctx.x_dtype)
        if dskip is not None:
            dx = dskip if dx is None else dx + dskip
        if dw1 is not None and dw1.dtype != ctx.dtypes[0]:
            dw1 = dw1.to(ctx.dtypes[0])
        if dw2 is not None and dw2.dtype != ctx.dtypes[2]:
            dw2 = dw2.to(ctx.dtypes[2])
        return dx, dw1, db1, dw2, db2, None, dln


class _FP8Module(torch.nn.Module):
    """Shared FP8 bookkeeping: lazily allocated meta windows, `_extra_state` (TE serialises its FP8
    metadata there; it ends up in `save_pretrained`, train_fp8.py:668-669)."""

    num_gemms = 1

    def __init__(self):
        super().__init__()
        self._meta_fwd: Optional[ModuleMeta] = None
        self._meta_bwd: Optional[ModuleMeta] = None
        self._meta_key = None
        self._pending_state = None
        self._wcache = {}  # FP8 weights kept across micro-batches (is_first_microbatch protocol)
        # used when forward() is called without `is_first_microbatch` (the decoder layer does not thread the flag through):
        # the training harness sets it per micro-batch of a gradient-accumulation window (train.train_step)
        self.default_is_first_microbatch = None

    def _prepare(self, device) -> Optional[Tuple[Recipe, Optional[ModuleMeta], Optional[ModuleMeta], bool]]:
        """Called at the top of forward.  None -> run the plain bf16 path."""
        if not FP8GlobalStateManager.is_fp8_enabled():
            return None
        recipe = FP8GlobalStateManager.get_fp8_recipe()
        first = FP8GlobalStateManager.is_first_fp8_module()
        if recipe.mxfp8():
            return recipe, None, None, first
        key = (recipe.fp8_format, recipe.amax_history_len, recipe.amax_compute_algo, recipe.margin, str(device))
        if self._meta_key != key:
            fa = FP8GlobalStateManager.arena(recipe, True, device)
            ba = FP8GlobalStateManager.arena(recipe, False, device)
            self._meta_fwd = ModuleMeta(fa, fa.alloc(3 * self.num_gemms), 3 * self.num_gemms)
            self._meta_bwd = ModuleMeta(ba, ba.alloc(2 * self.num_gemms), 2 * self.num_gemms)
            self._meta_key = key
            if self._pending_state is not None:
                self._meta_fwd.load_state(self._pending_state["fwd"])
                self._meta_bwd.load_state(self._pending_state["bwd"])
                self._pending_state = None
        else:  # arenas may carry a new group / reduce flag
            FP8GlobalStateManager.arena(recipe, True, device)
            FP8GlobalStateManager.arena(recipe, False, device)
        return recipe, self._meta_fwd, self._meta_bwd, first

    def get_extra_state(self):
        if self._meta_fwd is None:
            return torch.empty(0, dtype=torch.uint8)
        buf = io.BytesIO()
        to_cpu = lambda d: {k: v.cpu() for k, v in d.items()}
        torch.save({"fwd": to_cpu(self._meta_fwd.state()), "bwd": to_cpu(self._meta_bwd.state())}, buf)
        return torch.frombuffer(bytearray(buf.getvalue()), dtype=torch.uint8)

    def set_extra_state(self, state):
        if state is None or (isinstance(state, torch.Tensor) and state.numel() == 0):
            return
        st = torch.load(io.BytesIO(state.cpu().numpy().tobytes()), weights_only=True)
        if self._meta_fwd is not None:
            self._meta_fwd.load_state(st["fwd"])
            self._meta_bwd.load_state(st["bwd"])
        else:
            self._pending_state = st


def _rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, zero_centered_gamma: bool = False) -> torch.Tensor:
    w = weight + 1 if zero_centered_gamma else weight
    return F.rms_norm(x, (x.shape[-1],), w, eps)


def _layernorm(x, weight, bias, eps, zero_centered_gamma=False):
    w = weight + 1 if zero_centered_gamma else weight
    return F.layer_norm(x, (x.shape[-1],), w, bias, eps)


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-5, params_dtype=None, device="cuda", zero_centered_gamma=False):
        super().__init__()
        self.eps, self.zero_centered_gamma = eps, zero_centered_gamma
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, dtype=params_dtype or torch.get_default_dtype(), device=device))

    def forward(self, x):
        return _rmsnorm(x, self.weight, self.eps, self.zero_centered_gamma)


class LayerNorm(torch.nn.Module):
    """Present for `isinstance` checks in accelerate (utils/transformer_engine.py:109) and as a plain LN."""

    def __init__(self, hidden_size: int, eps: float = 1e-5, params_dtype=None, device="cuda", zero_centered_gamma=False):
        super().__init__()
        self.eps, self.zero_centered_gamma = eps, zero_centered_gamma
        dt = params_dtype or torch.get_default_dtype()
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, dtype=dt, device=device))
        self.bias = torch.nn.Parameter(torch.zeros(hidden_size, dtype=dt, device=device))

    def forward(self, x):
        return _layernorm(x, self.weight, self.bias, self.eps, self.zero_centered_gamma)


def _init_weight(shape, dtype, device, init_method=None):
    w = torch.empty(shape, dtype=dtype, device=device)
    if init_method is not None:
        init_method(w)
    else:
        torch.nn.init.normal_(w, mean=0.0, std=0.023)  # TE default init_method_normal(0.023)
    return torch.nn.Parameter(w)


class Linear(_FP8Module):
    """Drop-in for `te.pytorch.Linear(in_features, out_features, bias=..., params_dtype=...)`."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, params_dtype=None, device="cuda",
                 init_method=None, **_ignored):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        dt = params_dtype or torch.get_default_dtype()
        self.weight = _init_weight((out_features, in_features), dt, device, init_method)
        self.bias = torch.nn.Parameter(torch.zeros(out_features, dtype=dt, device=device)) if bias else None
        self.use_bias = bias

    def forward(self, inp: torch.Tensor, is_first_microbatch=None) -> torch.Tensor:
        if is_first_microbatch is None:
            is_first_microbatch = self.default_is_first_microbatch
        st = self._prepare(inp.device)
        if st is None:
            if getattr(self, "_pending_norm", None) is not None:
                raise RuntimeError("Linear: a deferred RMSNorm is pending but FP8 is off for this forward")
            return F.linear(inp, _master(self.weight).to(inp.dtype), None if self.bias is None else self.bias.to(inp.dtype))
        recipe, mf, mb, first = st
        # `offer_dy_handoff` (set on the lm_head by train.prepare_model): the output carries a DyHandoff through which the op
        # that consumes it directly (loss.causal_lm_loss) can deliver this layer's grad_output already quantised
        handoff = DyHandoff() if (getattr(self, "offer_dy_handoff", False) and self.training and torch.is_grad_enabled()
                                  and self.bias is None) else None
        # `_pending_norm` (llama._DeferredFinalNorm, the causal-LM head only): the RMSNorm in front of this Linear handed its
        # weight over instead of running, and is fused into the input cast as in LayerNormLinear (K9)
        pn, self._pending_norm = getattr(self, "_pending_norm", None), None
        ln_w, eps, rstd = None, 1e-5, None
        if pn is not None:
            ln_w, eps, rs, ptr, shape = pn
            if inp.data_ptr() != ptr or tuple(inp.shape) != shape:
                raise RuntimeError("Linear: a deferred RMSNorm is pending for another tensor than the one this forward received")
            rstd = _usable_rstd(rs, inp, eps)
        spec = _GemmSpec(recipe, mf, mb, 0, first, self.training, eps, wcache=self._wcache, first_mb=is_first_microbatch,
                         rstd=rstd, dy_handoff=handoff)
        y = _FP8LinearFn.apply(inp, self.bias, spec, ln_w, self.weight)
        if handoff is not None and handoff.offered():
            y._mi_dy_handoff = handoff
        return y

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.use_bias}"


class LayerNormLinear(_FP8Module):
    """Norm -> FP8 Linear.  With `parameters_split` the weight is kept as separate Parameters
    (`query_weight`, `key_weight`, `value_weight`, te_llama.py:200-217) that form ONE GEMM operand and
    share ONE weight amax/scale slot (SURVEY.md Appendix A "Fused-QKV")."""

    def __init__(self, in_features: int, out_features: int, eps: float = 1e-5, bias: bool = True,
                 normalization: str = "LayerNorm", parameters_split=None, params_dtype=None, device="cuda",
                 zero_centered_gamma: bool = False, init_method=None, return_layernorm_output: bool = False, **_ignored):
        super().__init__()
        assert normalization in ("LayerNorm", "RMSNorm")
        self.in_features, self.out_features, self.eps = in_features, out_features, eps
        self.normalization, self.zero_centered_gamma = normalization, zero_centered_gamma
        self.return_layernorm_output = return_layernorm_output
        dt = params_dtype or torch.get_default_dtype()
        self.layer_norm_weight = torch.nn.Parameter(
            torch.zeros(in_features, dtype=dt, device=device) if zero_centered_gamma else torch.ones(in_features, dtype=dt, device=device))
        self.layer_norm_bias = (torch.nn.Parameter(torch.zeros(in_features, dtype=dt, device=device))
                                if normalization == "LayerNorm" else None)
        if parameters_split is None:
            self.weight_names, sizes = ["weight"], [out_features]
            self.bias_names = ["bias"]
        else:
            if isinstance(parameters_split, dict):
                names, sizes = list(parameters_split.keys()), list(parameters_split.values())
            else:
                names = list(parameters_split)
                assert out_features % len(names) == 0
                sizes = [out_features // len(names)] * len(names)
            assert sum(sizes) == out_features
            self.weight_names = [f"{n.rstrip('_')}_weight" for n in names]
            self.bias_names = [f"{n.rstrip('_')}_bias" for n in names]
        self.split_sizes = sizes
        for n, sz in zip(self.weight_names, sizes):
            setattr(self, n, _init_weight((sz, in_features), dt, device, init_method))
        self.use_bias = bias
        for n, sz in zip(self.bias_names, sizes):
            if bias:
                setattr(self, n, torch.nn.Parameter(torch.zeros(sz, dtype=dt, device=device)))
            else:
                setattr(self, n, None)

    def _weights(self):
        return [getattr(self, n) for n in self.weight_names]

    def _bias(self):
        if not self.use_bias:
            return None
        bs = [getattr(self, n) for n in self.bias_names]
        return bs[0] if len(bs) == 1 else torch.cat(bs, 0)

    def _norm(self, x):
        if self.normalization == "RMSNorm":
            return _rmsnorm(x, self.layer_norm_weight, self.eps, self.zero_centered_gamma)
        return _layernorm(x, self.layer_norm_weight, self.layer_norm_bias, self.eps, self.zero_centered_gamma)

    def forward(self, inp: torch.Tensor, is_first_microbatch=None, _with_skip: bool = False, _rstd=None, _dy_handoff=None):
        """`_with_skip` (extension used by MultiheadAttention / the decoder layer): returns (out, skip) where `skip` carries
        `inp` for the residual add, its gradient fused into the RMSNorm backward when the fused-norm path is active.
        `_rstd`: (rstd, eps) of `inp` from residual_add_stats, used instead of a statistics pass when eps matches."""
        if is_first_microbatch is None:
            is_first_microbatch = self.default_is_first_microbatch
        st = self._prepare(inp.device)
        ws, b = self._weights(), self._bias()
        if st is not None and _can_fuse_norm(self, st[0], inp) and not self.return_layernorm_output:
            recipe, mf, mb, first = st
            return _FP8LinearFn.apply(inp, b, _GemmSpec(recipe, mf, mb, 0, first, self.training, self.eps, self._wcache,
                                                        is_first_microbatch, with_skip=_with_skip, rstd=_usable_rstd(_rstd, inp, self.eps),
                                                        dy_handoff=_dy_handoff),
                                      self.layer_norm_weight, *ws)
        ln = self._norm(inp)
        if st is None:
            w = _master(ws[0]) if len(ws) == 1 else torch.cat([_master(w_) for w_ in ws], 0)
            out = F.linear(ln, w.to(ln.dtype), None if b is None else b.to(ln.dtype))
        else:
            recipe, mf, mb, first = st
            out = _FP8LinearFn.apply(ln, b, _GemmSpec(recipe, mf, mb, 0, first, self.training, wcache=self._wcache,
                                                      first_mb=is_first_microbatch), None, *ws)
        if _with_skip:  # unfused route: the residual is the input itself (autograd adds its gradient)
            return out, inp
        return (out, ln) if self.return_layernorm_output else out


def _usable_rstd(handoff, inp: torch.Tensor, eps: float):
    """rstd from a (rstd, eps) hand-off if it belongs to `inp` (row count) and was computed with this module's eps."""
    if handoff is None:
        return None
    rstd, e = handoff
    if rstd is None or e != eps or rstd.numel() != inp.numel() // inp.shape[-1] or rstd.device != inp.device:
        return None
    return rstd


def _can_fuse_norm(mod, recipe, inp) -> bool:
    """K9 applies to RMSNorm (delayed scaling and MXFP8); the backward kernel keeps a whole row per wave (cols % 512, <= 8192)."""
    h = inp.shape[-1]
    return (getattr(mod, "fused_norm", True) and mod.normalization == "RMSNorm" and not mod.zero_centered_gamma
            and (recipe.delayed() or recipe.mxfp8()) and h % 512 == 0 and h // 512 in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16))


def _swiglu(a: torch.Tensor) -> torch.Tensor:
    f = a.shape[-1] // 2
    return F.silu(a[..., :f]) * a[..., f:]


_ACTS = {
    "swiglu": (_swiglu, 2), "gelu": (lambda a: F.gelu(a, approximate="tanh"), 1), "relu": (F.relu, 1),
    "geglu": (lambda a: F.gelu(a[..., :a.shape[-1] // 2], approximate="tanh") * a[..., a.shape[-1] // 2:], 2),
    "silu": (F.silu, 1),
}


class LayerNormMLP(_FP8Module):
    """Norm -> fc1 (FP8) -> activation -> fc2 (FP8); parameters `layer_norm_weight`, `fc1_weight`
    [gate|up stacked: 2*ffn, hidden], `fc1_bias`, `fc2_weight`, `fc2_bias` (te_llama.py:58-63,219-238).
    `bias=True` is TE's default and the reference does not override it."""

    num_gemms = 2

    def __init__(self, hidden_size: int, ffn_hidden_size: int, eps: float = 1e-5, bias: bool = True,
                 normalization: str = "LayerNorm", activation: str = "gelu", params_dtype=None, device="cuda",
                 zero_centered_gamma: bool = False, init_method=None, output_layer_init_method=None, **_ignored):
        super().__init__()
        assert normalization in ("LayerNorm", "RMSNorm") and activation in _ACTS
        self.hidden_size, self.ffn_hidden_size, self.eps = hidden_size, ffn_hidden_size, eps
        self.normalization, self.activation, self.zero_centered_gamma = normalization, activation, zero_centered_gamma
        self.act_fn, mult = _ACTS[activation]
        dt = params_dtype or torch.get_default_dtype()
        self.layer_norm_weight = torch.nn.Parameter(
            torch.zeros(hidden_size, dtype=dt, device=device) if zero_centered_gamma else torch.ones(hidden_size, dtype=dt, device=device))
        self.layer_norm_bias = (torch.nn.Parameter(torch.zeros(hidden_size, dtype=dt, device=device))
                                if normalization == "LayerNorm" else None)
        self.fc1_weight = _init_weight((mult * ffn_hidden_size, hidden_size), dt, device, init_method)
        self.fc2_weight = _init_weight((hidden_size, ffn_hidden_size), dt, device, output_layer_init_method or init_method)
        self.use_bias = bias
        self.fc1_bias = torch.nn.Parameter(torch.zeros(mult * ffn_hidden_size, dtype=dt, device=device)) if bias else None
        self.fc2_bias = torch.nn.Parameter(torch.zeros(hidden_size, dtype=dt, device=device)) if bias else None
        self.fused_swiglu = True  # K10: SwiGLU fused with the FP8 cast (delayed scaling); False -> two Linears + torch ops

    def _norm(self, x):
        if self.normalization == "RMSNorm":
            return _rmsnorm(x, self.layer_norm_weight, self.eps, self.zero_centered_gamma)
        return _layernorm(x, self.layer_norm_weight, self.layer_norm_bias, self.eps, self.zero_centered_gamma)

    def forward(self, inp: torch.Tensor, is_first_microbatch=None, _with_skip: bool = False, _rstd=None, _defer_bias: bool = False):
        """`_defer_bias` (extension, with `_with_skip`): returns (out, skip, bias) where `out` lacks the fc2 bias and `bias` is the
        tensor the caller must add (residual_add_stats(skip, out, eps, bias=bias)), or None when nothing was deferred."""
        if is_first_microbatch is None:
            is_first_microbatch = self.default_is_first_microbatch
        st = self._prepare(inp.device)
        if (st is not None and self.activation == "swiglu" and self.fused_swiglu and _can_fuse_norm(self, st[0], inp)):
            recipe, mf, mb, first = st  # K9 + K10: norm -> cast, fc1, SwiGLU -> cast, fc2 in one autograd node
            # (the residual add that takes the deferred fc2 bias is recipe-independent: MXFP8 defers too; the fc1 bias stays in the
            # MXFP8 GEMM's epilogue, its quantising SwiGLU kernels have no bias form)
            defer = bool(_defer_bias and _with_skip and _FUSE_MLP_BIAS and self.fc2_bias is not None)
            res = _FP8SwiGLUMLPFn.apply(inp, self.fc1_weight, self.fc1_bias, self.fc2_weight, self.fc2_bias,
                                        _GemmSpec(recipe, mf, mb, 0, first, self.training, self.eps, self._wcache,
                                                  is_first_microbatch, with_skip=_with_skip, rstd=_usable_rstd(_rstd, inp, self.eps),
                                                  defer_bias=defer),
                                        self.layer_norm_weight)
            if _defer_bias and _with_skip:
                return res[0], res[1], (self.fc2_bias if defer else None)
            return res
        out = self._unfused(inp, st, is_first_microbatch)
        if _defer_bias and _with_skip:
            return out, inp, None
        return (out, inp) if _with_skip else out  # unfused route: the residual is the input itself

    def _unfused(self, inp, st, is_first_microbatch):
        ln = self._norm(inp)
        if st is None:
            h = F.linear(ln, _master(self.fc1_weight).to(ln.dtype), None if self.fc1_bias is None else self.fc1_bias.to(ln.dtype))
            return F.linear(self.act_fn(h), _master(self.fc2_weight).to(ln.dtype),
                            None if self.fc2_bias is None else self.fc2_bias.to(ln.dtype))
        recipe, mf, mb, first = st
        if self.activation == "swiglu" and self.fused_swiglu and (recipe.delayed() or inp.numel() // inp.shape[-1] % 32 == 0):
            return _FP8SwiGLUMLPFn.apply(ln, self.fc1_weight, self.fc1_bias, self.fc2_weight, self.fc2_bias,
                                         _GemmSpec(recipe, mf, mb, 0, first, self.training, wcache=self._wcache,
                                                   first_mb=is_first_microbatch))
        # fc1's backward is the last FP8 op of this module's backward -> it carries the update trigger
        h = _FP8LinearFn.apply(ln, self.fc1_bias, _GemmSpec(recipe, mf, mb, 0, first, self.training, wcache=self._wcache,
                                                            first_mb=is_first_microbatch), None, self.fc1_weight)
        a = self.act_fn(h)
        return _FP8LinearFn.apply(a, self.fc2_bias, _GemmSpec(recipe, mf, mb, 1, False, self.training, wcache=self._wcache,
                                                              first_mb=is_first_microbatch), None, self.fc2_weight)
