"""Counterpart of the reference's training harness on top of llm_fp8_amd: same flag names as
train_fp8.py:684-788 / config.py, same model-selection rules (ModelManager, train_fp8.py:83-124), same
prepare order (convert -> bf16 autocast -> outer fp8 autowrap; accelerate accelerator.py:2098-2131,
1818-1833) and the same optimiser step order (train_fp8.py:270-291).  Data and weights are synthetic /
random-init: the hub dataset, tokenizer and checkpoints are unreachable offline.

Multi-GPU follows train_multi_gpu.py: one process per GPU, `nccl` (= RCCL over xGMI) backend, FSDP
FULL_SHARD around each decoder layer (`:381-460`) or DDP (`:355-378`).
"""
from __future__ import annotations

import argparse
import contextlib
import functools
import json
import math
import os
import time
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import llama
from .pytorch.fp8 import FP8GlobalStateManager


@dataclass
class TrainingConfig:
    """Field names follow config.py:5-47 / train_fp8.py argparse."""
    model_name: str = "meta-llama/Llama-3.2-3B"
    batch_size: int = 16
    max_seq_length: int = 512
    mixed_precision: str = "fp8"          # "bf16" | "fp8"
    fp8_scenario: str = "default"         # "default" | "hybrid" | "mxfp8"
    use_te: bool = True
    learning_rate: float = 1.41e-5
    num_warmup_steps: int = 100
    num_training_steps: int = 1000
    gradient_accumulation_steps: int = 1
    max_grad_norm: float = 1.0
    sharding_mode: str = "auto"           # one of SHARDING_MODES
    seed: int = 42
    num_hidden_layers: Optional[int] = None  # override for small tests only
    vocab_size: Optional[int] = None
    output_dir: Optional[str] = None      # ModelSaver.save_model (train_fp8.py:657-681)
    load_dir: Optional[str] = None        # local checkpoint directory (from_pretrained_local, te_llama.py:100-178)


def create_model(cfg: TrainingConfig, device) -> torch.nn.Module:
    """ModelManager.create_model (train_fp8.py:83-124): TE-style wrapper when --use_te, else plain HF bf16."""
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    over = {}
    if cfg.num_hidden_layers is not None:
        over["num_hidden_layers"] = cfg.num_hidden_layers
    if cfg.vocab_size is not None:
        over["vocab_size"] = cfg.vocab_size
    config = llama.llama_config(cfg.model_name, **over)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(device):
            if cfg.use_te:
                model = llama.TELlamaForCausalLM(config, cfg.fp8_scenario)
            else:
                model = LlamaForCausalLM(config)
    finally:
        torch.set_default_dtype(prev)
    model.to(device)
    model.config.use_cache = False
    if cfg.load_dir:
        from . import checkpoint
        if cfg.use_te:
            checkpoint.load_into(model, cfg.load_dir, config)
        else:  # plain HF model: HF-layout shards load by name
            for shard in checkpoint.resolve_shards(cfg.load_dir):
                model.load_state_dict(checkpoint.load_shard(shard), strict=False)
        model.tie_weights()
    return model


def save_model(model, cfg: TrainingConfig, layout: str = "hf"):
    """ModelSaver.save_model (train_fp8.py:657-681) without the tokenizer (unreachable offline): config.json + safetensors."""
    from . import checkpoint
    # (under ShardedFP8DP the caller runs model.gather_master_weights() on EVERY rank first -- it is a collective; main() does)
    return checkpoint.save_pretrained(checkpoint.unwrap(model), cfg.output_dir, layout=layout, save_fp8_state=True)


def prepare_model(model: torch.nn.Module, cfg: TrainingConfig) -> torch.nn.Module:
    """Accelerator.prepare's model steps for mixed_precision in {bf16, fp8}."""
    if cfg.mixed_precision == "fp8":
        llama.convert_model(model)  # lm_head with --use_te; every q/k/v/o/gate/up/down/lm_head without it
        recipe = llama.outer_recipe_for_scenario(cfg.fp8_scenario) if cfg.use_te else None
        llama.apply_fp8_autowrap(model, recipe)
    if torch.cuda.is_available() and os.environ.get("LLM_FP8_AMD_HF_LOSS") != "1":
        from .loss import causal_lm_loss
        head = model.get_output_embeddings() if hasattr(model, "get_output_embeddings") else None
        if type(head).__name__ == "Linear" and type(head).__module__.startswith("llm_fp8_amd"):
            head.offer_dy_handoff = True  # causal_lm_loss delivers the lm_head's grad_output in FP8 (loss._CEFn)
        try:
            model.loss_function = causal_lm_loss  # HF resolves `self.loss_function(logits=..., labels=..., vocab_size=...)`
        except Exception:
            model._loss_function = causal_lm_loss
    inner = model.forward

    def forward(*args, **kwargs):
        with torch.autocast(device_type="cuda" if torch.cuda.is_available() else "cpu", dtype=torch.bfloat16):
            return inner(*args, **kwargs)

    model.forward = forward
    return model


# auto: replicated (gradient arena) while the replicated training state fits the device, else fsdp_fp8; replicated:
# distributed.GradArenaDP; fsdp_fp8: distributed.ShardedFP8DP (FULL_SHARD counterpart with FP8 all-gather); fsdp_full / ddp: the
# torch wrappers the reference uses (train_multi_gpu.py:414-445, :447-470); none: no wrapper
SHARDING_MODES = ("auto", "replicated", "fsdp_fp8", "fsdp_full", "ddp", "none")


def resolve_sharding_mode(mode: str, model: torch.nn.Module, device) -> str:
    """`auto` (DistributedConfig._auto_detect_sharding, train_multi_gpu.py:137-146, picks FSDP FULL_SHARD whenever there is
    more than one GPU): here the run shards only when it has to -- the replicated state of every BASELINE.json model fits
    one MI355X (3B: 39 GB, 8B: 96 GB of 288 GB), so `auto` = gradient-arena data parallelism (distributed.GradArenaDP),
    and the full-shard mode (`fsdp_fp8`) only for a model whose replicated state would take more than half of the device.  An explicit
    ddp / fsdp_full / replicated request is honoured even at world size 1 (rehearsal on a one-GPU box)."""
    import torch.distributed as dist
    if mode != "auto":
        return mode
    if dist.get_world_size() == 1:
        return "none"
    from .distributed import fits_replicated
    # does not fit: the FULL_SHARD counterpart with FP8 all-gathers (distributed.ShardedFP8DP; masters, gradients and AdamW
    # moments of every GEMM weight at 1/world per rank), not torch FSDP (bf16 gathers in forward and backward, a cast after each)
    return "replicated" if fits_replicated(model, device) else "fsdp_fp8"


def _single_process(model: torch.nn.Module, device) -> torch.nn.Module:
    """No data-parallel wrapper: the embedding table's gradient rows are added in place (distributed._LocalEmbeddingGrad)."""
    if getattr(device, "type", "cpu") == "cuda" and os.environ.get("LLM_FP8_AMD_DENSE_EMBEDDING_GRAD") != "1":
        from .distributed import install_local_embedding_grad
        install_local_embedding_grad(model)
    return model


def wrap_distributed(model: torch.nn.Module, cfg: TrainingConfig, device) -> torch.nn.Module:
    """DistributedWrapper (train_multi_gpu.py:328-510)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or cfg.sharding_mode == "none":
        return _single_process(model, device)
    mode = resolve_sharding_mode(cfg.sharding_mode, model, device)
    if mode == "none":
        return _single_process(model, device)
    if mode == "replicated":
        from .distributed import GradArenaDP
        return GradArenaDP(model, bucket_mb=float(os.environ.get("LLM_FP8_AMD_BUCKET_MB", "256")))
    if mode == "fsdp_fp8":
        # FULL_SHARD counterpart with an FP8 all-gather (distributed.ShardedFP8DP): optimiser state, update and weight cast
        # row-sharded; 1 byte per parameter gathered per step instead of FSDP's two bf16 gathers
        from .distributed import ShardedFP8DP
        return ShardedFP8DP(model, bucket_mb=float(os.environ.get("LLM_FP8_AMD_BUCKET_MB", "256")))
    if mode == "ddp":
        from torch.nn.parallel import DistributedDataParallel as DDP
        return DDP(model, device_ids=[device.index] if device.type == "cuda" else None, gradient_as_bucket_view=True)
    if mode == "fsdp_full":
        from torch.distributed.fsdp import BackwardPrefetch, FullyShardedDataParallel as FSDP, MixedPrecision, ShardingStrategy
        from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy
        from transformers.models.llama.modeling_llama import LlamaDecoderLayer
        layer_cls = {llama.TELlamaDecoderLayer, LlamaDecoderLayer}
        policy = functools.partial(transformer_auto_wrap_policy, transformer_layer_cls=layer_cls)
        mp = MixedPrecision(param_dtype=torch.bfloat16, reduce_dtype=torch.bfloat16, buffer_dtype=torch.bfloat16)
        return FSDP(model, sharding_strategy=ShardingStrategy.FULL_SHARD, auto_wrap_policy=policy, mixed_precision=mp,
                    backward_prefetch=BackwardPrefetch.BACKWARD_PRE, forward_prefetch=True, limit_all_gathers=True,
                    use_orig_params=True, sync_module_states=True,
                    device_id=device if device.type == "cuda" else None)
    raise ValueError(f"unsupported sharding_mode {cfg.sharding_mode!r} (one of {SHARDING_MODES})")


def create_optimizer(model, cfg: TrainingConfig):
    """AdamW(fused=True) + linear warmup/decay (train_fp8.py:200-210).  On the single-GPU path (bf16 parameters on the
    device, no FSDP/DDP wrapper) the clip + AdamW pair runs as `ClippedAdamW` (two HIP passes, same update rule)."""
    groups = model.optimizer_param_groups() if hasattr(model, "optimizer_param_groups") else None
    params = [p for g in groups for p in g["params"]] if groups is not None else [p for p in model.parameters() if p.requires_grad]
    fused = all(p.is_cuda for p in params)
    wrapped = type(model).__name__ in ("FullyShardedDataParallel", "DistributedDataParallel")
    if (fused and not wrapped and all(p.dtype == torch.bfloat16 for p in params)
            and os.environ.get("LLM_FP8_AMD_TORCH_ADAMW") != "1"):
        from .optim import ClippedAdamW
        opt = ClippedAdamW(groups if groups is not None else params, lr=cfg.learning_rate, max_grad_norm=cfg.max_grad_norm)
    else:
        opt = torch.optim.AdamW(params, lr=cfg.learning_rate, fused=fused)

    def lr_lambda(step):
        if step < cfg.num_warmup_steps:
            return float(step) / float(max(1, cfg.num_warmup_steps))
        return max(0.0, float(cfg.num_training_steps - step) / float(max(1, cfg.num_training_steps - cfg.num_warmup_steps)))

    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)
    return opt, sched


def synthetic_batch(cfg: TrainingConfig, vocab_size: int, device, generator=None):
    """What DataCollatorForLanguageModeling(mlm=False) yields without padding (data.py:59-63): labels = input_ids."""
    ids = torch.randint(0, vocab_size, (cfg.batch_size, cfg.max_seq_length), device=device, generator=generator)
    return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids.clone()}


def _set_first_microbatch(model, value):
    """`is_first_microbatch` protocol of the TE modules: FP8 weights are cast on the first micro-batch of an accumulation
    window and reused (w8 / w8T / scale-inverse cached) for the others.  None = cast on every forward (the reference never
    passes the flag)."""
    for m in model.modules():
        if hasattr(m, "_wcache") and hasattr(m, "default_is_first_microbatch"):
            m.default_is_first_microbatch = value


def train_step(model, batch, optimizer, scheduler, cfg: TrainingConfig):
    """One optimiser step of Trainer._train_epoch (train_fp8.py:276-291).  `batch` is one batch dict or, with
    gradient_accumulation_steps = N > 1, a list of N micro-batches: loss / N, no gradient exchange on the non-final
    micro-batches (`no_sync`, train_multi_gpu.py:714-737), FP8 weights cast once per window, one clip + optimiser + LR
    step at the end.  Returns the (mean) loss tensor (no host sync)."""
    micro = batch if isinstance(batch, (list, tuple)) else [batch]
    n = len(micro)
    total = None
    for i, mb in enumerate(micro):
        last = i == n - 1
        if n > 1:
            _set_first_microbatch(model, i == 0)
        ctx = model.no_sync() if (not last and hasattr(model, "no_sync")) else contextlib.nullcontext()
        with ctx:
            loss = model(**mb).loss
            if n > 1:
                loss = loss / n
            loss.backward()
        total = loss.detach() if total is None else total + loss.detach()
    if n > 1:
        _set_first_microbatch(model, None)
    if getattr(optimizer, "max_grad_norm", None) is not None:
        pass  # ClippedAdamW: the clip coefficient is folded into the update
    elif hasattr(model, "clip_grad_norm_"):
        model.clip_grad_norm_(cfg.max_grad_norm)  # FSDP
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.max_grad_norm)
    optimizer.step()
    if hasattr(model, "after_optimizer_step"):
        model.after_optimizer_step()  # ShardedFP8DP: FP8 all-gather of the freshly quantised weight shards
    scheduler.step()
    optimizer.zero_grad()
    return total if n > 1 else loss


def setup_distributed():
    """setup_distributed (train_multi_gpu.py:969-1006): env rendezvous, nccl (= RCCL) backend, one device per rank."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        if os.environ.get("LLM_FP8_AMD_SHARE_DEVICE") == "1":  # rehearsal: several ranks on the one GPU of a test box
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if (world > 1 or os.environ.get("LLM_FP8_AMD_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        backend = os.environ.get("LLM_FP8_AMD_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
        if backend == "nccl":
            # RCCL's kernels on a high-priority stream: a bucket's all-reduce gets CUs as the GEMM's workgroups drain
            # instead of queueing behind the whole backward pass
            kw = {"device_id": device}
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except (AttributeError, TypeError):  # a torch build without the options object: default-priority streams
                pass
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world, device


def main(argv=None):
    ap = argparse.ArgumentParser(description="FP8 fine-tuning harness (synthetic data) on llm_fp8_amd")
    ap.add_argument("--model_name", default="meta-llama/Llama-3.2-3B")
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--max_seq_length", type=int, default=512)
    ap.add_argument("--mixed_precision", choices=["bf16", "fp8"], default="fp8")
    ap.add_argument("--fp8_scenario", choices=["default", "hybrid", "mxfp8"], default="default")
    ap.add_argument("--use_te", action="store_true")
    ap.add_argument("--sharding_mode", choices=SHARDING_MODES, default="auto")
    ap.add_argument("--num_steps", type=int, default=10)
    ap.add_argument("--learning_rate", type=float, default=1.41e-5)
    ap.add_argument("--num_hidden_layers", type=int, default=None)
    ap.add_argument("--vocab_size", type=int, default=None)
    ap.add_argument("--num_warmup_steps", type=int, default=100)
    ap.add_argument("--gradient_accumulation_steps", type=int, default=1)
    ap.add_argument("--output_dir", default=None, help="save config.json + safetensors (HF parameter names) here after training")
    ap.add_argument("--load_dir", default=None, help="local checkpoint directory (model.safetensors[.index.json]) to start from")
    ap.add_argument("--save_layout", choices=["hf", "te"], default="hf")
    ap.add_argument("--repeat_batch", action="store_true", help="debug: train on one fixed synthetic batch")
    a = ap.parse_args(argv)
    cfg = TrainingConfig(model_name=a.model_name, batch_size=a.batch_size, max_seq_length=a.max_seq_length,
                         mixed_precision=a.mixed_precision, fp8_scenario=a.fp8_scenario, use_te=a.use_te,
                         sharding_mode=a.sharding_mode, learning_rate=a.learning_rate,
                         num_hidden_layers=a.num_hidden_layers, vocab_size=a.vocab_size, num_warmup_steps=a.num_warmup_steps,
                         gradient_accumulation_steps=a.gradient_accumulation_steps, output_dir=a.output_dir, load_dir=a.load_dir)
    rank, local, world, device = setup_distributed()
    torch.manual_seed(cfg.seed + rank)
    model = prepare_model(create_model(cfg, device), cfg)
    vocab = model.config.vocab_size
    model = wrap_distributed(model, cfg, device)
    opt, sched = create_optimizer(model, cfg)
    model.train()
    fixed = synthetic_batch(cfg, vocab, device) if a.repeat_batch else None
    for step in range(a.num_steps):
        t0 = time.perf_counter()
        acc = max(1, cfg.gradient_accumulation_steps)
        batches = [fixed if fixed is not None else synthetic_batch(cfg, vocab, device) for _ in range(acc)]
        loss = train_step(model, batches if acc > 1 else batches[0], opt, sched, cfg)
        lv = loss.item()
        if not math.isfinite(lv):
            print("Non-finite loss detected, stopping training.")
            break
        dt = time.perf_counter() - t0
        if rank == 0:
            toks = cfg.batch_size * cfg.max_seq_length * world * acc / dt  # train_multi_gpu.py:751-755
            print(json.dumps({"step": step, "loss": lv, "ms": dt * 1e3, "tokens_per_s": toks}), flush=True)
    if cfg.output_dir:
        if hasattr(model, "gather_master_weights"):
            model.gather_master_weights()  # a collective: every rank takes part, rank 0 writes
        if rank == 0:
            files = save_model(model, cfg, a.save_layout)
            print(json.dumps({"saved": files}), flush=True)
        if hasattr(model, "reshard"):
            model.reshard()


if __name__ == "__main__":
    main()
