#!/usr/bin/env python3
"""Headline benchmark: train tokens/sec + FP8 GEMM % of MFMA peak, Llama-3.2-3B, seq 512, batch 16
(BASELINE.json `metric`).  One "step" = one full optimiser step (forward, loss, backward, grad clip,
fused AdamW, LR step) of the te_llama counterpart on synthetic tokens and random-init weights, with every
decoder Linear and lm_head running the hand-written HIP FP8 path (cast+amax, scale update, MFMA GEMM).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every FP8 GEMM launch of every
second timed step (an event costs ~4.5 us of queue time: bracketing every kernel of every step slowed the step by 5 %,
the GEMMs of every step by 3 %); `cpu_baseline` is the repo's own no-TE HF bf16 path timed on this host's CPU cores
(rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP8_DENSE_PEAK_TFLOPS = 5000.0  # MI355X dense FP8 MFMA peak (MI355X_MICROARCH.md: ~5 PF dense)


def cpu_baseline(model_name: str, seq: int = 128, layers=(2, 4), timed_steps: int = 2):
    """Reference's no-TE HF bf16 path (train_fp8.py:118-124 model, :270-291 step) on the host CPUs.
    Bounded sample: full widths, batch 1 x `seq` tokens, two reduced depths, extrapolated linearly in depth."""
    from llm_fp8_amd import llama, train
    from transformers.models.llama.modeling_llama import LlamaForCausalLM
    full_layers = llama.llama_config(model_name).num_hidden_layers
    times = {}
    for L in layers:
        cfg = train.TrainingConfig(model_name=model_name, batch_size=1, max_seq_length=seq, mixed_precision="bf16",
                                   use_te=False, num_hidden_layers=L, num_warmup_steps=0)
        config = llama.llama_config(model_name, num_hidden_layers=L)
        torch.manual_seed(42)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        try:
            model = LlamaForCausalLM(config)
        finally:
            torch.set_default_dtype(prev)
        model.train()
        opt, sched = train.create_optimizer(model, cfg)
        batch = train.synthetic_batch(cfg, config.vocab_size, torch.device("cpu"))
        train.train_step(model, batch, opt, sched, cfg)  # warm-up
        t0 = time.perf_counter()
        for _ in range(timed_steps):
            train.train_step(model, batch, opt, sched, cfg)
        times[L] = (time.perf_counter() - t0) / timed_steps
        del model, opt, sched
    l0, l1 = layers
    per_layer = max((times[l1] - times[l0]) / (l1 - l0), 0.0)
    fixed = max(times[l0] - l0 * per_layer, 0.0)
    t_full = fixed + full_layers * per_layer
    return {
        "value": seq / t_full, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"HF LlamaForCausalLM bf16 (no TE), {model_name} widths, fwd+bwd+clip+AdamW, batch 1 x seq {seq}; "
                   f"timed at {l0} and {l1} of {full_layers} layers ({times[l0]:.2f}s, {times[l1]:.2f}s per step) and "
                   f"extrapolated linearly in depth to {t_full:.2f}s per step; host has {os.cpu_count()} logical CPUs"),
    }


def pmc_traffic(by_tag):
    """Per-launch fabric bytes of the GEMM kernel from the committed rocprofv3 PMC passes (profiles/), averaged over the
    launch mix of this run; shapes without a PMC row (lm_head) are left out of the average."""
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_gemm_pmc_traffic.json"))
    if not cands:
        return None, "no PMC file"
    path = os.path.join(ROOT, "profiles", cands[-1])  # newest round
    tab = json.load(open(path))["sites"]
    num = den = 0.0
    for tag, v in by_tag.items():
        if tag in tab:
            num += v["launches"] * (tab[tag]["fabric_read_bytes"] + tab[tag]["fabric_write_bytes"])
            den += v["launches"]
    if den == 0:
        return None, "no measured shape in this run"
    return num / den, ("avg bytes per launch over the decoder GEMM sites, rocprofv3 --pmc FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE, "
                       f"separate passes (profiles/{cands[-1]}); counts Infinity-Cache hits, i.e. L2-miss traffic")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="llama-3.2-3b")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--scenario", default="default", choices=["default", "hybrid", "mxfp8"])
    ap.add_argument("--sharding_mode", default="auto")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the FP8 path)")
    from llm_fp8_amd import train
    from llm_fp8_amd.pytorch.profiler import KernelTimer
    import torch.distributed as dist

    rank, local, world, device = train.setup_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    cfg = train.TrainingConfig(model_name=args.model, batch_size=args.batch, max_seq_length=args.seq,
                               mixed_precision="fp8", fp8_scenario=args.scenario, use_te=True,
                               sharding_mode=args.sharding_mode, num_hidden_layers=args.layers)
    torch.manual_seed(cfg.seed)  # same weights on every rank; data differs per rank below
    model = train.prepare_model(train.create_model(cfg, device), cfg)
    vocab = model.config.vocab_size
    n_layers = model.config.num_hidden_layers
    resolved_mode = train.resolve_sharding_mode(args.sharding_mode, model, device) if dist.is_initialized() else "none"
    model = train.wrap_distributed(model, cfg, device)
    opt, sched = train.create_optimizer(model, cfg)
    model.train()
    gen = torch.Generator(device=device).manual_seed(cfg.seed + rank)
    batches = [train.synthetic_batch(cfg, vocab, device, gen) for _ in range(4)]  # resident in HBM before timing

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    aux_timer = KernelTimer(kinds=("cast_amax", "mxfp8_quantize", "attn_fwd", "attn_bwd"))  # informational, outside the timed region
    for i in range(args.warmup):
        if i == args.warmup - 1 and not args.no_kernel_timing:
            with aux_timer.install():
                train.train_step(model, batches[i % 4], opt, sched, cfg)
        else:
            train.train_step(model, batches[i % 4], opt, sched, cfg)
    # inside the timed region only the FP8 GEMM launches are bracketed by HIP events (every event costs queue time)
    timer = KernelTimer(kinds=("gemm_fp8", "gemm_mxfp8"))
    sync()
    t0 = time.perf_counter()
    if args.no_kernel_timing:
        for i in range(args.steps):
            loss = train.train_step(model, batches[i % 4], opt, sched, cfg)
    else:
        for i in range(args.steps):
            if i % 2 == 0:  # GEMM launches of every other timed step carry HIP events (each event costs ~4.5 us of queue time)
                with timer.install():
                    loss = train.train_step(model, batches[i % 4], opt, sched, cfg)
            else:
                loss = train.train_step(model, batches[i % 4], opt, sched, cfg)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss_val = float(loss.item())

    if rank == 0:
        tokens = args.batch * args.seq * world * args.steps
        out = {
            "metric": "train tokens/sec + FP8 GEMM % of MFMA peak, Llama-3.2-3B seq512 b16",
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8 (E4M3/E5M2 operands, fp32 accumulate, bf16 activations)", "data": "synthetic",
            "config": {"workload": (f"{args.model} ({n_layers} layers) full fine-tuning step, batch {args.batch}/GPU x seq {args.seq}, "
                                     f"fp8_scenario={args.scenario} (te_llama counterpart, lm_head FP8 under the outer recipe), "
                                     "random-init weights, synthetic tokens"),
                       "global_batch": args.batch * world, "seq_len": args.seq,
                       "parallelism": "single" if resolved_mode == "none" else
                       f"dp{world} {resolved_mode}" + (" (gradient arena, bucketed RCCL all-reduce)" if resolved_mode == "replicated" else "")},
            "final_loss": loss_val,
        }
        if not args.no_kernel_timing:
            summ = timer.summarize()
            kind = "gemm_mxfp8" if args.scenario == "mxfp8" and "gemm_mxfp8" in summ else "gemm_fp8"
            g = summ.get(kind)
            if g and g["seconds"] > 0:
                achieved = g["work"] / g["seconds"] / 1e12
                traffic, traffic_note = pmc_traffic(g["by_tag"])
                out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": FP8_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": achieved / FP8_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                                   "avg_algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
                                   "kernel": kind, "launches": g["launches"],
                                   "avg_launch_us": g["seconds"] / g["launches"] * 1e6,
                                   "avg_flop_per_launch": g["work"] / g["launches"],
                                   "gemm_ms_per_step": g["seconds"] / ((args.steps + 1) // 2) * 1e3,
                                   "bracketed_steps": f"every 2nd of the {args.steps} timed steps ({(args.steps + 1) // 2} steps, {g['launches']} launches)"}
                sites = {}
                for tag, v in sorted(g["by_tag"].items(), key=lambda kv: -kv[1]["seconds"]):
                    sites[tag] = {"tflops": v["work"] / v["seconds"] / 1e12, "us": v["seconds"] / v["launches"] * 1e6,
                                  "launches_per_step": v["launches"] / ((args.steps + 1) // 2)}
                out["gemm_sites"] = sites
            hbm = {}
            aux = aux_timer.summarize()  # one warm-up step, not part of the timed region
            for k in ("cast_amax", "mxfp8_quantize"):
                if k in aux and aux[k]["seconds"] > 0:
                    hbm[k] = {"GB/s": aux[k]["bytes"] / aux[k]["seconds"] / 1e9, "ms_per_step": aux[k]["seconds"] * 1e3,
                              "launches_per_step": aux[k]["launches"]}
            for k in ("attn_fwd", "attn_bwd"):
                if k in aux and aux[k]["seconds"] > 0:
                    hbm[k] = {"TFLOP/s": aux[k]["work"] / aux[k]["seconds"] / 1e12, "ms_per_step": aux[k]["seconds"] * 1e3,
                              "launches_per_step": aux[k]["launches"]}
            out["hbm_kernels"] = hbm
            out["hbm_kernels_note"] = "cast / quantise / attention kernels bracketed during the last warm-up step only"
        if world == 1 and not args.no_cpu_baseline:
            del model, opt
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.model)
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
