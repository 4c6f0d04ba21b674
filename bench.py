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


def cpu_baseline(bench_model: str, seq: int = 128, timed_steps: int = 3, extrapolate_bench_model: bool = False):
    """SURVEY.md 8d "CPU baseline (same run)": the repo's own no-TE HF bf16 path (train_fp8.py:118-124 model, :270-291 step) on
    the host CPUs -- BASELINE.json config #1: Llama-3.2-1B at FULL depth (16 layers), batch 1 x seq 128, 1 warm-up + 3 timed
    optimiser steps, tokens/s = 128 / step time.  No extrapolation.  Plus the Linear-level micro-baseline: `F.linear` bf16
    fprop + dgrad + wgrad on the CPU for every GEMM site of the benchmarked model at M = 128."""
    from llm_fp8_amd import llama, train
    from transformers.models.llama.modeling_llama import LlamaForCausalLM

    def time_model(model_name, layers, steps):
        cfg = train.TrainingConfig(model_name=model_name, batch_size=1, max_seq_length=seq, mixed_precision="bf16",
                                   use_te=False, num_hidden_layers=layers, num_warmup_steps=0)
        config = llama.llama_config(model_name, **({} if layers is None else {"num_hidden_layers": layers}))
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
        per = []
        for _ in range(steps):
            t0 = time.perf_counter()
            train.train_step(model, batch, opt, sched, cfg)
            per.append(time.perf_counter() - t0)
        del model, opt, sched
        return per

    per = time_model("llama-3.2-1b", None, timed_steps)
    t_step = sum(per) / len(per)
    out = {
        "value": seq / t_step, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": (f"BASELINE.json config #1: HF LlamaForCausalLM bf16 (no TE, sdpa), Llama-3.2-1B at full depth (16 layers), "
                   f"fwd+bwd+clip+AdamW, batch 1 x seq {seq}, 1 warm-up + {timed_steps} timed steps "
                   f"({', '.join(f'{t:.2f}' for t in per)} s); torch.get_num_threads() = {torch.get_num_threads()}, "
                   f"os.cpu_count() = {os.cpu_count()}"),
        "linear_sites": linear_micro_baseline(bench_model),
    }
    if extrapolate_bench_model and bench_model != "llama-3.2-1b":  # optional extra (round 1's figure): the benchmarked model, by depth
        full = llama.llama_config(bench_model).num_hidden_layers
        t2, t4 = (sum(x) / len(x) for x in (time_model(bench_model, 2, 2), time_model(bench_model, 4, 2)))
        per_layer = max((t4 - t2) / 2, 0.0)
        t_full = max(t2 - 2 * per_layer, 0.0) + full * per_layer
        out["extrapolated_bench_model"] = {"tokens/s": seq / t_full, "note": f"{bench_model}: timed at 2 and 4 of {full} layers, linear in depth"}
    return out


def linear_micro_baseline(model_name: str, M: int = 128, iters: int = 3):
    """`F.linear` bf16 forward + dgrad + wgrad on the host CPUs for each GEMM site of `model_name` at M = 128 tokens
    (BASELINE.md section 3 / SURVEY.md 8d): the per-site counterpart of the GPU `gemm_sites` table."""
    import torch.nn.functional as F
    from llm_fp8_amd import llama
    c = llama.llama_config(model_name)
    hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    sites = {"qkv": ((c.num_attention_heads + 2 * c.num_key_value_heads) * hd, c.hidden_size),
             "o_proj": (c.hidden_size, c.num_attention_heads * hd),
             "fc1": (2 * c.intermediate_size, c.hidden_size), "fc2": (c.hidden_size, c.intermediate_size),
             "lm_head": (c.vocab_size, c.hidden_size)}
    res = {}
    g = torch.Generator().manual_seed(0)
    for name, (N, K) in sites.items():
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).requires_grad_(True)
        dy = (torch.randn(M, N, generator=g) / 32).to(torch.bfloat16)
        F.linear(x, w).backward(dy)  # warm-up
        ts = []
        for _ in range(iters):
            x.grad = w.grad = None
            t0 = time.perf_counter()
            F.linear(x, w).backward(dy)
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        res[name] = {"shape_MxNxK": f"{M}x{N}x{K}", "ms_fwd_dgrad_wgrad": t * 1e3, "GFLOP/s": 6.0 * M * N * K / t / 1e9}
    return res


def gemm_clock_probe(sites, top: int = 4):
    """In-kernel clock of the FP8 GEMM (MI355X_MICROARCH.md "DVFS give-back" item 6): mi_gemm_fp8_clock (the production kernel with
    two clock reads around each workgroup's whole tile walk) reports d(s_memtime) / d(s_memrealtime) x 100 MHz; run after >= 1 s of
    back-to-back launches of the production kernel on random FP8 bytes, for the `top` sites by time, time-weighted."""
    from llm_fp8_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.cuda.current_stream().cuda_stream
    one = torch.ones(1, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    tot_w = tot = 0.0
    per_site = {}
    for tag, v in sorted(sites.items(), key=lambda kv: -kv[1]["seconds"])[:top]:
        m, n, k = (int(t) for t in tag.split("+")[0].split("x"))  # a grouped launch "AxBxC+DxExF": probe its first problem
        if (m % 256 and m % 192) or (n % 256 and n % 192) or k % 256 or m * k >= 2 ** 31 or n * k >= 2 ** 31 or m * n * 2 >= 2 ** 31:
            continue
        a = torch.randint(0, 256, (m, k), generator=g, device=dev, dtype=torch.uint8)
        b = torch.randint(0, 256, (n, k), generator=g, device=dev, dtype=torch.uint8)
        a[(a & 0x7F) >= 0x78] &= 0x3F
        b[(b & 0x7F) >= 0x78] &= 0x3F
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        dbg = torch.zeros((256, 4), dtype=torch.int64, device=dev)

        def run(ptr):
            if ptr is None:
                rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), None, m, n, k, k, k, n, 0, 0, 0, 0, st)
            else:
                rc = lib.mi_gemm_fp8_clock(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), m, n, k, k, k, n, 0, ptr, st)
            assert rc == 0, lib.mi_last_error()

        t0 = time.time()
        while time.time() - t0 < 1.0:
            for _ in range(40):
                run(None)
            torch.cuda.synchronize()
        run(dbg.data_ptr())
        torch.cuda.synchronize()
        d = dbg.cpu().double()
        d = d[d[:, 1] > 0]
        if d.numel() == 0:
            continue
        clk = float((d[:, 0] / d[:, 1] * 100.0).median()) / 1e3  # GHz
        per_site[tag] = clk
        tot_w += v["seconds"]
        tot += v["seconds"] * clk
    return (tot / tot_w if tot_w else None), per_site


def pmc_traffic(by_tag):
    """Per-launch fabric bytes of the GEMM launches from the committed rocprofv3 PMC passes (profiles/), averaged over the launch
    mix of this run, TOGETHER with the algorithmic bytes of the same launches: (traffic, algorithmic, coverage, note).  A launch
    tag without a PMC row is left out of BOTH averages; `coverage` = the share of this run's launches that has a row."""
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_gemm_pmc_traffic.json"))
    if not cands:
        return None, None, 0.0, "no PMC file"
    path = os.path.join(ROOT, "profiles", cands[-1])  # newest round
    tab = json.load(open(path))["sites"]
    num = alg = den = tot = 0.0
    for tag, v in by_tag.items():
        tot += v["launches"]
        if tag in tab:
            num += v["launches"] * (tab[tag]["fabric_read_bytes"] + tab[tag]["fabric_write_bytes"])
            alg += v["launches"] * tab[tag]["algorithmic_bytes"]
            den += v["launches"]
    if den == 0:
        return None, None, 0.0, "no measured shape in this run"
    return num / den, alg / den, den / tot, (
        "avg bytes per launch over the launches of this run that have a PMC row (forward sites, grouped dgrad + wgrad launches, lm_head), "
        "rocprofv3 --pmc FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE, separate passes "
        f"(profiles/{cands[-1]}); counts Infinity-Cache hits, i.e. L2-miss traffic; `avg_algorithmic_bytes_per_launch` covers the SAME launches")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="llama-3.2-3b")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--scenario", default="default", choices=["default", "hybrid", "mxfp8"])
    ap.add_argument("--sharding_mode", default="auto", choices=["auto", "replicated", "fsdp_fp8", "fsdp_full", "ddp", "none"],
                    help="BASELINE config #5 (Llama-3.1-8B, FSDP full shard over 8 GPUs) = "
                         "`torchrun --nproc-per-node 8 bench.py --gpus 8 --model llama-3.1-8b --batch 12 --sharding_mode fsdp_fp8`")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then not the headline metric)")
    ap.add_argument("--route", default="fused", choices=["fused", "reference"],
                    help="reference = drive every decoder layer exactly as te_llama.py:76-81 does (public kwargs, plain residual adds)")
    ap.add_argument("--cpu-baseline-extrapolate", action="store_true", help="also time the benchmarked model at 2 and 4 layers on the CPU")
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
    if args.route == "reference":
        from llm_fp8_amd import llama as _llama
        _llama.TELlamaDecoderLayer.route = "reference"
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

    # the HBM-bound kernels (casts, block quantisers) and attention are bracketed during ONE step of the timed region (an odd one:
    # the even steps carry the GEMM events)
    aux_timer = KernelTimer(kinds=("cast_amax", "mxfp8_quantize", "attn_fwd", "attn_bwd"))
    aux_step = 1 if args.steps > 1 else -1
    for i in range(args.warmup):
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
            if i % 4 == 0:  # GEMM launches of every 4th timed step carry HIP events (each event costs ~4.5 us of queue time: ~2.8 ms per bracketed step)
                with timer.install():
                    loss = train.train_step(model, batches[i % 4], opt, sched, cfg)
            elif i == aux_step:
                with aux_timer.install():
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
                       "route": args.route, "global_batch": args.batch * world, "seq_len": args.seq,
                       "parallelism": "single" if resolved_mode == "none" else
                       f"dp{world} {resolved_mode}" + (" (gradient arena, bucketed RCCL all-reduce)" if resolved_mode == "replicated" else
                                                       " (full shard: bf16 master rows, gradients, AdamW moments 1/N per rank; FP8 all-gather)" if resolved_mode == "fsdp_fp8" else ""),
                       "config5_command": "python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8 "
                                          "--model llama-3.1-8b --batch 12 --sharding_mode fsdp_fp8"},
            "final_loss": loss_val,
        }
        if not args.no_kernel_timing:
            summ = timer.summarize()
            kind = "gemm_mxfp8" if args.scenario == "mxfp8" and "gemm_mxfp8" in summ else "gemm_fp8"
            g = summ.get(kind)
            if g and g["seconds"] > 0:
                achieved = g["work"] / g["seconds"] / 1e12
                traffic, alg_bytes, coverage, traffic_note = pmc_traffic(g["by_tag"])
                out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": FP8_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": achieved / FP8_DENSE_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                                   "avg_algorithmic_bytes_per_launch": alg_bytes if alg_bytes is not None else g["bytes"] / g["launches"],
                                   "traffic_launch_coverage": coverage,
                                   "kernel": kind, "launches": g["launches"],
                                   "avg_launch_us": g["seconds"] / g["launches"] * 1e6,
                                   "avg_flop_per_launch": g["work"] / g["launches"],
                                   "gemm_ms_per_step": g["seconds"] / ((args.steps + 3) // 4) * 1e3,
                                   "bracketed_steps": f"every 4th of the {args.steps} timed steps ({(args.steps + 3) // 4} steps, {g['launches']} launches)"}
                sites = {}
                for tag, v in sorted(g["by_tag"].items(), key=lambda kv: -kv[1]["seconds"]):
                    sites[tag] = {"tflops": v["work"] / v["seconds"] / 1e12, "us": v["seconds"] / v["launches"] * 1e6,
                                  "launches_per_step": v["launches"] / ((args.steps + 3) // 4)}
                out["gemm_sites"] = sites
                if world == 1 and kind == "gemm_fp8":
                    try:
                        clk, per_site = gemm_clock_probe(g["by_tag"])
                    except Exception as e:  # a diagnostic must never cost the headline line
                        clk, per_site = None, {"error": str(e)}
                    if clk:
                        out["roofline"]["clock_ghz"] = clk
                        out["roofline"]["frac_at_clock"] = achieved / (FP8_DENSE_PEAK_TFLOPS * clk / 2.4)
                        out["roofline"]["clock_note"] = ("in-kernel clock of the GEMM on random FP8 bytes: mi_gemm_fp8_clock, d(s_memtime)/d(s_memrealtime), "
                                                         "median over workgroups after 1 s of back-to-back launches, time-weighted over the top sites: "
                                                         + ", ".join(f"{t} {c:.2f} GHz" for t, c in per_site.items())
                                                         + "; frac_at_clock = achieved / (5000 TFLOP/s x clock / 2.4 GHz)")
            hbm = {}
            aux = aux_timer.summarize()  # one step of the timed region
            for k in ("cast_amax", "mxfp8_quantize"):
                if k in aux and aux[k]["seconds"] > 0:
                    hbm[k] = {"GB/s": aux[k]["bytes"] / aux[k]["seconds"] / 1e9, "ms_per_step": aux[k]["seconds"] * 1e3,
                              "launches_per_step": aux[k]["launches"]}
            for k in ("attn_fwd", "attn_bwd"):
                if k in aux and aux[k]["seconds"] > 0:
                    hbm[k] = {"TFLOP/s": aux[k]["work"] / aux[k]["seconds"] / 1e12, "ms_per_step": aux[k]["seconds"] * 1e3,
                              "launches_per_step": aux[k]["launches"]}
            out["hbm_kernels"] = hbm
            out["hbm_kernels_note"] = ("cast / quantise / attention kernels bracketed by HIP events during timed step 1 (of the timed region); "
                                       "PMC FETCH_SIZE / WRITE_SIZE of the same kernels: profiles/*_hbm_pmc_traffic.json")
        if world == 1 and not args.no_cpu_baseline:
            del model, opt
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.model, extrapolate_bench_model=args.cpu_baseline_extrapolate)
        print(json.dumps(out), flush=True)
    if dist.is_available() and dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
