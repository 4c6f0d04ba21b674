"""Time ClippedAdamW.step() on the Llama-3.2-3B parameter set: plain multi-tensor update vs update + FP8 weight casts
(mi_adamw_cast_bf16_multi), with the FP8 outputs masked off one by one (MI_DEBUG_SINK = full | noT | none)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import train

dev = torch.device("cuda:0")
cfg = train.TrainingConfig(model_name="llama-3.2-3b", batch_size=2, max_seq_length=128, mixed_precision="fp8", fp8_scenario="default", use_te=True)
model = train.prepare_model(train.create_model(cfg, dev), cfg)
opt, sched = train.create_optimizer(model, cfg)
model.train()
batch = train.synthetic_batch(cfg, model.config.vocab_size, dev)
for _ in range(2):
    train.train_step(model, batch, opt, sched, cfg)
params = [p for p in model.parameters() if p.requires_grad]
n = sum(p.numel() for p in params)


def time_step(tag, iters=6):
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        p.grad.normal_(0, 1e-3)
    opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{tag:28s} {ms:7.2f} ms per step  ({n/1e9:.2f} B params: {n*16/ms/1e9:6.2f} TB/s at 16 B/param, {n*14/ms/1e9:6.2f} at 14)", flush=True)


for mode in ("full", "noT", "none"):
    os.environ["MI_DEBUG_SINK"] = mode
    opt._plans.clear()
    time_step(f"adamw_cast sinks={mode}")
os.environ.pop("MI_DEBUG_SINK")
os.environ["LLM_FP8_AMD_NO_OPT_WCAST"] = "1"
opt._plans.clear()
time_step("plain adamw_multi")
