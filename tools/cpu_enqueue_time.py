"""How far ahead of the GPU is the host?  Enqueue time of K steps (no sync) vs their GPU completion time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import train

dev = torch.device("cuda:0")
cfg = train.TrainingConfig(model_name="llama-3.2-3b", batch_size=16, max_seq_length=512, mixed_precision="fp8", use_te=True)
torch.manual_seed(0)
model = train.prepare_model(train.create_model(cfg, dev), cfg)
opt, sched = train.create_optimizer(model, cfg)
model.train()
batch = train.synthetic_batch(cfg, model.config.vocab_size, dev)
for _ in range(3):
    train.train_step(model, batch, opt, sched, cfg)
torch.cuda.synchronize()
for K in (1, 2, 4):
    t0 = time.perf_counter()
    for _ in range(K):
        train.train_step(model, batch, opt, sched, cfg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"K={K}: host enqueue {1e3 * (t1 - t0) / K:.1f} ms/step, until GPU done {1e3 * (t2 - t0) / K:.1f} ms/step", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
train.train_step(model, batch, opt, sched, cfg)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
