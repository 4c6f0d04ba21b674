"""Which Python frames launch the small torch kernels (fill / copy / add / reduce) inside one training step."""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import train
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
cfg = train.TrainingConfig(model_name="llama-3.2-3b", batch_size=16, max_seq_length=512, mixed_precision="fp8", use_te=True,
                           num_hidden_layers=2)
torch.manual_seed(0)
model = train.prepare_model(train.create_model(cfg, dev), cfg)
opt, sched = train.create_optimizer(model, cfg)
model.train()
batch = train.synthetic_batch(cfg, model.config.vocab_size, dev)
for _ in range(3):
    train.train_step(model, batch, opt, sched, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train.train_step(model, batch, opt, sched, cfg)
    torch.cuda.synchronize()
agg = collections.Counter()
par = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    ks = getattr(ev, "kernels", None) or []
    for k in ks:
        kn = k.name[:110]
        if "at::native" in k.name or "rocclr" in k.name:  # every torch-native kernel of the step
            p = ev.cpu_parent
            chain = []
            while p is not None and len(chain) < 4:
                chain.append(p.name)
                p = p.cpu_parent
            agg[(kn, ev.name, " < ".join(chain))] += 1
            dur[(kn, ev.name, " < ".join(chain))] += k.duration
for key, n in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:40]:
    kn, name, chain = key
    print(f"{n:3d}x {dur[key]:8.1f} us  {kn[:70]} | {name} | {chain}")
