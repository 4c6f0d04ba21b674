"""Worker of tests/test_distributed_gpu.py::test_fsdp_full_shard_matches_the_unwrapped_run: launched by torch.distributed.run.
The reference's FSDP FULL_SHARD wrap (train_multi_gpu.py:381-460, reproduced by train.wrap_distributed mode "fsdp_full") must not
change the arithmetic: from the same weights and the same fresh FP8 state, one forward + backward under the wrapper gives the
loss and the per-parameter gradients of the unwrapped model -- at world size 1 bit for bit, at world size 2 the mean over the
ranks' unwrapped gradients within the bf16 reduce-scatter's rounding.  Prints one JSON line per rank."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import train  # noqa: E402
from llm_fp8_amd.pytorch.fp8 import FP8GlobalStateManager as G  # noqa: E402


def build(cfg, device, seed):
    torch.manual_seed(seed)
    return train.prepare_model(train.create_model(cfg, device), cfg)


def main():
    scenario = sys.argv[1]
    rank, local, world, device = train.setup_distributed()
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=4, max_seq_length=128, mixed_precision="fp8",
                               fp8_scenario=scenario, use_te=True, sharding_mode="fsdp_full", num_hidden_layers=2,
                               vocab_size=4096, learning_rate=1e-3, num_warmup_steps=0)
    batches = [train.synthetic_batch(cfg, 4096, device, torch.Generator(device=device).manual_seed(50 + r)) for r in range(world)]

    # ---- unwrapped reference: every rank computes every rank's gradient itself, each from a FRESH FP8 state (scale 1, empty
    # history: what the wrapped model's first step sees as well), and averages in fp32
    ref_loss, ref_grads = [], None
    for r in range(world):
        G.reset()
        m = build(cfg, device, 777)
        m.train()
        out = m(**batches[r])
        out.loss.backward()
        ref_loss.append(out.loss.item())
        names = [n for n, p in m.named_parameters() if p.grad is not None]
        gs = {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
        ref_grads = gs if ref_grads is None else {n: ref_grads[n] + gs[n] for n in names}
        del m, out
    ref_grads = {n: g / world for n, g in ref_grads.items()}

    # ---- the same step under FSDP FULL_SHARD
    G.reset()
    model = build(cfg, device, 777)
    fsdp = train.wrap_distributed(model, cfg, device)
    assert type(fsdp).__name__ == "FullyShardedDataParallel", type(fsdp).__name__
    fsdp.train()
    out = fsdp(**batches[rank])
    out.loss.backward()
    loss = out.loss.item()
    from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
    worst_rel, worst_name, exact, n = 0.0, "", 0, 0
    with FSDP.summon_full_params(fsdp, with_grads=True):
        for name, p in fsdp.named_parameters():
            name = name.replace("_fsdp_wrapped_module.", "")
            if p.grad is None or name not in ref_grads:
                continue
            g, want = p.grad.float(), ref_grads[name]
            n += 1
            exact += int(torch.equal(g, want))
            denom = want.abs().max().item() + 1e-30
            rel = (g - want).abs().max().item() / denom
            if rel > worst_rel:
                worst_rel, worst_name = rel, name
    print(json.dumps({"rank": rank, "world": world, "loss": loss, "ref_loss": ref_loss[rank], "params": n, "exact": exact,
                      "worst_rel": worst_rel, "worst_name": worst_name}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
