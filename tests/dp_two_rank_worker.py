"""Worker of tests/test_distributed_gpu.py::test_two_ranks_share_the_gpu_*: launched by torch.distributed.run with 2 ranks that
share the box's one GPU (gloo backend, CUDA tensors -- RCCL refuses two ranks on one device).  Prints one JSON line per
rank.  Everything a real 2-GPU run does on the FP8 path except the transport: gradient arena written by the wgrad GEMMs,
bucketed all-reduce from the autograd hooks, amax MAX-all-reduce, ClippedAdamW on arena addresses."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import llama, train  # noqa: E402


def main():
    scenario = sys.argv[1]
    rank, local, world, device = train.setup_distributed()
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=4, max_seq_length=128, mixed_precision="fp8",
                               fp8_scenario=scenario, use_te=True, sharding_mode="replicated", num_hidden_layers=2,
                               vocab_size=4096, learning_rate=1e-3, num_warmup_steps=0)
    torch.manual_seed(1234 + rank)  # different weights per rank on purpose: the wrapper broadcasts rank 0's
    model = train.prepare_model(train.create_model(cfg, device), cfg)
    from llm_fp8_amd.distributed import GradArenaDP
    dp = GradArenaDP(model, bucket_mb=8.0)
    params = [p for p in model.parameters() if p.requires_grad]
    out = {"rank": rank, "buckets": len(dp.buckets)}
    gen = torch.Generator(device=device).manual_seed(99 + rank)
    batch = train.synthetic_batch(cfg, 4096, device, gen)
    dp.train()
    # exactness of the reduction: one warm-up pass creates the delayed-scaling arenas; then the same batch from the same
    # FP8 state twice (kernels are bitwise reproducible): pass 1 unreduced (no_sync), pass 2 reduced
    from llm_fp8_amd.pytorch.fp8 import FP8GlobalStateManager as G
    dp(**batch).loss.backward()
    for p in params:
        p.grad = None
    saved = [(a, a.hist.clone(), a.scale.clone(), a.scale_inv.clone()) for a in G._arenas.values()]

    def restore():
        for a, h, sc, si in saved:
            a.hist.copy_(h); a.scale.copy_(sc); a.scale_inv.copy_(si); a._snap = None

    with dp.no_sync():
        dp(**batch).loss.backward()
    local_g = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    restore()
    dp(**batch).loss.backward()
    torch.cuda.synchronize()
    worst, aliased = 0.0, 0
    for p, lg in zip(params, local_g):
        parts = [torch.empty_like(lg) for _ in range(world)]
        dist.all_gather(parts, lg)
        want = (parts[0] + parts[1]) * 0.5  # gloo path: bf16 SUM, then x 1/world (exact)
        worst = max(worst, float((p.grad.float() - want.float()).abs().max()))
        aliased += int(p.grad.data_ptr() == p._mi_grad_buf.data_ptr())
    out.update(worst_grad_err=worst, aliased=aliased, n_params=len(params), n_arenas=len(saved),
               local_differs=float((local_g[0].float() - params[0].grad.float()).abs().max()))
    for p in params:
        p.grad = None
    opt, sched = train.create_optimizer(dp, cfg)
    out["optimizer"] = type(opt).__name__
    losses = [float(train.train_step(dp, batch, opt, sched, cfg).item()) for _ in range(4)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for p in params])
    out.update(losses=losses, checksum=float(flat.double().sum().item()), absum=float(flat.double().abs().sum().item()))
    print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
