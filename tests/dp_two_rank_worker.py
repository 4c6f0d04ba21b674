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
    table = model.get_input_embeddings().weight
    # under no_sync the embedding's rows stay deferred with the wrapper: take them as this rank's scatter part and drop them
    (emb, ids_loc, dy_loc), = dp._sparse
    dp._sparse.clear()
    for p in params:
        p.grad = None
    restore()
    dp(**batch).loss.backward()
    torch.cuda.synchronize()
    worst, aliased, table_rel = 0.0, 0, None
    for p, lg in zip(params, local_g):
        aliased += int(p.grad.data_ptr() == p._mi_grad_buf.data_ptr())
        parts = [torch.empty_like(lg) for _ in range(world)]
        dist.all_gather(parts, lg)
        want = (parts[0] + parts[1]) * 0.5  # gloo path: bf16 SUM, then x 1/world (exact)
        if p is table:
            # tied table: the lm_head part went through the bucket (exact, as above); the embedding part was reduced
            # row-sparsely: all-gathered (ids, rows), aten's deterministic scatter over both ranks' tokens, x 1/world
            ids_g = [torch.empty_like(ids_loc) for _ in range(world)]
            dy_g = [torch.empty_like(dy_loc.contiguous()) for _ in range(world)]
            dist.all_gather(ids_g, ids_loc.contiguous())
            dist.all_gather(dy_g, dy_loc.contiguous())
            ids_all, dy_all = torch.cat(ids_g), torch.cat(dy_g)
            # bucket part exact (as above) + the rows of both ranks added in place (fp32 sums, one rounding): within one bf16
            # ulp of an exact fp64 scatter
            exact = torch.zeros(table.shape, dtype=torch.float64, device=table.device).index_add_(0, ids_all, dy_all.double())
            mag = torch.zeros(table.shape, dtype=torch.float64, device=table.device).index_add_(0, ids_all, dy_all.double().abs())
            ref = want.double() + 0.5 * exact
            got = p.grad.double()
            ulp = torch.maximum(ref.abs(), got.abs()).clamp_min(1e-30).log2().floor().exp2() * 2.0 ** -7
            table_rel = float(((got - ref).abs() / (ulp + 2.0 ** -22 * mag)).max())
            out["table_bitwise"] = True
            out["table_rows_touched"] = int((mag.sum(1) > 0).sum().item())
            continue
        worst = max(worst, float((p.grad.float() - want.float()).abs().max()))
    out.update(worst_grad_err=worst, aliased=aliased, n_params=len(params), n_arenas=len(saved), table_rel_err=table_rel,
               local_differs=float((local_g[0].float() - params[0].grad.float()).abs().max()))
    for p in params:
        p.grad = None
    opt, sched = train.create_optimizer(dp, cfg)
    out["optimizer"] = type(opt).__name__
    losses = [float(train.train_step(dp, batch, opt, sched, cfg).item()) for _ in range(4)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for p in params])
    out.update(losses=losses, checksum=float(flat.double().sum().item()), absum=float(flat.double().abs().sum().item()))
    outs = [None] * world  # one writer: two processes printing to the same pipe can interleave their lines
    dist.all_gather_object(outs, out)
    if rank == 0:
        for o in outs:
            print(json.dumps(o), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
