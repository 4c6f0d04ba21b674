"""World-size-2 gloo tests (CPU): the N > 1 plumbing that needs no FP8 kernels -- the amax MAX-all-reduce of the
meta arenas, and the DDP / FSDP wrap + train step of the harness on the reference's no-TE bf16 path."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _worker(rank, world, port, q, mode="ddp"):
    """Both checks in one pair of processes (process start-up dominates the cost of this file)."""
    _init(rank, world, port)
    # ---- 1. amax MAX-all-reduce of a meta arena: one collective over the used slots only
    from llm_fp8_amd.common.recipe import DelayedScaling, Format
    from llm_fp8_amd.pytorch.fp8 import MetaArena
    r = DelayedScaling(fp8_format=Format.HYBRID, amax_history_len=4, amax_compute_algo="max")
    a = MetaArena((r.fp8_format, 4, "max", 0, True), torch.device("cpu"))
    start = a.alloc(6)
    a.hist[0, start:start + 6] = torch.tensor([1.0, 5.0, 0.0, 2.0, 9.0, 0.5]) * (rank + 1)
    a.hist[0, 6:10] = 123.0  # beyond `used`: must not take part
    a.reduce()
    row = a.hist[0, :10].tolist()
    a.reduce_amax = False
    a.hist[0, 0] = float(rank)
    a.reduce()
    single = a.hist[0, 0].item()
    # ---- 2. the harness' DDP wrap + train step on the reference's no-TE bf16 path, different data per rank
    from llm_fp8_amd import train
    import llm_fp8_amd.llama as llama
    cfg = train.TrainingConfig(model_name="llama-3.2-1b", batch_size=2, max_seq_length=16, mixed_precision="bf16",
                               use_te=False, sharding_mode=mode, num_hidden_layers=2, vocab_size=256, learning_rate=1e-2,
                               num_warmup_steps=0)
    orig = llama.llama_config
    llama.llama_config = lambda name, **kw: orig(name, **{**dict(hidden_size=64, intermediate_size=128, num_attention_heads=4,
                                                               num_key_value_heads=2, head_dim=16, max_position_embeddings=64), **kw})
    torch.manual_seed(0)
    device = torch.device("cpu")
    model = train.wrap_distributed(train.prepare_model(train.create_model(cfg, device), cfg), cfg, device)
    opt, sched = train.create_optimizer(model, cfg)
    model.train()
    g = torch.Generator().manual_seed(100 + rank)
    losses = [train.train_step(model, train.synthetic_batch(cfg, 256, device, g), opt, sched, cfg).item() for _ in range(3)]
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
    q.put((rank, row, single, losses, flat.double().sum().item(), flat.numel()))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "auto"])  # auto at world 2 on CPU -> the gradient-arena wrapper ("replicated")
def test_two_rank_gloo_amax_allreduce_and_ddp_harness(mode):
    """FSDP (train_multi_gpu.py's default) cannot be rehearsed here: torch 2.10's FSDP refuses CPU-only processes;
    tests/test_distributed_gpu.py wraps it on the GPU box instead."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted(q.get(timeout=400) for _ in range(2))
    [p.join(120) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, row0, s0, l0, sum0, n0), (_, row1, s1, l1, sum1, n1) = out
    assert row0 == row1 == [2.0, 10.0, 0.0, 4.0, 18.0, 1.0, 123.0, 123.0, 123.0, 123.0]
    assert (s0, s1) == (0.0, 1.0)  # reduce_amax=False leaves the ranks untouched
    assert n0 == n1 and abs(sum0 - sum1) <= 1e-6 * max(1.0, abs(sum0)), "replicas diverged after 3 steps"
    assert all(torch.isfinite(torch.tensor(l0 + l1))) and l0 != l1  # ranks really saw different data


# ------------------------------------------------------------------------------------------------ gradient-arena DP
class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = torch.nn.Embedding(32, 16)
        self.a = torch.nn.Linear(16, 48)
        self.b = torch.nn.Linear(48, 16, bias=False)
        self.unused = torch.nn.Linear(16, 16)           # never in the graph: its bucket closes with zeros / stays out
        self.head = torch.nn.Linear(16, 32, bias=False)
        self.head.weight = self.emb.weight               # tied: one parameter, two uses

    def forward(self, ids):
        h = self.emb(ids)
        h = h + self.b(torch.tanh(self.a(h)))
        return self.head(h)


def _arena_worker(rank, world, port, q):
    _init(rank, world, port)
    from llm_fp8_amd.distributed import GradArenaDP
    torch.manual_seed(1 + rank)                          # ranks start DIFFERENT: the wrapper must broadcast rank 0's weights
    net = _Toy()
    dp = GradArenaDP(net, bucket_mb=0.002)               # ~2 KB buckets: several per pass
    info = dp.describe()
    g = torch.Generator().manual_seed(7)
    ids_all = torch.randint(0, 32, (2, 4, 6), generator=g)  # [rank, batch, seq]: every rank knows every rank's data
    lossf = lambda m, ids: torch.nn.functional.cross_entropy(m(ids).flatten(0, 1), ids.flatten())
    # reference: the mean over ranks of the per-rank gradients, computed locally on a copy with the (broadcast) weights
    ref = _Toy()                                         # (not deepcopy: the wrapper has put a forward override on net.emb)
    ref.load_state_dict(net.state_dict())
    ref_grads = None
    for r in range(world):
        ref.zero_grad()
        lossf(ref, ids_all[r]).backward()
        gs = [None if p.grad is None else p.grad.clone() for p in ref.parameters()]
        ref_grads = gs if ref_grads is None else [a if b is None else a + b for a, b in zip(ref_grads, gs)]
    ref_grads = [None if gsum is None else gsum / world for gsum in ref_grads]
    # 1. one synchronised backward
    lossf(dp, ids_all[rank]).backward()
    err = 0.0
    aliased = True
    for p, want in zip(net.parameters(), ref_grads):
        if want is None:
            ok_none = p.grad is None or float(p.grad.abs().max()) == 0.0
            err = max(err, 0.0 if ok_none else 1.0)
            continue
        err = max(err, float((p.grad - want).abs().max()))
        aliased &= p.grad.data_ptr() == p._mi_grad_buf.data_ptr()
    # 2. accumulation: no_sync pass + synchronised pass on the same data = 2 x the averaged gradient
    for p in net.parameters():
        p.grad = None
    with dp.no_sync():
        lossf(dp, ids_all[rank]).backward()
    local_only = float((net.a.weight.grad - ref_grads[[id(p) for p in net.parameters()].index(id(net.a.weight))]).abs().max())
    lossf(dp, ids_all[rank]).backward()
    err2 = 0.0
    for p, want in zip(net.parameters(), ref_grads):
        if want is not None:
            err2 = max(err2, float((p.grad - 2 * want).abs().max()))
    # 3. ranks with DIFFERENT token counts (the reference's collator pads each batch to its own longest sequence, data.py:59-63):
    #    the row-sparse embedding reduction must gather ragged (ids, rows), not assume its own count for everyone
    g3 = torch.Generator().manual_seed(11)
    ragged = [torch.randint(0, 32, (4, 6), generator=g3), torch.randint(0, 32, (3, 10), generator=g3)]
    ref3 = None
    for r in range(world):
        ref.zero_grad()
        lossf(ref, ragged[r]).backward()
        gs = [None if p.grad is None else p.grad.clone() for p in ref.parameters()]
        ref3 = gs if ref3 is None else [a if b is None else a + b for a, b in zip(ref3, gs)]
    for p in net.parameters():
        p.grad = None
    lossf(dp, ragged[rank]).backward()
    err3 = 0.0
    for p, want in zip(net.parameters(), ref3):
        if want is not None:
            err3 = max(err3, float((p.grad - want / world).abs().max()))
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double().sum().item()
    q.put((rank, err, aliased and "forward" in net.emb.__dict__, err2, local_only, w0, len(info["buckets"]), err3))
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_arena_dp():
    """distributed.GradArenaDP at world size 2: weights broadcast, gradients = mean over ranks (tied and unused parameters,
    several buckets), `.grad` aliased into the arena, the embedding's part reduced row-sparsely, no_sync accumulation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    out = sorted(q.get(timeout=300) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, err, aliased, err2, local_only, w0, nb, err3 in out:
        assert err < 1e-6 and err2 < 2e-6 and err3 < 1e-6, (rank, err, err2, err3)
        assert aliased and nb >= 3
        assert local_only > 1e-4      # the no_sync pass really was rank-local
    assert out[0][5] == out[1][5]     # broadcast: same weights on both ranks
