"""Attention core timings on the device: hand-written HIP kernels vs torch SDPA (fwd and fwd+bwd)."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops

dev = torch.device("cuda:0")


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, S, H, G, D) in [(16, 512, 24, 8, 128), (12, 512, 32, 8, 128), (4, 2048, 24, 8, 128), (16, 512, 32, 8, 64)]:
    q, k, v = (torch.randn(B, S, n, D, device=dev, dtype=torch.bfloat16) for n in (H, G, G))
    scale = D ** -0.5
    fl = 4.0 * B * H * S * S * D * 0.5
    t_hip = timeit(lambda: ops.attn_fwd(q, k, v, scale, True))
    qs, ks, vs = (t.transpose(1, 2) for t in (q, k, v))
    t_sdpa = timeit(lambda: F.scaled_dot_product_attention(qs, ks, vs, is_causal=True, enable_gqa=True))
    line = f"B{B} S{S} H{H} G{G} D{D}: fwd hip {t_hip:8.1f} us ({fl / t_hip / 1e6:7.1f} TF/s)   sdpa {t_sdpa:8.1f} us ({fl / t_sdpa / 1e6:7.1f} TF/s)"
    if hasattr(ops, "attn_bwd"):
        o, lse = ops.attn_fwd(q, k, v, scale, True)
        do = torch.randn_like(o)
        t_b = timeit(lambda: ops.attn_bwd(do, q, k, v, o, lse, scale, True))
        qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (qs, ks, vs))
        def sd():
            out = F.scaled_dot_product_attention(qr, kr, vr, is_causal=True, enable_gqa=True)
            out.backward(do.transpose(1, 2))
        t_sb = timeit(sd)
        line += f"   bwd hip {t_b:8.1f} us ({2.5 * fl / t_b / 1e6:7.1f} TF/s)   sdpa fwd+bwd {t_sb:8.1f} us"
    print(line, flush=True)
