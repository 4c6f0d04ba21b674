"""The FP8 GEMMs of one Llama-3.2-3B training step and NOTHING else, in the step's order and with the step's operand reuse:
28 layers of (qkv, o, fc1, fc2) forward GEMMs, the lm_head, then the backward pairs in reverse order -- every layer with its OWN
weights (3.2 GB of FP8 copies: cold like in the step), activations rotating over a few buffers, launched as the step launches them
(ops.gemm_fp8 default algo; backward pairs through ops.grouped_gemm_choice / two launches).  Prints the FLOP-weighted rate of the
sequence = what bench.py's roofline.achieved would be if the kernels between the GEMMs cost nothing and disturbed nothing.

    python tools/bench_gemm_mix.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops  # noqa: E402
from tools.bench_kernels import rand_fp8  # noqa: E402

SITES = (("qkv", 5120, 3072), ("o", 3072, 3072), ("fc1", 16384, 3072), ("fc2", 3072, 8192))
M, LAYERS, VOCAB, H = 8192, 28, 128256, 3072


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    one = torch.ones(1, device=dev)
    W = [{n: (rand_fp8((N, K), dev, g), rand_fp8((K, N), dev, g)) for n, N, K in SITES} for _ in range(LAYERS)]
    Wh = (rand_fp8((VOCAB, H), dev, g), rand_fp8((H, VOCAB), dev, g))
    act = {}
    for n, N, K in SITES + (("lm_head", VOCAB, H),):
        act[n] = {"x8": [rand_fp8((M, K), dev, g) for _ in range(2)], "x8t": [rand_fp8((K, M), dev, g) for _ in range(2)],
                  "g8": [rand_fp8((M, N), dev, g) for _ in range(2 if N < 100000 else 1)],
                  "g8t": [rand_fp8((N, M), dev, g) for _ in range(2 if N < 100000 else 1)],
                  "y": [torch.empty((M, N), dtype=torch.bfloat16, device=dev) for _ in range(2 if N < 100000 else 1)],
                  "dx": [torch.empty((M, K), dtype=torch.bfloat16, device=dev) for _ in range(2)],
                  "dw": torch.empty((N, K), dtype=torch.bfloat16, device=dev)}
    flops = 0.0

    def pair(n, N, K, w8t, l):
        a = act[n]
        g8, g8t, x8t, dx, dw = a["g8"][l % len(a["g8"])], a["g8t"][l % len(a["g8t"])], a["x8t"][l % 2], a["dx"][l % 2], a["dw"]
        probs = [(g8, w8t, one, one, dx), (g8t, x8t, one, one, dw)]
        cfg = ops.grouped_gemm_choice(probs, 0, 0) if ops.grouped_gemm_ok(((M, K, N), (N, K, M))) else -1
        if cfg >= 0:
            ops.gemm_fp8_grouped(probs, 0, 0, tile_cfg=cfg)
        else:
            ops.gemm_fp8(g8, w8t, one, one, 0, 0, out=dx)
            ops.gemm_fp8(g8t, x8t, one, one, 0, 0, out=dw)

    def step():
        for l in range(LAYERS):
            for n, N, K in SITES:
                ops.gemm_fp8(act[n]["x8"][l % 2], W[l][n][0], one, one, 0, 0, out=act[n]["y"][l % 2])
        ops.gemm_fp8(act["lm_head"]["x8"][0], Wh[0], one, one, 0, 0, out=act["lm_head"]["y"][0])
        pair("lm_head", VOCAB, H, Wh[1], 0)
        for l in reversed(range(LAYERS)):
            for n, N, K in reversed(SITES):
                pair(n, N, K, W[l][n][1], l)

    for n, N, K in SITES:
        flops += LAYERS * 3 * 2.0 * M * N * K
    flops += 3 * 2.0 * M * VOCAB * H
    for _ in range(2):
        step()  # includes the grouped autotune of every pair shape
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        step()
    e.record()
    torch.cuda.synchronize()
    dt = s.elapsed_time(e) * 1e-3 / iters
    print(f"gemm mix: {flops/1e12:.1f} TFLOP per step in {dt*1e3:.2f} ms = {flops/dt/1e15:.3f} PFLOP/s = {flops/dt/5e15:.4f} of 5 PF "
          f"({LAYERS * 4 + 1} forward launches + the backward pairs; host {time.perf_counter() - t0:.2f} s for {iters} steps)", flush=True)


if __name__ == "__main__":
    main()
