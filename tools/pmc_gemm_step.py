"""The FP8 GEMM launch mix of one Llama-3.2-3B training step, once per launch kind, for rocprofv3 --pmc passes (FETCH_SIZE /
WRITE_SIZE / SQ counters in SEPARATE runs): every decoder site's forward GEMM through the default algo (auto: the four-wave or
the eight-wave persistent kernel), its backward as the step launches it (ONE grouped dgrad + wgrad launch where
ops.grouped_gemm_plan groups, else two launches), and the lm_head (N = 128 256) the same way.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_gemm_step.py 3 <manifest.json>

Writes the manifest (launch order: tag as bench.py's KernelTimer names it, kind, algorithmic bytes) for tools/pmc_postprocess_step.py."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops  # noqa: E402

SITES = {"qkv": (5120, 3072), "o": (3072, 3072), "fc1": (16384, 3072), "fc2": (3072, 8192), "lm_head": (128256, 3072)}
M = 8192


def rand_fp8(shape, dev, g):
    t = torch.randint(0, 256, shape, generator=g, device=dev, dtype=torch.uint8)
    t[(t & 0x7F) >= 0x78] &= 0x3F
    return t


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    manifest_path = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    one = torch.ones(1, device=dev)
    manifest = []
    for name, (N, K) in SITES.items():
        x8, w8 = rand_fp8((M, K), dev, g), rand_fp8((N, K), dev, g)
        y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for _ in range(reps):
            ops.gemm_fp8(x8, w8, one, one, 0, 0, out=y)
        manifest.append({"site": f"{name} fprop", "tag": f"{M}x{N}x{K}", "launches": 1, "reps": reps,
                         "algorithmic_bytes": M * K + N * K + 2 * M * N})
        del y
        g8, w8t = rand_fp8((M, N), dev, g), rand_fp8((K, N), dev, g)
        g8t, x8t = rand_fp8((N, M), dev, g), rand_fp8((K, M), dev, g)
        dx = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
        dw = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
        shapes = ((M, K, N), (N, K, M))
        cfg = ops.grouped_gemm_plan(shapes) if ops.grouped_gemm_ok(shapes) else -1
        probs = [(g8, w8t, one, one, dx), (g8t, x8t, one, one, dw)]
        for _ in range(reps):
            if cfg >= 0:
                ops.gemm_fp8_grouped(probs, 0, 0, tile_cfg=cfg)
            else:
                ops.gemm_fp8(g8, w8t, one, one, 0, 0, out=dx)
                ops.gemm_fp8(g8t, x8t, one, one, 0, 0, out=dw)
        alg = [M * N + K * N + 2 * M * K, N * M + K * M + 2 * N * K]
        if cfg >= 0:
            manifest.append({"site": f"{name} dgrad+wgrad (grouped, tile cfg {cfg})", "tag": f"{M}x{K}x{N}+{N}x{K}x{M}", "launches": 1,
                             "reps": reps, "algorithmic_bytes": sum(alg)})
        else:
            manifest.append({"site": f"{name} dgrad, wgrad (two launches)", "tag": [f"{M}x{K}x{N}", f"{N}x{K}x{M}"], "launches": 2,
                             "reps": reps, "algorithmic_bytes": alg})
        del g8, w8t, g8t, x8t, dx, dw, x8, w8
        torch.cuda.empty_cache()
    torch.cuda.synchronize()
    if manifest_path:
        json.dump(manifest, open(manifest_path, "w"), indent=1)
    print("done", len(manifest))


if __name__ == "__main__":
    main()
