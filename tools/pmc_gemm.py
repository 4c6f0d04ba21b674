"""Run each FP8 GEMM site of Llama-3.2-3B once per algo (for rocprofv3 --pmc passes: FETCH_SIZE / WRITE_SIZE)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops
from tools.bench_kernels import SHAPES_3B, rand_fp8

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
one = torch.ones(1, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, (M, N, K) in SHAPES_3B.items():
    for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
        a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        for _ in range(reps):
            ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=0)
torch.cuda.synchronize()
print("done")
