"""FETCH_SIZE A/B of the K-serpentine timing build (algo 30: odd tiles walk K downwards) against the default (algo 4):
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/pmc_gemm_serp.py
Each shape: 3 launches of algo 4, then 3 of algo 30 (the last of each triple is the one to read)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import _lib as _mi_lib
_mi_lib.use_lab_library()  # algo 30 is a lab build
from llm_fp8_amd.pytorch import ops
from tools.bench_kernels import rand_fp8

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
one = torch.ones(1, device=dev)
for (m, n, k) in ((8192, 16384, 3072), (8192, 8192, 3072), (8192, 5120, 3072), (16384, 3072, 8192), (8192, 8192, 8192)):
    a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    for algo in (4, 30):
        for _ in range(3):
            ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=algo)
    r4 = ops.gemm_fp8(a, b, one, one, 0, 0, algo=4).float()
    r30 = ops.gemm_fp8(a, b, one, one, 0, 0, algo=30).float()
    print(f"{m}x{n}x{k}: max |algo30 - algo4| / rms = {((r30 - r4).abs().max() / r4.pow(2).mean().sqrt()).item():.2e}", flush=True)
torch.cuda.synchronize()
print("done")
