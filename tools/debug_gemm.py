"""Debug helper (GPU box): error pattern of a GEMM algo vs the generic kernel on integer data."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops
from oracle import fp8_oracle as O

dev = torch.device("cuda:0")
M, N, K = [int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (256, 256, 256))]
algos = [int(x) for x in sys.argv[1].split(",")]
rng = np.random.default_rng(0)
a = rng.integers(-3, 4, size=(M, K)).astype(np.float32)
b = rng.integers(-3, 4, size=(N, K)).astype(np.float32)
a8 = torch.from_numpy(O.fp8_encode_sat(a, 0)).to(dev)
b8 = torch.from_numpy(O.fp8_encode_sat(b, 0)).to(dev)
one = torch.ones(1, device=dev)
ref = torch.from_numpy(a @ b.T).to(dev)
for algo in algos:
    for rep in range(3):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm_fp8(a8, b8, one, one, 0, 0, out=out, algo=algo)
        torch.cuda.synchronize()
        bad = ~(out.float() == ref.to(torch.bfloat16).float())
        nb = int(bad.sum())
        print(f"algo {algo} rep {rep}: {nb} bad of {M*N}")
        if nb:
            r, c = torch.nonzero(bad, as_tuple=True)
            print("  rows%16 hist:", torch.bincount(r % 16, minlength=16).tolist())
            print("  rows//16%16 hist:", torch.bincount((r // 16) % 16, minlength=16).tolist())
            print("  cols%64 hist:", torch.bincount(c % 64, minlength=64).tolist())
            print("  cols//64%4 hist:", torch.bincount((c // 64) % 4, minlength=4).tolist())
            print("  nan count:", int(torch.isnan(out.float()).sum()), " first bad:", [(int(r[i]), int(c[i]), float(out[r[i], c[i]]), float(ref[r[i], c[i]])) for i in range(min(6, nb))])
