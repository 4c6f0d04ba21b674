"""One forward + backward of the attention kernels at the headline shape (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops
dev = torch.device("cuda:0")
B, S, H, G, D = 16, 512, 24, 8, 128
q, k, v = (torch.randn(B, S, n, D, device=dev, dtype=torch.bfloat16) for n in (H, G, G))
do = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    o, lse = ops.attn_fwd(q, k, v, D ** -0.5, True)
    ops.attn_bwd(do, q, k, v, o, lse, D ** -0.5, True)
torch.cuda.synchronize()
print("done")
