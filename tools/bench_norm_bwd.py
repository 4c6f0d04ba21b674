"""RMSNorm backward: time vs number of dgamma partial blocks (interleaved A/B timing)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops

def main():
    dev = torch.device("cuda:0")
    for R, C in ((8192, 3072), (8192, 2048), (6144, 4096)):
        dy = torch.randn(R, C, device=dev, dtype=torch.bfloat16)
        x = torch.randn(R, C, device=dev, dtype=torch.bfloat16)
        dres = torch.randn(R, C, device=dev, dtype=torch.bfloat16)
        g = torch.ones(C, device=dev, dtype=torch.bfloat16)
        rstd = ops.rmsnorm_stats(x, 1e-5)
        cands = (256, 512, 1024, 2048)
        ev = {n: [] for n in cands}
        for rep in range(30):
            for n in cands:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.rmsnorm_bwd(dy, x, rstd, g, dres=dres, n_partials=n, dgamma_dtype=torch.bfloat16)
                e.record()
                ev[n].append((s, e))
        torch.cuda.synchronize()
        gb = 4 * R * C * 2 / 1e9
        print(f"{R}x{C}: " + "  ".join(f"n={n}: {sorted(a.elapsed_time(b) for a, b in ev[n][5:])[12] * 1e3:6.1f} us" for n in cands), f"  ({gb * 1e3:.0f} MB)")

main()
