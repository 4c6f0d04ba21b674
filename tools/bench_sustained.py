"""Sustained (power-limited) rate of the FP8 GEMM by variant and by operand content: each candidate runs back to back for `secs`
seconds, the rate is taken over the second half (the clock has settled by then).  Lab library (timing builds).

    python tools/bench_sustained.py [secs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import _lib  # noqa: E402

_lib.use_lab_library()
from llm_fp8_amd.pytorch import ops  # noqa: E402
from tools.bench_kernels import rand_fp8  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    one = torch.ones(1, device=dev)
    m, n, k = [int(x) for x in os.environ.get("SUSTAINED_SHAPE", "8192,16384,3072").split(",")]
    data = {
        "random bytes": (rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)),
        "gaussian, quantised as in training (amax -> 448)": tuple(
            (lambda t: (t * (448.0 / t.abs().max())).to(torch.float8_e4m3fn).view(torch.uint8))(torch.randn(s, device=dev, generator=g))
            for s in ((m, k), (n, k))),
        "zeros": (torch.zeros((m, k), dtype=torch.uint8, device=dev), torch.zeros((n, k), dtype=torch.uint8, device=dev)),
    }
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    variants = [("eight-wave persistent (algo 4)", 4), ("four-wave persistent (algo 9)", 9), ("algo 4, no stores (15)", 15),
                ("algo 4, every tile reads panel (0,0): no fabric traffic (20)", 20), ("algo 9, no epilogue (12)", 12),
                ("algo 4 with plain (write-back) stores (17)", 17), ("algo 4 with nt stores (25)", 25), ("algo 4 with sc1 nt stores (26)", 26),
                ("algo 9 with plain stores (70)", 70), ("algo 9 with nt stores (71)", 71), ("algo 9 with sc1 nt stores (72)", 72)]
    flop = 2.0 * m * n * k
    only = [int(x) for x in os.environ.get("SUSTAINED_ALGOS", "").split(",") if x]
    for dname, (a, b) in data.items():
        if only and dname.startswith("zeros"):
            continue
        for vname, algo in variants:
            if only and algo not in only:
                continue
            if not only and (algo in (17, 25, 26, 70, 71, 72) or (dname != "random bytes" and algo not in (4, 9))):
                continue
            def launch():
                ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=algo)
            for _ in range(20):
                launch()
            torch.cuda.synchronize()
            t_end = time.perf_counter() + secs
            marks = []
            while time.perf_counter() < t_end:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50):
                    launch()
                e.record()
                marks.append((s, e))
                torch.cuda.synchronize()
            ts = [s.elapsed_time(e) / 50 * 1e-3 for s, e in marks]
            first, second = ts[: len(ts) // 2], ts[len(ts) // 2:]
            f = lambda v: flop / (sum(v) / len(v)) / 1e15
            print(f"sustained {dname[:28]:28s} | {vname:62s} first half {f(first):.3f}  second half {f(second):.3f} PFLOP/s ({f(second)/5:.3f} of 5 PF)", flush=True)
            time.sleep(1.0)


if __name__ == "__main__":
    main()
