"""A/B of the four-wave GEMM (mi_gemm_w4.hip, algos 6/7/8 and the persistent forms) against the eight-wave kernels (GPU box).

  parity : algo 6 (and whatever --algos names) must be BITWISE algo 4 -- same fp32 summation order per output element
  timing : interleaved A/B of main loops with and without epilogue stores
  clock  : in-kernel clock + cycles per K-tile (stamped builds)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import _lib  # noqa: E402
from llm_fp8_amd.pytorch import ops  # noqa: E402
from tools.bench_kernels import rand_fp8, time_interleaved  # noqa: E402

SITES = {"qkv": (5120, 3072), "o": (3072, 3072), "fc1": (16384, 3072), "fc2": (3072, 8192)}


def shapes_3b(M=8192):
    out = []
    for name, (N, K) in SITES.items():
        out += [(f"{name}.fprop", M, N, K), (f"{name}.dgrad", M, K, N), (f"{name}.wgrad", N, K, M)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="parity,timing,clock")
    ap.add_argument("--algos", default="3,6,13,7,4,15", help="timing candidates")
    ap.add_argument("--parity-algos", default="6")
    ap.add_argument("--sites", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    one = torch.ones(1, device=dev)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    shapes = [s for s in shapes_3b() if not args.sites or s[0].split(".")[0] in args.sites.split(",")]
    if "parity" in args.which:
        sc = torch.tensor([0.37], device=dev)
        for name, m, n, k in shapes + [("small", 512, 768, 256), ("small2", 256, 256, 512), ("tall", 2048, 256, 1024)]:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            ref = ops.gemm_fp8(a, b, sc, one, 0, 0, algo=4)
            for al in [int(x) for x in args.parity_algos.split(",")]:
                for fa, fb in ((0, 0), (1, 0), (0, 1)):
                    r4 = ref if (fa, fb) == (0, 0) else ops.gemm_fp8(a, b, sc, one, fa, fb, algo=4)
                    got = ops.gemm_fp8(a, b, sc, one, fa, fb, algo=al)
                    torch.cuda.synchronize()
                    ok = torch.equal(r4, got)
                    print(f"parity {name:10s} {m}x{n}x{k} algo {al} fmt {fa}{fb}: {'bitwise = algo 4' if ok else 'MISMATCH max|d| ' + str((r4.float() - got.float()).abs().max().item())}", flush=True)
                    assert ok
    if "timing" in args.which:
        algos = [int(x) for x in args.algos.split(",")]
        tot = {al: 0.0 for al in algos}
        for name, m, n, k in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            fns = {al: (lambda al=al: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=al)) for al in algos}
            res = time_interleaved(fns, rounds=8, inner=5)
            for al in algos:
                tot[al] += res[al]
            print(f"timing {name:10s} {m}x{n}x{k}: " + "  ".join(f"a{al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:5.0f} TF" for al, t in res.items()), flush=True)
        print("timing total: " + "  ".join(f"a{al}: {t*1e6:8.1f} us" for al, t in tot.items()), flush=True)
    if "clock" in args.which:
        import time
        for name, m, n, k in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            for label, heat, stamped, width in (("8ph", 13, 14, 2), ("w4 ", 7, 8, 4)):
                ntiles = (m // 256) * (n // 256)
                dbg = torch.zeros((ntiles, width), dtype=torch.int64, device=dev)

                def run(algo, bias_ptr):
                    rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), bias_ptr,
                                         m, n, k, k, k, n, 0, 0, 0, algo, st)
                    assert rc == 0, lib.mi_last_error()
                t0 = time.time()
                while time.time() - t0 < 1.5:
                    for _ in range(30):
                        run(heat, None)
                    torch.cuda.synchronize()
                run(stamped, dbg.data_ptr())
                torch.cuda.synchronize()
                d = dbg.cpu().double()
                d = d[d[:, 1] > 0]
                clk = d[:, 0] / d[:, 1] * 100.0
                cyc = d[:, 0] / (k / 128)
                print(f"clock {name:10s} {label}: clock median {float(clk.median()):7.1f} MHz  loop cycles/K-tile median {float(cyc.median()):7.1f} "
                      f"(min {float(cyc.min()):.0f} max {float(cyc.max()):.0f}; 2048 = MFMA-bound)", flush=True)


if __name__ == "__main__":
    main()
