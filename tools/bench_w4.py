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
from llm_fp8_amd import _lib as _mi_lib  # noqa: E402
_mi_lib.use_lab_library()  # timing / ablation builds live in tools/bin/libmi_fp8_lab.so (make -C llm_fp8_amd/csrc lab)
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
    ap.add_argument("--kt-algo", type=int, default=73, help="ktstamps build of the lab library")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    one = torch.ones(1, device=dev)
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    shapes = [s for s in shapes_3b() if not args.sites or s[0].split(".")[0] in args.sites.split(",")]
    if "parity" in args.which:
        sc = torch.tensor([0.37], device=dev)
        extra = [("small", 512, 768, 512), ("small2", 256, 256, 512), ("tall", 2048, 256, 1024), ("one", 256, 256, 768),
                 ("nk4x4", 8192, 8192, 512), ("nk6odd", 4096, 4352, 768), ("nk8odd", 4352, 4096, 1024), ("nk10", 2048, 4096, 1280)]
        for name, m, n, k in shapes + extra:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            ref = ops.gemm_fp8(a, b, sc, one, 0, 0, algo=4)
            for al in [int(x) for x in args.parity_algos.split(",")]:
                for fa, fb in (((0, 0),) if al >= 70 else ((0, 0), (1, 0), (0, 1))):  # the lab timing builds are E4M3 x E4M3 only
                    r4 = ref if (fa, fb) == (0, 0) else ops.gemm_fp8(a, b, sc, one, fa, fb, algo=4)
                    got = ops.gemm_fp8(a, b, sc, one, fa, fb, algo=al)
                    torch.cuda.synchronize()
                    ok = torch.equal(r4, got)
                    print(f"parity {name:10s} {m}x{n}x{k} algo {al} fmt {fa}{fb}: {'bitwise = algo 4' if ok else 'MISMATCH max|d| ' + str((r4.float() - got.float()).abs().max().item())}", flush=True)
                    assert ok
    if "bias" in args.which:  # the persistent four-wave kernel with a bias against the eight-wave kernel with the same bias: bitwise, then timing
        for name, m, n, k in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            bias = (torch.randn(n, device=dev, generator=g) * 3).to(torch.bfloat16)
            sc = torch.tensor([0.37], device=dev)
            ref = ops.gemm_fp8(a, b, sc, one, 0, 0, bias=bias, algo=4)
            got = ops.gemm_fp8(a, b, sc, one, 0, 0, bias=bias, algo=9)
            assert torch.equal(ref, got), f"bias: algo 9 differs from algo 4 on {name}: {(ref.float() - got.float()).abs().max().item()}"
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            res = time_interleaved({al: (lambda al=al: ops.gemm_fp8(a, b, one, one, 0, 0, bias=bias, out=out, algo=al)) for al in (4, 9)}, rounds=8, inner=5)
            print(f"bias {name:10s} {m}x{n}x{k}: bitwise ok;  " + "  ".join(f"a{al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:5.0f} TF" for al, t in res.items()), flush=True)
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
    if "cold" in args.which:  # every candidate twice IN ONE interleaved timing: "w" re-launches on one operand set (everything stays in
        # the 256-MiB Infinity Cache, as in "timing"), "c" rotates over enough operand / output sets that no launch finds anything
        # cached -- the condition of the training step for weights and saved activations.  (Timed in separate blocks the two drift
        # apart with the clock: compare only inside one line.)
        algos = [int(x) for x in args.algos.split(",")]
        for name, m, n, k in shapes:
            per = m * k + n * k + 2 * m * n
            nset = max(3, int(6e8 // per) + 1)
            sets = [(rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g), torch.empty((m, n), dtype=torch.bfloat16, device=dev)) for _ in range(nset)]
            ctr = {}
            def make(al, cold):
                key = (al, cold)
                ctr[key] = 0
                def f():
                    a, b, out = sets[ctr[key] % nset if cold else 0]
                    ctr[key] += 1
                    ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=al)
                return f
            res = time_interleaved({f"a{al}{'c' if cold else 'w'}": make(al, cold) for al in algos for cold in (False, True)}, rounds=8, inner=nset)
            print(f"cold {name:10s} {m}x{n}x{k} ({nset} operand sets): " + "  ".join(
                f"a{al}: warm {res[f'a{al}w']*1e6:7.1f} cold {res[f'a{al}c']*1e6:7.1f} us (x{res[f'a{al}c']/res[f'a{al}w']:.3f})" for al in algos), flush=True)
            del sets
            torch.cuda.empty_cache()
    if "sched" in args.which:  # phase schedules of the one-tile-per-workgroup kernel (mi_gemm_w4.hip w4::sched_*): 50 + 4 S + kind
        scheds = {"S0": (6, 7, 8), "S1": (54, 55, 56), "S3": (62, 63, 64), "S4": (66, 67, 68)}
        import time
        for name, m, n, k in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            ref = ops.gemm_fp8(a, b, one, one, 0, 0, algo=4)
            for sn, (prod, nost, stamp) in scheds.items():
                assert torch.equal(ref, ops.gemm_fp8(a, b, one, one, 0, 0, algo=prod)), f"schedule {sn} differs from algo 4 on {name}"
            res = time_interleaved({sn: (lambda al=v[1]: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=al)) for sn, v in scheds.items()}, rounds=8, inner=5)
            line = f"sched {name:10s} {m}x{n}x{k} (no stores): " + "  ".join(f"{sn}: {t*1e6:7.1f} us" for sn, t in res.items())
            cyc = {}
            for sn, (prod, nost, stamp) in scheds.items():
                dbg = torch.zeros(((m // 256) * (n // 256), 4), dtype=torch.int64, device=dev)
                for _ in range(60):
                    ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=nost)
                rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), dbg.data_ptr(), m, n, k, k, k, n, 0, 0, 0, stamp, st)
                assert rc == 0, lib.mi_last_error()
                torch.cuda.synchronize()
                d = dbg.cpu().double()
                cyc[sn] = float((d[:, 0] / d[:, 2]).median())
            print(line + "   loop cycles/K-tile: " + "  ".join(f"{sn}: {c:6.0f}" for sn, c in cyc.items()), flush=True)
    if "ktstamps" in args.which:  # per-K-tile timeline of workgroup 0 of the persistent four-wave kernel (algo 73)
        for name, m, n, k in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            dbg = torch.zeros(1024, dtype=torch.int64, device=dev)
            for _ in range(100):
                ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=9)
            for label in ("warm", "cold"):
                if label == "cold":  # fresh operands and output, then 1 GB of unrelated traffic: nothing of this launch is in L2 / the Infinity Cache
                    a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
                    junk.fill_(1.0)
                    junk.mul_(2.0)
                    del junk
                dbg.zero_()
                rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), dbg.data_ptr(), m, n, k, k, k, n, 0, 0, 0, args.kt_algo, st)
                assert rc == 0, lib.mi_last_error()
                torch.cuda.synchronize()
                t = dbg.cpu().numpy()
                t = t[t > 0]
                d = (t[1:] - t[:-1])
                nk = k // 128
                print(f"ktstamps(algo {args.kt_algo}) {label} {name} {m}x{n}x{k}: nk {nk}, {len(d)} K-tiles, total {int(d.sum())} cycles; cycles per K-tile (rows = tiles):", flush=True)
                for ti in range(0, len(d), nk):
                    print("   " + " ".join(f"{int(x):5d}" for x in d[ti:ti + nk]), flush=True)
    if "clock" in args.which:
        import time
        for name, m, n, k in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            for label, heat, stamped, width in (("8ph ", 13, 14, 2), ("w4  ", 7, 8, 4), ("p8  ", 4, 21, 4), ("w4p ", 9, 11, 4)):
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
                cyc = d[:, 0] / (d[:, 2] if width == 4 else (k / 128))
                print(f"clock {name:10s} {label}: clock median {float(clk.median()):7.1f} MHz  loop cycles/K-tile median {float(cyc.median()):7.1f} "
                      f"(min {float(cyc.min()):.0f} max {float(cyc.max()):.0f}; 2048 = MFMA-bound)", flush=True)


if __name__ == "__main__":
    main()
