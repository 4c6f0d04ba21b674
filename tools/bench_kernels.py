"""Micro-benchmarks of the HIP kernels (GPU box).  Prints one line per (kernel, shape, algo)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import _lib as _mi_lib  # noqa: E402
_mi_lib.use_lab_library()  # timing / ablation builds live in tools/bin/libmi_fp8_lab.so (make -C llm_fp8_amd/csrc lab)
from llm_fp8_amd.pytorch import ops  # noqa: E402

SHAPES_3B = {"qkv": (8192, 5120, 3072), "o": (8192, 3072, 3072), "fc1": (8192, 16384, 3072), "fc2": (8192, 3072, 8192)}


def time_fn(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def time_interleaved(fns, rounds=12, inner=4, warmup=2):
    """A/B timing that is fair under clock/thermal drift: every round runs each candidate `inner` times, in rotating order."""
    keys = list(fns)
    for k in keys:
        for _ in range(warmup):
            fns[k]()
    torch.cuda.synchronize()
    tot = {k: 0.0 for k in keys}
    for r in range(rounds):
        order = keys[r % len(keys):] + keys[:r % len(keys)]
        for k in order:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(inner):
                fns[k]()
            e.record()
            torch.cuda.synchronize()
            tot[k] += s.elapsed_time(e) * 1e-3
    return {k: tot[k] / (rounds * inner) for k in keys}


def rand_fp8(shape, dev, g):
    t = torch.randint(0, 256, shape, generator=g, device=dev, dtype=torch.uint8)
    t[(t & 0x7F) >= 0x78] &= 0x3F  # keep |v| modest, no NaN
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algos", default="2")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--which", default="gemm,cast,mx")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    one = torch.ones(1, device=dev)
    if "gemm" in args.which:
        for name, (M, N, K) in SHAPES_3B.items():
            for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                for algo in [int(x) for x in args.algos.split(",")]:
                    t = time_fn(lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=algo), args.iters)
                    tf = 2.0 * m * n * k / t / 1e12
                    print(f"gemm {name:4s} {kind:5s} {m:6d}x{n:6d}x{k:6d} algo {algo}: {t*1e6:9.1f} us {tf:8.1f} TFLOP/s "
                          f"({tf/5000*100:5.1f}% of 5 PF)", flush=True)
    if "algos" in args.which:  # interleaved A/B of the persistent (4) and the non-persistent 8-phase kernel (3: the default under torch.distributed)
        for name, (M, N, K) in SHAPES_3B.items():
            for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                fns = {al: (lambda al_: (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=al_)))(al) for al in ([int(x) for x in os.environ.get('ALGOS', '4,5,3').split(',')])}
                res = time_interleaved(fns)
                print(f"algos {name:4s} {kind:5s} {m}x{n}x{k}: " + "  ".join(f"algo {al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:6.0f} TF" for al, t in res.items()), flush=True)
    if "tiles" in args.which:
        for (m, n, k) in ((8192, 3072, 3072), (8192, 3072, 8192), (3072, 8192, 8192), (3072, 3072, 8192), (8192, 3072, 16384),
                          (8192, 6144, 3072), (6144, 6144, 4096), (8192, 3072, 5120), (8192, 5120, 3072), (5120, 3072, 8192),
                          (6144, 4096, 4096), (4096, 4096, 6144), (6144, 4096, 6144), (6144, 4096, 14336), (4096, 14336, 6144), (6144, 14336, 4096),
                          (6144, 28672, 4096), (6144, 4096, 28672), (28672, 4096, 6144), (8192, 2048, 2048), (8192, 3072, 2048), (3072, 2048, 8192), (8192, 2048, 8192), (2048, 8192, 8192)):
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            fns = {}
            for algo in (40, 41, 42, 43, 44):
                bm, bn = {40: (256, 256), 41: (256, 192), 42: (192, 256), 43: (192, 192), 44: (256, 256)}[algo]
                if m % bm or n % bn or (algo == 44 and ((m // 256) * (n // 256) <= 256 or (m // 256) * (n // 256) % 256 == 0)):
                    continue
                fns[algo] = (lambda al: (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=al)))(algo)
            fns[4] = (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=4))  # the picker's choice
            res = time_interleaved(fns)
            print(f"tiles {m}x{n}x{k}: " + "  ".join(f"algo {al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:6.0f} TF" for al, t in res.items()), flush=True)
    if "mxab" in args.which:  # interleaved A/B: per-tensor-scaled vs block-scaled GEMM (same persistent kernel, MX adds the scale path)
        for name, (M, N, K) in SHAPES_3B.items():
            for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                sa = torch.randint(120, 131, (k // 32, m), generator=g, device=dev, dtype=torch.uint8)
                sb = torch.randint(120, 131, (k // 32, n), generator=g, device=dev, dtype=torch.uint8)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                res = time_interleaved({"fp8": lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=4),
                                        "mx": lambda: ops.gemm_mxfp8(a, sa, b, sb, 0, 0, out=out, algo=4),
                                        "mx-noread": lambda: ops.gemm_mxfp8(a, sa, b, sb, 0, 0, out=out, algo=18),
                                        "mx-read-unused": lambda: ops.gemm_mxfp8(a, sa, b, sb, 0, 0, out=out, algo=19)})
                print(f"mxab {name:4s} {kind:5s} {m}x{n}x{k}: " + "  ".join(f"{al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:6.0f} TF" for al, t in res.items()), flush=True)
    if "mxgemm" in args.which:
        for name, (M, N, K) in SHAPES_3B.items():
            for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                sa = torch.randint(120, 131, (k // 32, m), generator=g, device=dev, dtype=torch.uint8)
                sb = torch.randint(120, 131, (k // 32, n), generator=g, device=dev, dtype=torch.uint8)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                t = time_fn(lambda: ops.gemm_mxfp8(a, sa, b, sb, 0, 0, out=out, algo=4), args.iters)
                tf = 2.0 * m * n * k / t / 1e12
                print(f"mxgemm {name:4s} {kind:5s} {m:6d}x{n:6d}x{k:6d}: {t*1e6:9.1f} us {tf:8.1f} TFLOP/s ({tf/5000*100:5.1f}% of 5 PF)", flush=True)
    if "ksweep" in args.which:
        # fixed 8192x8192 output (1024 tiles = 4 full rounds), K sweep: slope = mainloop cost per K-tile, intercept = per-tile overhead
        for algo in [int(x) for x in args.algos.split(",")]:
            for k in (256, 512, 1024, 2048, 4096, 8192):
                m = n = 8192
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                t = time_fn(lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=algo), args.iters)
                print(f"ksweep algo {algo} K={k:5d}: {t*1e6:9.1f} us  per-round {t*1e6/4:8.1f} us  {2.0*m*n*k/t/1e12:8.1f} TFLOP/s", flush=True)
    if "stagger" in args.which:  # start stagger of the persistent kernel (algo 16 + MI_GEMM_STAGGER units of 512 cycles per class)
        shapes = [(8192, 8192, 2048), (8192, 8192, 4096), (8192, 16384, 3072), (8192, 3072, 16384), (16384, 3072, 8192), (8192, 3072, 8192), (8192, 8192, 3072), (8192, 5120, 3072), (8192, 3072, 3072)]
        for (m, n, k) in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            def mk(units):
                def f():
                    os.environ["MI_GEMM_STAGGER"] = str(units)
                    ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=16)
                return f
            fns = {"block46": (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=46)),
                   "algo4": (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=4))}
            fns["half24"] = (lambda: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=24))
            res = time_interleaved(fns, rounds=8, inner=6)
            print(f"stagger {m}x{n}x{k}: " + "  ".join(f"{al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:6.0f} TF" for al, t in res.items()), flush=True)
    if "rotout" in args.which:  # do the half-line stores of the 192-column tiles pay extra when the output lines are cold?
        for (m, n, k) in ((8192, 3072, 3072), (8192, 3072, 8192), (8192, 5120, 3072)):
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            outs = [torch.empty((m, n), dtype=torch.bfloat16, device=dev) for _ in range(12)]  # 12 x 50-84 MB > the 256-MiB Infinity Cache
            cnt = [0]
            def mk(al, rot):
                def f():
                    cnt[0] += 1
                    ops.gemm_fp8(a, b, one, one, 0, 0, out=outs[cnt[0] % 12 if rot else 0], algo=al)
                return f
            fns = {f"{name}{'_rot' if rot else ''}": mk(al, rot) for name, al in (("256x256", 40), ("256x192", 41)) for rot in (False, True)
                   if not (al == 41 and n % 192)}
            res = time_interleaved(fns, rounds=8, inner=6)
            print(f"rotout {m}x{n}x{k}: " + "  ".join(f"{al}: {t*1e6:7.1f} us" for al, t in res.items()), flush=True)
    if "rotcast" in args.which:  # do the 64-byte row segments of the cast kernels pay for cold output lines, as the GEMM's half lines do?
        scale, amax = torch.ones(1, device=dev), torch.zeros(1, device=dev)
        for (r, c) in ((8192, 3072), (8192, 8192)):
            xs = [torch.randn(r, c, device=dev, dtype=torch.bfloat16) for _ in range(8)]
            ys = [torch.empty((r, c), dtype=torch.uint8, device=dev) for _ in range(16)]
            yts = [torch.empty((c, r), dtype=torch.uint8, device=dev) for _ in range(16)]
            cnt = [0]
            def mk(rot_in, rot_out):
                def f():
                    cnt[0] += 1
                    ops.cast_amax(xs[cnt[0] % 8 if rot_in else 0], scale, amax, 0, y=ys[cnt[0] % 16 if rot_out else 0], yT=yts[cnt[0] % 16 if rot_out else 0])
                return f
            fns = {"warm": mk(False, False), "rot_in": mk(True, False), "rot_out": mk(False, True), "rot_both": mk(True, True)}
            res = time_interleaved(fns, rounds=8, inner=8)
            nb = r * c * 4
            print(f"rotcast cast_amax {r}x{c}: " + "  ".join(f"{k_}: {t*1e6:6.1f} us {nb/t/1e12:5.2f} TB/s" for k_, t in res.items()), flush=True)
        for (r, c) in ((8192, 8192), (8192, 16384)):  # which copy pays: the row-major one or the transposed one?
            x0 = torch.randn(r, c, device=dev, dtype=torch.bfloat16)
            ys = [torch.empty((r, c), dtype=torch.uint8, device=dev) for _ in range(16)]
            yts = [torch.empty((c, r), dtype=torch.uint8, device=dev) for _ in range(16)]
            cnt = [0]
            def mk1(want_y, want_t, rot):
                def f():
                    cnt[0] += 1
                    i = cnt[0] % 16 if rot else 0
                    ops.cast_amax(x0, scale, amax, 0, want_y=want_y, want_t=want_t, y=ys[i] if want_y else None, yT=yts[i] if want_t else None)
                return f
            fns = {"y_warm": mk1(True, False, False), "y_rot": mk1(True, False, True), "yT_warm": mk1(False, True, False), "yT_rot": mk1(False, True, True)}
            res = time_interleaved(fns, rounds=8, inner=8)
            print(f"rotcast cast_amax {r}x{c} one copy: " + "  ".join(f"{k_}: {t*1e6:6.1f} us" for k_, t in res.items()), flush=True)
        h_in = [torch.randn(8192, 16384, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        cnt = [0]
        def sw(rot):
            def f():
                cnt[0] += 1
                ops.swiglu_cast(h_in[cnt[0] % 4 if rot else 0], scale, amax, 0)
            return f
        res = time_interleaved({"warm_in": sw(False), "rot_in": sw(True)}, rounds=8, inner=6)
        print("rotcast swiglu_cast 8192x8192 (outputs are fresh allocations either way): " + "  ".join(f"{k_}: {t*1e6:6.1f} us" for k_, t in res.items()), flush=True)
    if "mlpchunk" in args.which:  # does an M-chunked fc1 -> SwiGLU keep the bf16 fc1 output in the 256-MiB Infinity Cache?
        scale, amax = torch.ones(1, device=dev), torch.zeros(1, device=dev)
        M, F2, K = 8192, 16384, 3072
        a, b = rand_fp8((M, K), dev, g), rand_fp8((F2, K), dev, g)
        bias = torch.zeros(F2, device=dev, dtype=torch.bfloat16)
        hs = [torch.empty((M, F2), dtype=torch.bfloat16, device=dev) for _ in range(6)]  # rotate: 6 x 268 MB
        cnt = [0]
        def run(chunks):
            def f():
                cnt[0] += 1
                h = hs[cnt[0] % 6]
                mc = M // chunks
                for c in range(chunks):
                    ops.gemm_fp8(a[c * mc:(c + 1) * mc], b, one, one, 0, 0, bias=bias, out=h[c * mc:(c + 1) * mc])
                    ops.swiglu_cast(h[c * mc:(c + 1) * mc], scale, amax, 0)
            return f
        res = time_interleaved({"1 chunk": run(1), "2 chunks": run(2), "4 chunks": run(4)}, rounds=8, inner=4)
        print("mlpchunk fc1 fprop + swiglu_cast, 8192 x 16384 x 3072: " + "  ".join(f"{k_}: {t*1e6:7.1f} us" for k_, t in res.items()), flush=True)
    if "storepol" in args.which:  # epilogue store cache policy: sc1 (default) / plain / nt / sc1+nt / no stores, interleaved A/B
        shapes = [(8192, 8192, 2048), (8192, 16384, 3072), (8192, 3072, 16384), (16384, 3072, 8192), (8192, 3072, 8192), (8192, 8192, 3072), (8192, 5120, 3072), (8192, 3072, 3072),
                  (8192, 128256, 3072), (128256, 3072, 8192), (8192, 28672, 4096), (6144, 28672, 4096)]
        for (m, n, k) in shapes:
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            fns = {name: (lambda al=al: ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=al))
                   for name, al in (("sc1", 4), ("serp", 30), ("hybrid", 29), ("plain", 17), ("nt", 25), ("sc1nt", 26), ("nostore", 15), ("noepi", 28), ("block46", 46))}
            ref = ops.gemm_fp8(a, b, one, one, 0, 0, algo=4)
            hyb = ops.gemm_fp8(a, b, one, one, 0, 0, algo=29)
            assert torch.equal(ref, hyb), f"algo 29 differs from algo 4 on {m}x{n}x{k}: {(ref.float() - hyb.float()).abs().max().item()}"
            res = time_interleaved(fns, rounds=8, inner=6)
            print(f"storepol {m}x{n}x{k}: " + "  ".join(f"{al}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:6.0f} TF" for al, t in res.items()), flush=True)
    if "stamps" in args.which:  # per-phase timeline of workgroup 0 (algo 22), around the tile boundaries
        from llm_fp8_amd import _lib
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        for (m, n, k) in ((8192, 8192, 2048), (8192, 16384, 3072), (8192, 8192, 3072)):
            a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
            out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
            dbg = torch.zeros((2, 1024), dtype=torch.int64, device=dev)
            def run(algo, bias_ptr):
                rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), bias_ptr,
                                     m, n, k, k, k, n, 0, 0, 0, algo, st)
                assert rc == 0, lib.mi_last_error()
            for _ in range(300):
                run(4, None)
            run(22, dbg.data_ptr())
            torch.cuda.synchronize()
            d = dbg.cpu().numpy()
            nk = k // 128
            for grp in (0, 1):
                t = d[grp]
                t = t[t > 0]
                # stamps alternate: A (start of MFMA segment), B (end of phase); 8 per K-tile
                A, B = t[0::2], t[1::2]
                nph = min(len(A), len(B))
                load_seg = A[1:nph] - B[:nph - 1]      # end of phase p-1 -> MFMA start of phase p (fragment reads, issues, wait, barrier)
                mfma_seg = B[:nph] - A[:nph]
                per_kt = (B[4:nph:4] - B[0:nph - 4:4])
                print(f"stamps {m}x{n}x{k} group {grp}: {nph} phases; cycles per K-tile: " + " ".join(str(int(x)) for x in per_kt[:3 * nk]), flush=True)
                for ti in range(1, 3):
                    base = ti * nk * 4
                    if base + 12 > nph:
                        break
                    lo = base - 12
                    print(f"   boundary into tile {ti}, K-tiles nk-3 .. +2, per phase [load/wait seg | mfma seg]: "
                          + "  ".join((("|| " if (x - base) % 4 == 0 else "") + f"{int(load_seg[x - 1])}/{int(mfma_seg[x])}") for x in range(lo, base + 12)), flush=True)
    if "abbase" in args.which:  # same-process A/B of the current library against a saved build (default: the round-1 library)
        import ctypes
        from llm_fp8_amd import _lib
        cur = _lib.load()
        base_path = os.environ.get("MI_BASE_LIB")  # a saved build of an earlier tree (tools/bin/ is scratch: nothing is kept there)
        if not base_path or not os.path.exists(base_path):
            raise SystemExit("--which abbase: set MI_BASE_LIB to the libmi_fp8.so of the build to compare against")
        base = ctypes.CDLL(base_path)
        for lib_ in (base,):
            lib_.mi_gemm_fp8.argtypes = cur.mi_gemm_fp8.argtypes
            lib_.mi_gemm_fp8.restype = ctypes.c_int
        st = torch.cuda.current_stream().cuda_stream
        models = {"3b": (8192, {"qkv": (5120, 3072), "o": (3072, 3072), "fc1": (16384, 3072), "fc2": (3072, 8192)}),
                  "1b": (8192, {"qkv": (3072, 2048), "o": (2048, 2048), "fc1": (16384, 2048), "fc2": (2048, 8192)}),
                  "8b": (6144, {"qkv": (6144, 4096), "o": (4096, 4096), "fc1": (28672, 4096), "fc2": (4096, 14336)})}
        tot = {}
        for mname in os.environ.get("MODELS", "3b").split(","):
            M, sites = models[mname]
            tb = tc = 0.0
            for name, (N, K) in sites.items():
                for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
                    a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                    def mk(lib_):
                        def f():
                            rc = lib_.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), None,
                                                  m, n, k, k, k, n, 0, 0, 0, 4, st)
                            assert rc == 0
                        return f
                    res = time_interleaved({"base": mk(base), "cur": mk(cur)}, rounds=10, inner=5)
                    tb += res["base"]; tc += res["cur"]
                    fl = 2.0 * m * n * k
                    print(f"abbase {mname} {name:4s} {kind:5s} {m}x{n}x{k}: base {res['base']*1e6:7.1f} us {fl/res['base']/1e12:6.0f} TF   cur {res['cur']*1e6:7.1f} us "
                          f"{fl/res['cur']/1e12:6.0f} TF   {res['base']/res['cur']:.3f}x", flush=True)
            print(f"abbase {mname} decoder-layer GEMM time: base {tb*1e6:8.1f} us  cur {tc*1e6:8.1f} us  {tb/tc:.3f}x", flush=True)
    if "grouped" in args.which:  # a Linear's dgrad + wgrad: two launches vs mi_gemm_fp8_grouped, interleaved A/B
        models = {"3b": (8192, {"qkv": (5120, 3072), "o": (3072, 3072), "fc1": (16384, 3072), "fc2": (3072, 8192), "lm_head": (128256, 3072)}),
                  "1b": (8192, {"qkv": (3072, 2048), "o": (2048, 2048), "fc1": (16384, 2048), "fc2": (2048, 8192)}),
                  "8b": (6144, {"qkv": (6144, 4096), "o": (4096, 4096), "fc1": (28672, 4096), "fc2": (4096, 14336)})}
        for mname in os.environ.get("MODELS", "3b").split(","):
            M, sites = models[mname]
            tsep = tgrp = 0.0
            for name, (N, K) in sites.items():
                g8, w8t = rand_fp8((M, N), dev, g), rand_fp8((K, N), dev, g)      # dgrad: dX[M,K] = G8[M,N] . W8T[K,N]^T
                g8t, x8t = rand_fp8((N, M), dev, g), rand_fp8((K, M), dev, g)    # wgrad: dW[N,K] = G8T[N,M] . X8T[K,M]^T
                dx = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
                dw = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
                if not ops.grouped_gemm_ok([(M, K, N), (N, K, M)]):
                    print(f"grouped {mname} {name}: not eligible", flush=True)
                    continue
                def sep():
                    ops.gemm_fp8(g8, w8t, one, one, 0, 0, out=dx, algo=4)
                    ops.gemm_fp8(g8t, x8t, one, one, 0, 0, out=dw, algo=4)
                def grp():
                    ops.gemm_fp8_grouped([(g8, w8t, one, one, dx), (g8t, x8t, one, one, dw)], 0, 0)
                def grp4():
                    ops.gemm_fp8_grouped([(g8, w8t, one, one, dx), (g8t, x8t, one, one, dw)], 0, 0, tile_cfg=4)
                cands = {"separate": sep, "grouped": grp}
                if M % 256 == 0 and N % 256 == 0 and K % 256 == 0:
                    cands["grouped_w4"] = grp4
                res = time_interleaved(cands, rounds=10, inner=4)
                fl = 4.0 * M * N * K
                tsep += res["separate"]; tgrp += res["grouped"]
                print(f"grouped {mname} {name:7s} bwd M={M} N={N} K={K}: separate {res['separate']*1e6:7.1f} us {fl/res['separate']/1e12:6.0f} TF   "
                      f"grouped {res['grouped']*1e6:7.1f} us {fl/res['grouped']/1e12:6.0f} TF   {res['separate']/res['grouped']:.3f}x"
                      + (f"   grouped_w4 {res['grouped_w4']*1e6:7.1f} us {fl/res['grouped_w4']/1e12:6.0f} TF   {res['grouped']/res['grouped_w4']:.3f}x vs grouped" if "grouped_w4" in res else ""), flush=True)
            print(f"grouped {mname} total: separate {tsep*1e6:8.1f} us  grouped {tgrp*1e6:8.1f} us  {tsep/tgrp:.3f}x", flush=True)
    if "clock" in args.which:
        from llm_fp8_amd import _lib
        lib = _lib.load()
        m = n = 8192
        for fill in ("random", "zeros"):
            for k in (2048, 8192):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                if fill == "zeros":
                    a.zero_(); b.zero_()
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                dbg = torch.zeros((1024, 2), dtype=torch.int64, device=dev)
                st = torch.cuda.current_stream().cuda_stream
                def run(algo, bias_ptr):
                    rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), bias_ptr,
                                         m, n, k, k, k, n, 0, 0, 0, algo, st)
                    assert rc == 0, lib.mi_last_error()
                for _ in range(200):  # heat up
                    run(13, None)
                run(14, dbg.data_ptr())
                torch.cuda.synchronize()
                d = dbg.cpu().double()
                cyc, ticks = d[:, 0], d[:, 1]
                clk = (cyc / ticks * 100.0)  # MHz
                t = time_fn(lambda: run(13, None), args.iters)
                print(f"clock {fill:6s} K={k}: loop cycles/K-tile median {float((cyc / (k/128)).median()):8.1f} "
                      f"in-kernel clock median {float(clk.median()):7.1f} MHz (min {float(clk.min()):.0f} max {float(clk.max()):.0f}); "
                      f"no-store kernel {t*1e6:8.1f} us = {2.0*m*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
    if "pclock" in args.which:  # in-kernel clock of the PERSISTENT kernel (algo 21 = stamped build), per decoder GEMM site, random data
        from llm_fp8_amd import _lib
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        shapes = [(8192, 8192, 8192)] + [s for name, (M, N, K) in SHAPES_3B.items() for s in ((M, N, K), (M, K, N), (N, K, M))]
        for (m, n, k) in shapes:
            for fill in ("random", "zeros"):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                if fill == "zeros":
                    a.zero_(); b.zero_()
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                dbg = torch.zeros((256, 4), dtype=torch.int64, device=dev)
                def run(algo, bias_ptr):
                    rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), bias_ptr,
                                         m, n, k, k, k, n, 0, 0, 0, algo, st)
                    assert rc == 0, lib.mi_last_error()
                t0 = time.time()
                while time.time() - t0 < (2.0 if fill == "random" else 1.0):  # heat up: >= 2 s back to back
                    for _ in range(50):
                        run(4, None)
                    torch.cuda.synchronize()
                run(21, dbg.data_ptr())
                torch.cuda.synchronize()
                d = dbg.cpu().double()
                d = d[d[:, 1] > 0]
                clk = d[:, 0] / d[:, 1] * 100.0
                cyc_per_step = d[:, 0] / d[:, 2]
                t = time_fn(lambda: run(4, None), args.iters)
                tf = 2.0 * m * n * k / t / 1e12
                print(f"pclock {m}x{n}x{k} {fill:6s}: clock median {float(clk.median()):7.1f} MHz (min {float(clk.min()):.0f} max {float(clk.max()):.0f}) "
                      f"cycles/K-tile-step median {float(cyc_per_step.median()):7.1f} (ideal 2048)  {t*1e6:8.1f} us {tf:7.1f} TF "
                      f"= {tf/5000*100:4.1f}% of 5 PF = {tf/(5000*float(clk.median())/2400)*100:4.1f}% at clock", flush=True)
    if "gridsweep" in args.which:  # is the tile-boundary cost a per-CU or a chip-wide limit?  same per-CU work on fewer CUs (MI_GEMM_GRID, algo 21)
        from llm_fp8_amd import _lib
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        for grid in (256, 128, 64, 32):
            os.environ["MI_GEMM_GRID"] = str(grid)
            for (m, n, k) in ((8192 * grid // 256, 8192, 3072), (8192 * grid // 256, 8192, 8192)):
                a, b = rand_fp8((m, k), dev, g), rand_fp8((n, k), dev, g)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
                dbg = torch.zeros((256, 4), dtype=torch.int64, device=dev)
                def run(algo, bias_ptr):
                    rc = lib.mi_gemm_fp8(a.data_ptr(), b.data_ptr(), out.data_ptr(), one.data_ptr(), one.data_ptr(), bias_ptr,
                                         m, n, k, k, k, n, 0, 0, 0, algo, st)
                    assert rc == 0, lib.mi_last_error()
                for _ in range(100):
                    run(21, dbg.data_ptr())
                torch.cuda.synchronize()
                d = dbg.cpu().double()
                d = d[d[:, 1] > 0]
                clk = d[:, 0] / d[:, 1] * 100.0
                cyc = d[:, 0] / d[:, 2]
                print(f"gridsweep grid {grid:3d} {m}x{n}x{k}: {len(d)} workgroups, {int(d[0, 2])} K-steps each, cycles/K-tile-step median {float(cyc.median()):7.1f} "
                      f"(min {float(cyc.min()):.0f} max {float(cyc.max()):.0f}), clock {float(clk.median()):6.0f} MHz", flush=True)
        os.environ.pop("MI_GEMM_GRID", None)
    if "cast" in args.which:
        for (R, C) in ((8192, 3072), (8192, 16384), (16384, 3072), (8192, 8192)):
            x = torch.randn((R, C), device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
            amax = torch.zeros(1, device=dev)
            y = torch.empty((R, C), dtype=torch.uint8, device=dev)
            yT = torch.empty((C, R), dtype=torch.uint8, device=dev)
            t = time_fn(lambda: ops.cast_amax(x, one, amax, 0, y=y, yT=yT), args.iters)
            print(f"cast+T {R}x{C}: {t*1e6:8.1f} us {4.0*R*C/t/1e9:8.1f} GB/s (alg 4 B/elem)", flush=True)
            t = time_fn(lambda: ops.cast_amax(x, one, amax, 0, y=y, want_t=False), args.iters)
            print(f"cast   {R}x{C}: {t*1e6:8.1f} us {3.0*R*C/t/1e9:8.1f} GB/s (alg 3 B/elem)", flush=True)
            t = time_fn(lambda: ops.cast_amax(x, one, amax, 0, yT=yT, want_y=False), args.iters)
            print(f"castT  {R}x{C}: {t*1e6:8.1f} us {3.0*R*C/t/1e9:8.1f} GB/s (alg 3 B/elem, transposed copy only)", flush=True)
            x2 = torch.empty_like(x)
            t = time_fn(lambda: x2.copy_(x), args.iters)
            print(f"copy   {R}x{C}: {t*1e6:8.1f} us {4.0*R*C/t/1e9:8.1f} GB/s (torch bf16 copy, 4 B/elem)", flush=True)
    if "swiglu" in args.which:
        R, F = 8192, 8192
        h = torch.randn((R, 2 * F), device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
        d = torch.randn((R, F), device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
        amax = torch.zeros(1, device=dev)
        t = time_fn(lambda: ops.swiglu_cast(h, one, amax, 0), args.iters)
        print(f"swiglu  fwd {R}x{F}: {t*1e6:8.1f} us {6.0*R*F/t/1e9:8.1f} GB/s (6 B/out-elem)", flush=True)
        t = time_fn(lambda: ops.dswiglu_cast(h, d, one, amax, 0, want_colsum=True), args.iters)
        print(f"dswiglu bwd {R}x{F}: {t*1e6:8.1f} us {10.0*R*F/t/1e9:8.1f} GB/s (10 B/gate-elem)", flush=True)
    if "mx" in args.which:
        for (R, C) in ((8192, 3072), (8192, 16384)):
            x = torch.randn((R, C), device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16)
            t = time_fn(lambda: ops.mxfp8_quantize(x), args.iters)
            print(f"mxquant {R}x{C}: {t*1e6:8.1f} us {(4.0+2/32)*R*C/t/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
