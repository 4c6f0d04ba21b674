import os, sys, torch
sys.path.insert(0, os.getcwd())
from llm_fp8_amd.pytorch import ops
from tools.bench_kernels import time_fn
dev = torch.device("cuda:0")
one = torch.ones(1, device=dev); amax = torch.zeros(1, device=dev)
for C in (1536, 3072, 6144, 12288, 24576):
    R = 8192
    xs = [torch.randn((R, C), device=dev, dtype=torch.bfloat16) for _ in range(4)]
    i = [0]
    def cast():
        i[0] += 1; ops.cast_amax(xs[i[0] % 4], one, amax, 0)
    def mx():
        i[0] += 1; ops.mxfp8_quantize(xs[i[0] % 4])
    def copy():
        i[0] += 1; xs[(i[0] + 1) % 4].copy_(xs[i[0] % 4])
    tc, tm, tp = time_fn(cast, 30), time_fn(mx, 30), time_fn(copy, 30)
    mb = R * C * 4 / 1e6
    print(f"size sweep 8192x{C}: {mb:7.1f} MB (4 B/elem)   cast+T {tc*1e6:7.1f} us {mb/tc/1e6:5.2f} TB/s   mxquant {tm*1e6:7.1f} us {mb*1.016/tm/1e6:5.2f} TB/s   torch bf16 copy {tp*1e6:7.1f} us {mb/tp/1e6:5.2f} TB/s", flush=True)
# the MX SwiGLU pair at the 3B step's shape (gate space 8192 x 8192): forward reads h (2 x 2 B) and writes both orientations
# (2 x 1 B + scales); backward reads h and d(act) (3 x 2 B) and writes d(h) in both orientations (2 x 2 x 1 B + scales)
R, F = 8192, 8192
hs = [torch.randn((R, 2 * F), device=dev, dtype=torch.bfloat16) for _ in range(2)]
ds = [torch.randn((R, F), device=dev, dtype=torch.bfloat16) for _ in range(2)]
k = [0]
def mx_fwd():
    k[0] += 1; ops.mxfp8_swiglu_quantize(hs[k[0] % 2])
def mx_bwd():
    k[0] += 1; ops.mxfp8_dswiglu_quantize(hs[k[0] % 2], ds[k[0] % 2], want_colsum=True)
def ds_bwd():
    k[0] += 1; ops.dswiglu_cast(hs[k[0] % 2], ds[k[0] % 2], one, amax, 0, want_colsum=True)
for name, fn, bpe in (("mx swiglu fwd", mx_fwd, 6.0 + 2 / 32), ("mx dswiglu bwd", mx_bwd, 10.0 + 4 / 32), ("delayed dswiglu bwd", ds_bwd, 10.0)):
    t = time_fn(fn, 30)
    print(f"swiglu pair {name:20s} {R}x{F}: {t*1e6:7.1f} us {bpe*R*F/t/1e12:5.2f} TB/s ({bpe:.2f} B/gate-elem)", flush=True)
