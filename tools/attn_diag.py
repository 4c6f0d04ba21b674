"""Where does a wave of attn_fwd_kernel spend its cycles?  (diagnostic stamp build, shares only)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd import _lib
lib = _lib.use_lab_library()  # mi_attn_fwd_diag lives in the lab build (make -C llm_fp8_amd/csrc lab)
P, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.mi_attn_fwd_diag.argtypes = [P, P, P, P, P, P, I, I, I, I, I, I64, I64, I64, I64, ctypes.c_float, P]
dev = torch.device("cuda:0")
B, S, H, G, D = 16, 512, 24, 8, 128
q, k, v = (torch.randn(B, S, n, D, device=dev, dtype=torch.bfloat16) for n in (H, G, G))
o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
dbg = torch.zeros(B, H, S // 128, 4, 8, dtype=torch.int64, device=dev)
for _ in range(2):
    rc = lib.mi_attn_fwd_diag(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), dbg.data_ptr(), B, S, H, G, D,
                              q.stride(1), k.stride(1), v.stride(1), o.stride(1), D ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
d = dbg.cpu().double()
names = ["S^T MFMAs (+K frag reads)", "softmax", "P.V MFMAs (+V tr reads)", "stage store (+vmcnt)", "barrier"]
for qb in range(S // 128):
    for w in range(4):
        x = d[:, :, qb, w, :5].mean((0, 1)); nt = d[0, 0, qb, w, 5].item()
        print(f"qb {qb} wave {w} tiles {int(nt)}: " + "  ".join(f"{n.split()[0]} {x[i].item()/nt:7.0f}" for i, n in enumerate(names)) + f"   total/tile {x.sum().item()/nt:7.0f}")
x = d[..., :5].sum((0, 1, 2, 3)); print("shares:", {n: round((x[i] / x.sum()).item(), 3) for i, n in enumerate(names)})
