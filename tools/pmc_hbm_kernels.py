"""Run the HBM-bound cast / quantise kernels of the Linear path a few times at the Llama-3.2-3B bench shapes, for
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs).  Inputs are re-randomised per repetition into DIFFERENT
buffers (8 rotating sets, > the 256-MiB Infinity Cache in total for the large shapes) so that a launch does not find its
input resident from the previous one."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ROT = 8
scale = torch.ones(1, device=dev)
amax = torch.zeros(1, device=dev)
M, H, F = 8192, 3072, 8192
xs = [torch.randn(M, H, device=dev, dtype=torch.bfloat16) for _ in range(ROT)]        # cast_amax input (activations / gradients)
hs = [torch.randn(M, 2 * F, device=dev, dtype=torch.bfloat16) for _ in range(ROT)]    # fc1 output (gate | up)
ds = [torch.randn(M, F, device=dev, dtype=torch.bfloat16) for _ in range(ROT)]        # d(act)
torch.cuda.synchronize()
for r in range(reps):
    ops.cast_amax(xs[r % ROT], scale, amax, 0)                       # mi::cast_amax_kernel  y + yT
for r in range(reps):
    ops.swiglu_cast(hs[r % ROT], scale, amax, 0)                     # mi::swiglu_cast_kernel (forward)
for r in range(reps):
    ops.dswiglu_cast(hs[r % ROT], ds[r % ROT], scale, amax, 1)       # mi::swiglu_cast_kernel (backward form)
for r in range(reps):
    ops.mxfp8_quantize(xs[r % ROT], 0, True, True)                   # mi::mxfp8_quant_kernel  row + column copies
torch.cuda.synchronize()
print("done")
