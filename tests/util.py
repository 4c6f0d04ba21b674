import numpy as np
import torch

from oracle import fp8_oracle as O


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.bfloat16).cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits_to_bf16(bits: np.ndarray, device=None) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)
    return t.to(device) if device is not None else t


def u8(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


def dequant_table(fmt: int, device) -> torch.Tensor:
    return torch.from_numpy(O.fp8_decode_table(fmt)).to(device)


def assert_gemm_close(got_f32: np.ndarray, ref_f32: np.ndarray, what=""):
    tol = O.gemm_tolerance(ref_f32)
    diff = np.abs(got_f32.astype(np.float64) - ref_f32.astype(np.float64))
    bad = diff > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} outside |d| <= 2^-7|ref| + 1e-3 rms; "
                           f"max diff {diff.max():.4g} at {np.unravel_index(diff.argmax(), diff.shape)}")
