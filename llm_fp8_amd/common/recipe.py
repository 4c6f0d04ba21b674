"""FP8 recipes -- same names, fields and defaults as `transformer_engine.common.recipe`, which the
reference imports at te_llama.py:25, te_llama_hybrid.py:25, te_llama_mxfp8.py:25 and accelerate
builds at utils/transformer_engine.py:156-177."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Callable, NamedTuple, Optional, Tuple, Union


class _FormatHelper(NamedTuple):
    max_fwd: float
    max_bwd: float


class Format(Enum):
    """E4M3: all tensors E4M3.  E5M2: all E5M2.  HYBRID: forward tensors E4M3, gradients E5M2."""
    E4M3 = _FormatHelper(max_fwd=448.0, max_bwd=448.0)
    E5M2 = _FormatHelper(max_fwd=57344.0, max_bwd=57344.0)
    HYBRID = _FormatHelper(max_fwd=448.0, max_bwd=57344.0)


class Recipe:
    def mxfp8(self) -> bool:
        return isinstance(self, MXFP8BlockScaling)

    def delayed(self) -> bool:
        return isinstance(self, DelayedScaling)


@dataclass(frozen=True)
class DelayedScaling(Recipe):
    """Per-tensor delayed scaling (te_llama.py:39-40: history 16, algo "max"; accelerate default:
    HYBRID, history 1024, "most_recent")."""
    margin: int = 0
    fp8_format: Format = Format.HYBRID
    amax_history_len: int = 1024
    amax_compute_algo: Union[str, Callable] = "max"
    scaling_factor_compute_algo: Optional[Callable] = None
    reduce_amax: bool = True
    fp8_dpa: bool = False
    fp8_mha: bool = False
    interval: int = 1  # deprecated in TE, accepted for accelerate's TERecipeKwargs
    override_linear_precision: Tuple[bool, bool, bool] = (False, False, False)

    def __post_init__(self):
        assert self.fp8_format in (Format.E4M3, Format.E5M2, Format.HYBRID)
        if self.amax_compute_algo not in ("max", "most_recent"):
            raise ValueError("amax_compute_algo must be 'max' or 'most_recent' (callables are not supported)")
        if self.scaling_factor_compute_algo is not None:
            raise ValueError("custom scaling_factor_compute_algo is not supported")
        if self.amax_history_len < 1 or self.amax_history_len > 4096:
            raise ValueError("amax_history_len must be in [1, 4096]")
        if self.override_linear_precision != (False, False, False):
            raise ValueError("override_linear_precision is not supported")


@dataclass(frozen=True)
class MXFP8BlockScaling(Recipe):
    """OCP MX block scaling: one E8M0 scale per 32 elements along the contraction axis, no history
    (te_llama_mxfp8.py:28-29)."""
    margin: int = 0
    fp8_format: Format = Format.E4M3
    fp8_dpa: bool = False
    fp8_mha: bool = False

    def __post_init__(self):
        assert self.fp8_format in (Format.E4M3, Format.E5M2, Format.HYBRID)
        if self.margin != 0:
            raise ValueError("MXFP8BlockScaling margin must be 0")


def fmt_codes(fmt: Format) -> Tuple[int, int]:
    """(forward code, backward code) for the C ABI: 0 = E4M3, 1 = E5M2."""
    fwd = 1 if fmt is Format.E5M2 else 0
    bwd = 0 if fmt is Format.E4M3 else 1
    return fwd, bwd
