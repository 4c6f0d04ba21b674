"""`llm_fp8_amd.pytorch` -- the subset of `transformer_engine.pytorch` the reference consumes
(SURVEY.md 8b): Linear, LayerNormLinear, LayerNormMLP, MultiheadAttention, LayerNorm, RMSNorm,
fp8_autocast, attention.RotaryPositionEmbedding, fp8.check_mxfp8_support."""
from . import attention, fp8, ops  # noqa: F401
from .attention import DotProductAttention, MultiheadAttention  # noqa: F401
from .fp8 import fp8_autocast  # noqa: F401
from .module import LayerNorm, LayerNormLinear, LayerNormMLP, Linear, RMSNorm  # noqa: F401


class TransformerLayer:  # accelerate only uses it for `isinstance` (utils/transformer_engine.py:109)
    def __init__(self, *a, **k):
        raise NotImplementedError("TransformerLayer is not on the reference's hot path; use MultiheadAttention + LayerNormMLP")
