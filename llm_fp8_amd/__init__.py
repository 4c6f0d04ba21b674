"""llm_fp8_amd -- MI355X-native FP8 Linear hot path (drop-in for the transformer_engine surface
that xuanvinh1997/llm-fp8's te_llama*.py wrappers consume).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"

from . import common  # noqa: F401
from . import pytorch  # noqa: F401  (so `import llm_fp8_amd as te; te.pytorch.Linear` works like transformer_engine)
