"""transformers_amd -- MI355X-native transformer-block execution path behind huggingface/transformers.

    import transformers_amd                      # registers attn_implementation="tamd"
    model = AutoModelForCausalLM.from_pretrained(..., attn_implementation="tamd", dtype=torch.bfloat16)
    transformers_amd.accelerate(model)           # swaps norm / MLP / attention / layer modules in place

See DESIGN.md for the kernel inventory and INTEGRATION.md for the boundary.
"""
__version__ = "0.1.0"

from . import attention as _attention
from . import layer_ops as _layer_ops  # noqa: F401  (registers torch.ops.tamd.llama_layer)
from .models.common import fallback_calls  # noqa: F401
from .optim import TamdAdamW  # noqa: F401
from .patch import accelerate, revert  # noqa: F401

try:  # register attn_implementation="tamd" with the reference on import
    _attention.register()
except Exception:  # transformers missing/broken: kernels stay usable through transformers_amd.ops
    pass
