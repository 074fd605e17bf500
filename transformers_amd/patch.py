"""In-place module replacement (SURVEY.md §8b "B2").

The reference's registry route (`register_patch_mapping` + `apply_patches`,
src/transformers/monkey_patching.py:85-155, 233-297) crashes in environments without torchvision
(`dir()` on lazy alias modules, monkey_patching.py:275), so the swap is done post-construction:
`module.__class__ = Replacement` for every module whose class is in the replacement tables.  The
replacement classes subclass the originals and add no parameters, so this is exact and reversible.
"""
from __future__ import annotations

from typing import Dict, Type

import torch
from torch import nn

from . import attention


def _tables() -> Dict[Type[nn.Module], Type[nn.Module]]:
    from .models import bert, clip, common, gpt2, llama

    table: Dict[Type[nn.Module], Type[nn.Module]] = {}
    for mod in (common, llama, bert, clip, gpt2):
        table.update(mod.REPLACEMENTS)
    return table


def accelerate(model: nn.Module, attn_implementation: bool = True, fuse_loss: bool = True,
               fused_lm_head_loss: bool = False) -> nn.Module:
    """Switch `model` (any PreTrainedModel containing Llama / BERT / CLIP / GPT-2 blocks) to the MI355X path.

    * registers and selects `attn_implementation="tamd"` (reference API: `set_attn_implementation`,
      src/transformers/modeling_utils.py:2041-2139);
    * swaps norm / MLP / attention / layer modules for their tamd subclasses in place;
    * installs the fused cross-entropy as `model.loss_function` for causal-LM heads
      (reference hook: modeling_utils.py:4652-4669);
    * `fused_lm_head_loss=True` (opt-in, Llama): training forwards with labels run the lm_head GEMM and the loss chunk by
      chunk and return `logits=None` -- no [tokens, vocab] tensor is allocated (ops.fused_linear_cross_entropy).
    Returns the same object.
    """
    attention.register()
    table = _tables()
    n = 0
    for m in model.modules():
        repl = table.get(type(m))
        if repl is not None:
            m.__class__ = repl
            n += 1
    model._tamd_swapped = n
    if n > 0:  # the unchanged Trainer builds DDP without bucket views and registers no hook: ddp.install_trainer_dropin
        from . import ddp

        ddp.install_trainer_dropin()
    if fused_lm_head_loss:
        import types

        from transformers.models.llama.modeling_llama import LlamaForCausalLM

        from .models.llama import fused_causal_lm_forward

        if not isinstance(model, LlamaForCausalLM):
            raise TypeError("fused_lm_head_loss=True is implemented for LlamaForCausalLM")
        model.forward = types.MethodType(fused_causal_lm_forward, model)  # instance attribute: the class is untouched
    from transformers.models.bert.modeling_bert import BertForMaskedLM

    if isinstance(model, BertForMaskedLM) and fuse_loss:
        import types

        from .models.bert import masked_lm_forward

        model.forward = types.MethodType(masked_lm_forward, model)  # instance attribute, like the fused Llama forward
    if attn_implementation and hasattr(model, "set_attn_implementation"):
        dtypes = {p.dtype for p in model.parameters() if p.is_floating_point()}
        if dtypes and not dtypes & {torch.bfloat16, torch.float16}:
            # the kernels are bf16/fp16: an fp32 model keeps the attention backend it has (under autocast the
            # registered function still accepts the fp32 q/k/v autocast leaves behind, attention.py)
            import warnings

            warnings.warn("transformers_amd.accelerate: fp32 model -- attn_implementation is left unchanged "
                          "(the MI355X kernels run bf16/fp16; load with dtype=torch.bfloat16 to use them)")
        else:
            model.set_attn_implementation(attention.ATTN_KEY)
    # eager weight fusion (before DDP wraps the model)
    for m in model.modules():
        fuse = getattr(m, "_fused", None)
        if callable(fuse) and all(p.is_cuda for p in m.parameters(recurse=True)):
            fuse().weight()
        pad = getattr(m, "_padded", None)
        if callable(pad) and all(p.is_cuda for p in m.parameters(recurse=True)):
            pad().buffers()
    from . import graph_stack

    graph_stack.attach(model)  # forward-only calls of a whole decoder stack replay one HIP graph (graph_stack.py)
    # `generate` with a pre-allocated cache wraps the forward in torch.compile on its own (generation/utils.py:2119-2160,
    # 2862-2863).  Our layers are opaque custom ops around host-side dispatch: the tracer gains nothing and breaks the graph in
    # every layer -- 30.8 ms per token against 5.5 without it at Llama-3-8B dimensions (profiles/r04t_decode_bench_32.jsonl).
    # The reference's own switch turns it off; a caller's explicit GenerationConfig still wins.
    gc = getattr(model, "generation_config", None)
    if gc is not None and n > 0 and not getattr(gc, "disable_compile", False):
        model.__dict__["_tamd_disable_compile_was"] = getattr(gc, "disable_compile", None)
        gc.disable_compile = True
    if fuse_loss and getattr(model, "loss_type", None) == "ForCausalLM":
        from .ops import causal_lm_loss

        model.loss_function = _LossDispatch(causal_lm_loss, model.loss_function)
    return model


class _LossDispatch:
    """GPU logits -> fused cross-entropy kernel; anything else -> the reference loss it replaced."""

    def __init__(self, fast, reference):
        self.fast, self.reference = fast, reference

    def __call__(self, logits, labels, vocab_size, **kwargs):
        from . import ops

        if logits.is_cuda or ops.backend_is_emulated():
            return self.fast(logits=logits, labels=labels, vocab_size=vocab_size, **kwargs)
        return self.reference(logits=logits, labels=labels, vocab_size=vocab_size, **kwargs)


def revert(model: nn.Module) -> nn.Module:
    """Undo `accelerate` (class swaps only; fused weight storage stays valid for the reference modules)."""
    inv = {v: k for k, v in _tables().items()}
    for m in model.modules():
        orig = inv.get(type(m))
        if orig is not None:
            m.__class__ = orig
    if isinstance(getattr(model, "loss_function", None), _LossDispatch):
        model.loss_function = model.loss_function.reference
    model.__dict__.pop("forward", None)  # the instance-level fused forward, if installed
    if "_tamd_disable_compile_was" in model.__dict__ and getattr(model, "generation_config", None) is not None:
        model.generation_config.disable_compile = model.__dict__.pop("_tamd_disable_compile_was")
    from . import graph_stack

    graph_stack.detach(model)
    return model
