"""Generic torch.nn leaves on the MI355X kernels: nn.Linear, nn.Embedding, nn.LayerNorm.

Used for the modules of the supported models that are plain torch.nn classes in the reference
(`lm_head`, `embed_tokens` in modeling_llama.py:353,381,436; BERT/CLIP/GPT-2 LayerNorms and dense layers).
CPU tensors, unsupported dtypes/shapes and exotic options take the original torch forward."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


def _gpu(t: torch.Tensor) -> bool:
    return t.is_cuda or ops.backend_is_emulated()


# ---- fallback accounting (VERDICT r2 "no silent fallbacks") ---------------------------------------------------------
# Every replacement class keeps the reference's own forward for what it does not implement (odd shapes, fp32 modules,
# cached decode, exotic options).  For CPU tensors that is the contract (BASELINE config 1 runs the reference unchanged);
# for a GPU tensor it means ATen / vendor kernels are doing work this package claims -- so each such call is counted, by
# class and reason.  `transformers_amd.fallback_calls()` reads the counters; bench.py prints their sum as
# "fallback_calls" and the GPU model tests assert it is zero for the BASELINE configurations.
_FALLBACKS: dict = {}


_WARNED = set()


def note_fallback(mod, x, reason: str = "unsupported") -> None:
    """Record that `mod` is about to serve the GPU tensor `x` through the reference's forward."""
    if x is not None and torch.is_tensor(x) and _gpu(x):
        key = f"{type(mod).__name__}:{reason}"
        _FALLBACKS[key] = _FALLBACKS.get(key, 0) + 1
        if reason == "kv_cache" and key not in _WARNED and torch.is_grad_enabled():
            # a training forward that left `use_cache` at the config's default (True): the reference builds a DynamicCache and
            # hands it to every layer (modeling_llama.py:383-384); the fused layer does not feed a cache under autograd.  The
            # reference's Trainer switches it off itself (trainer.py:616-617); a hand-written loop has to.
            _WARNED.add(key)
            import warnings

            warnings.warn(f"transformers_amd: {type(mod).__name__} received a KV cache under autograd and runs the reference "
                          "module around the kernels (slower than the fused layer): pass use_cache=False (or set "
                          "model.config.use_cache = False) in training forwards", stacklevel=3)


def fallback_calls(reset: bool = False) -> dict:
    """{"Class:reason": calls} of GPU tensors served by a reference forward since the last reset."""
    out = dict(_FALLBACKS)
    if reset:
        _FALLBACKS.clear()
    return out


class TamdLinear(nn.Linear):
    def forward(self, x):
        w = self.weight
        if (_gpu(x) and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and x.numel() > 0
                and ops.gemm_supported(x.numel() // x.shape[-1], w.shape[0], w.shape[1], x.dtype)):
            return ops.linear(x, w, self.bias)
        note_fallback(self, x)
        return super().forward(x)


class TamdEmbedding(nn.Embedding):
    def forward(self, ids):
        w = self.weight
        if (_gpu(w) and w.dtype in (torch.bfloat16, torch.float16, torch.float32) and self.max_norm is None
                and not self.sparse and not self.scale_grad_by_freq and w.shape[1] % 8 == 0
                and ids.dtype in (torch.int64, torch.int32)):
            return ops.embedding(ids, w, self.padding_idx)
        note_fallback(self, w)
        return super().forward(ids)


class TamdLayerNorm(nn.LayerNorm):
    def forward(self, x):
        if (_gpu(x) and self.elementwise_affine and len(self.normalized_shape) == 1
                and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.shape[-1] % 8 == 0
                and x.shape[-1] <= 8192 and self.weight.dtype == x.dtype):
            return ops.layernorm(x, self.weight, self.bias, self.eps)
        note_fallback(self, x)
        return super().forward(x)


REPLACEMENTS = {nn.Linear: TamdLinear, nn.Embedding: TamdEmbedding, nn.LayerNorm: TamdLayerNorm}


def _capture_hook_is_idle(fn) -> bool:
    """The reference's output recorders (utils/output_capturing.py:100-119) stay installed on attention modules after
    the first call that asked for hidden states / attentions, and do nothing unless the active collector wants their
    key.  Such a hook must not disable the fused layer for every later call."""
    if getattr(fn, "__name__", "") != "output_capturing_hook" or "output_capturing" not in getattr(fn, "__module__", ""):
        return False
    try:
        from transformers.utils.output_capturing import _active_collector

        wanted = _active_collector.get()
        key = dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in fn.__closure__))).get("key")
    except Exception:  # reference internals moved: be conservative
        return False
    return wanted is None or key not in wanted.keys()


def _has_hooks(*mods: nn.Module) -> bool:
    for m in mods:
        if m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return True
        if any(not _capture_hook_is_idle(fn) for fn in m._forward_hooks.values()):
            return True
    return False


def _hook_moves_weights(hook, device) -> bool:
    """Does bypassing this accelerate forward wrapper change what the child computes on?  `dispatch_model` attaches an
    `AlignDevicesHook` to EVERY submodule of a multi-device `device_map`, also to layers that sit wholly on one GPU: those only
    move inputs to `execution_device` (the layer's own wrapper has done that already) and are safe to bypass.  Unsafe: a hook
    that offloads (weights on `meta` / CPU until the wrapper runs), one that executes elsewhere, and anything unknown."""
    inner = getattr(hook, "hooks", None)  # SequentialHook
    if inner is not None:
        return any(_hook_moves_weights(h, device) for h in inner)
    if not hasattr(hook, "offload") or hook.offload:
        return True
    ed = getattr(hook, "execution_device", None)
    if ed is None:
        return False
    try:
        ed = torch.device("cuda", ed) if isinstance(ed, int) else torch.device(ed)
    except (RuntimeError, TypeError):
        return True
    if ed.type == "cuda" and ed.index is None and device.type == "cuda":
        return False
    return ed != device


def _placement_ok(layer: nn.Module, device, *probe: torch.Tensor) -> bool:
    """A fused layer path reads its children's weights directly instead of calling the children, so it bypasses the forward
    wrappers accelerate installs for `device_map` / CPU / disk offload (`_hf_hook`: accelerate replaces `forward`, it does not
    register a hook, so `_has_hooks` cannot see it) -- with offload a leaf's weight sits on `meta` until its wrapper runs.
    True when no CHILD of `layer` carries a wrapper that offloads or executes elsewhere (`_hook_moves_weights`: the plain
    execution-device hooks of a multi-GPU `device_map="auto"` do not count) and every parameter is on `device` (a wrapper on
    `layer` itself has already run by the time its forward is entered).  Cached on the identity of the `probe` weights: materialising or
    moving a weight replaces the parameter object, which invalidates the entry."""
    key = (device,) + tuple(id(t) for t in probe)
    cached = layer.__dict__.get("_tamd_placement")
    if cached is not None and cached[0] == key:
        return cached[1]
    ok = True
    for m in layer.modules():
        if m is not layer and "_hf_hook" in m.__dict__ and _hook_moves_weights(m.__dict__["_hf_hook"], device):
            ok = False
            break
    if ok:
        ok = all(p.device == device for p in layer.parameters())
    layer.__dict__["_tamd_placement"] = (key, ok)
    return ok

