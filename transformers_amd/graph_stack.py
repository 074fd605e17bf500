"""A whole decoder-layer stack as ONE replayed HIP graph (forward only).

Why: a prompt-sized forward (LLaVA-1.5: 1088 positions through 32 Llama layers) is ~300 kernel launches of 20-80 us; the
launch path of each -- dispatcher op, Python, ctypes -- costs as much host time as the kernel takes on the GPU, so the
stack is host-bound (profiles/r03a_llava_graph_ab.jsonl: 28.3 ms eager).  HIP graphs are the MI355X answer to launch-bound
inner loops.  The reference's own model code cannot be captured as a whole (LLaVA's `get_placeholder_mask` formats a
device scalar into an error message, models/llava/modeling_llava.py:205-213: a host sync), so the graph is taken where
this package owns the code: the decoder layers.  `LlamaModel.forward` (models/llama/modeling_llama.py:395-404) calls them
in a plain loop, `hidden_states = decoder_layer(hidden_states, ...)`; layer 0 runs the whole stack as a graph and hands
back the final hidden state, the other layers recognise that tensor (by identity) and pass it through.

Used when nothing is being differentiated, no KV cache is in play and no hook wants per-layer outputs; a shape is captured
the second time it is seen (a one-off forward stays eager).  `TAMD_HIP_GRAPH=0` switches it off.
"""
from __future__ import annotations

import os
import threading
from typing import Optional

import torch

from . import ops

_ENABLED = os.environ.get("TAMD_HIP_GRAPH", "1") != "0"
_MAX_GRAPHS = 4  # captured shapes kept per stack (each holds its own activation pool)


def enabled() -> bool:
    return _ENABLED


def set_enabled(on: bool) -> bool:
    """Switch graph replay of decoder stacks on / off (process-wide); returns the previous setting."""
    global _ENABLED
    old, _ENABLED = _ENABLED, bool(on)
    return old


class LlamaStackGraph:
    """The layers of one `LlamaModel` and the graphs captured for them (one per input signature)."""

    def __init__(self, layers):
        self.layers = list(layers)
        self._graphs = {}   # key -> (graph, static inputs, static output)
        self._seen = {}     # key -> times seen before capture
        self._pass = threading.local()  # the output handed out by layer 0, for the other layers to recognise
        self.replays = 0

    # ---- per-call protocol (called from TamdLlamaDecoderLayer.forward)
    def passthrough(self, hidden_states) -> bool:
        return getattr(self._pass, "out", None) is hidden_states

    def run(self, hidden_states, cos, sin, key_valid, q_start) -> Optional[torch.Tensor]:
        """Layer 0: the final hidden state of the stack from a graph replay, or None (run eagerly this time)."""
        self._pass.out = None
        if not (_ENABLED and hidden_states.is_cuda and not torch.is_grad_enabled()
                and not torch.cuda.is_current_stream_capturing()):
            return None
        from .models.common import _has_hooks

        # every layer must be on its fused path, and nobody may be listening to per-layer outputs (the reference's
        # output recorders hook the decoder layers for `output_hidden_states`, utils/output_capturing.py)
        if not all(layer._fused_ok(hidden_states, None) and not _has_hooks(layer) for layer in self.layers):
            return None
        key = self._key(hidden_states, cos, sin, key_valid, q_start)
        entry = self._graphs.get(key)
        if entry is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n == 0 or len(self._graphs) >= _MAX_GRAPHS:
                return None  # first sighting of this signature (or the cache is full): eager
            entry = self._capture(key, hidden_states, cos, sin, key_valid, q_start)
            if entry is None:
                return None
        graph, static, out = entry
        static["h"].copy_(hidden_states)
        static["cos"].copy_(cos)
        static["sin"].copy_(sin)
        if key_valid is not None:
            static["key_valid"].copy_(key_valid)
        if q_start is not None:
            static["q_start"].copy_(q_start)
        graph.replay()
        self.replays += 1
        self._pass.out = out
        return out

    # ---- internals
    def _weights_signature(self):
        sig = []
        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            sig.append((attn._fused().weight().data_ptr(), attn.o_proj.weight.data_ptr(),
                        mlp._fused().weight().data_ptr(), mlp.down_proj.weight.data_ptr(),
                        layer.input_layernorm.weight.data_ptr(), layer.post_attention_layernorm.weight.data_ptr()))
        return hash(tuple(sig))

    def _key(self, h, cos, sin, key_valid, q_start):
        return (tuple(h.shape), h.dtype, h.device.index, tuple(cos.shape), cos.dtype,
                None if key_valid is None else (tuple(key_valid.shape), key_valid.dtype),
                None if q_start is None else tuple(q_start.shape), self._weights_signature())

    def _forward_all(self, h, cos, sin, key_valid, q_start):
        from . import layer_ops

        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            qkv, gu = attn._fused(), mlp._fused()
            h = layer_ops.llama_layer(
                h, cos, sin, key_valid, q_start, layer.input_layernorm.weight, qkv.weight(), attn.q_proj.weight,
                attn.k_proj.weight, attn.v_proj.weight, attn.o_proj.weight, layer.post_attention_layernorm.weight,
                gu.weight(), mlp.gate_proj.weight, mlp.up_proj.weight, mlp.down_proj.weight,
                eps=layer.input_layernorm.variance_epsilon, hq=attn.config.num_attention_heads,
                hkv=attn.config.num_key_value_heads, d=attn.head_dim, scale=attn.scaling,
                causal=bool(attn.is_causal) and h.shape[1] > 1)
        return h

    def _capture(self, key, h, cos, sin, key_valid, q_start):
        static = {"h": h.clone(), "cos": cos.clone(), "sin": sin.clone(),
                  "key_valid": None if key_valid is None else key_valid.clone(),
                  "q_start": None if q_start is None else q_start.clone()}
        args = (static["h"], static["cos"], static["sin"], static["key_valid"], static["q_start"])
        try:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side), torch.no_grad():  # (the allocator wants a warm-up on a side stream)
                self._forward_all(*args)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph), torch.no_grad():
                out = self._forward_all(*args)
        except Exception:  # a capture that fails must leave the eager path intact: never try this signature again
            self._seen[key] = -(1 << 30)
            torch.cuda.synchronize()
            return None
        entry = (graph, static, out)
        self._graphs[key] = entry
        return entry


def attach(model) -> int:
    """Find the decoder stacks of `model` (a ModuleList made only of TamdLlamaDecoderLayer) and give their layers a shared
    LlamaStackGraph.  Called by `accelerate`; returns the number of stacks."""
    from torch import nn

    from .models.llama import TamdLlamaDecoderLayer

    n = 0
    for mod in model.modules():
        layers = getattr(mod, "layers", None)
        if isinstance(layers, nn.ModuleList) and len(layers) > 1 and all(type(l) is TamdLlamaDecoderLayer for l in layers):
            stack = LlamaStackGraph(layers)
            for i, layer in enumerate(layers):
                layer.__dict__["_tamd_stack"] = (stack, i)
            n += 1
    return n


def detach(model) -> None:
    for mod in model.modules():
        mod.__dict__.pop("_tamd_stack", None)
