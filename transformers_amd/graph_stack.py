"""A whole decoder-layer stack as ONE replayed HIP graph (forward only).

Why: a prompt-sized forward (LLaVA-1.5: 1088 positions through 32 Llama layers) is ~300 kernel launches of 20-80 us; the
launch path of each -- dispatcher op, Python, ctypes -- costs as much host time as the kernel takes on the GPU, so the
stack is host-bound (profiles/r03a_llava_graph_ab.jsonl: 28.3 ms eager).  HIP graphs are the MI355X answer to launch-bound
inner loops.  The reference's own model code cannot be captured as a whole (LLaVA's `get_placeholder_mask` formats a
device scalar into an error message, models/llava/modeling_llava.py:205-213: a host sync), so the graph is taken where
this package owns the code: the decoder layers.  `LlamaModel.forward` (models/llama/modeling_llama.py:395-404) calls them
in a plain loop, `hidden_states = decoder_layer(hidden_states, ...)`; layer 0 runs the whole stack as a graph and hands
back the final hidden state, the other layers recognise that tensor (by identity) and pass it through.

Used when nothing is being differentiated, no KV cache is in play, no hook wants per-layer outputs and every layer's
weights live on the device of the hidden state (no `device_map` spread, no accelerate offload hooks); a shape is captured
the second time it is seen (a one-off forward stays eager); a signature whose capture failed is never tried again.
`TAMD_HIP_GRAPH=0` switches it off.

Aliasing contract: the tensor `run()` returns is the graph's static output buffer -- the NEXT replay of the same signature
overwrites it in place.  `LlamaModel.forward` feeds it straight into its final norm (a fresh tensor), which is the only
consumer this module serves; a caller that keeps the returned hidden state across two forwards must clone it.
"""
from __future__ import annotations

import os
import threading
from typing import Optional

import torch

from . import ops

_ENABLED = os.environ.get("TAMD_HIP_GRAPH", "1") != "0"
_MAX_GRAPHS = 4  # captured shapes kept per stack (each holds its own activation pool)
_MAX_SEEN = 64   # signatures remembered before capture (sightings / failures); beyond that the stack stays eager


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
        self._seen = {}     # key -> sightings before capture
        self._failed = set()  # keys whose capture raised: eager from then on
        self._lock = threading.Lock()  # capture / replay write the shared static buffers: one thread at a time
        self._pass = threading.local()  # the output handed out by layer 0, for the other layers to recognise
        self.replays = 0
        self.capture_failures = 0

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
        sig = self._weights_signature()
        if not self._placement_ok(hidden_states.device, sig):
            return None
        key = self._key(hidden_states, cos, sin, key_valid, q_start, sig)
        with self._lock:
            entry = self._graphs.get(key)
            if entry is None:
                if key in self._failed:
                    return None
                n = self._seen.get(key, 0)
                if n == 0 and len(self._seen) >= _MAX_SEEN:
                    return None  # (a stream of ever-new shapes: do not remember them all)
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
    def _placement_ok(self, device, sig) -> bool:
        """Every parameter of every layer on `device` and materialised, no accelerate hook on a layer: `_forward_all` calls
        the layer op directly, i.e. it bypasses the forward wrappers accelerate installs for `device_map` / offloading.
        (Cached on the weights' signature: their storage does not move without it changing.)"""
        cached = self.__dict__.get("_placement")
        if cached is not None and cached[0] == (sig, device):
            return cached[1]
        ok = True
        for layer in self.layers:
            if hasattr(layer, "_hf_hook") or any(p.device != device for p in layer.parameters()):
                ok = False  # (an offloaded / meta / other-GPU weight, or a wrapper that moves weights per call)
                break
        self.__dict__["_placement"] = ((sig, device), ok)
        return ok

    def _weights_signature(self):
        sig = []
        for layer in self.layers:
            attn, mlp = layer.self_attn, layer.mlp
            sig.append((attn._fused().weight().data_ptr(), attn.o_proj.weight.data_ptr(),
                        mlp._fused().weight().data_ptr(), mlp.down_proj.weight.data_ptr(),
                        layer.input_layernorm.weight.data_ptr(), layer.post_attention_layernorm.weight.data_ptr()))
        return hash(tuple(sig))

    def _key(self, h, cos, sin, key_valid, q_start, sig):
        return (tuple(h.shape), h.dtype, h.device.index, tuple(cos.shape), cos.dtype,
                None if key_valid is None else (tuple(key_valid.shape), key_valid.dtype),
                None if q_start is None else tuple(q_start.shape), sig)

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
            self._failed.add(key)
            self.capture_failures += 1
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
