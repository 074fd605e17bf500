"""MI355X execution path for the Llama blocks (src/transformers/models/llama/modeling_llama.py).

Replacement classes subclass the reference classes (SURVEY.md §8b "B2" invariants: isinstance-based
output recorders and gradient checkpointing keep working, `RMSNorm` stays in the class name, parameter
names / shapes / state_dict keys are untouched) and are swapped in post-construction by
`transformers_amd.accelerate` (`module.__class__ = ...`).

Fusion levels, outermost first -- each falls back to the next when its preconditions fail:
  TamdLlamaDecoderLayer   whole layer as ONE op / autograd node (torch.ops.tamd.llama_layer): 4 GEMMs, attention, 2 norms,
                          rope and SwiGLU launches forward; residual adds live in GEMM epilogues.
  TamdLlamaAttention / TamdLlamaMLP / TamdLlamaRMSNorm   module-level replacements (fused QKV, fused gate|up).
CPU tensors always take the reference's own forward (config 1, GPT-2 on CPU, must run unchanged).
"""
from __future__ import annotations

import torch
from torch import nn
from transformers.models.llama import modeling_llama as ref

from .. import layer_ops, ops
from ..fused_params import FusedWeights


from .common import _has_hooks, _placement_ok, note_fallback  # noqa: E402


def _on_gpu(t: torch.Tensor) -> bool:
    return t.is_cuda or ops.backend_is_emulated()


class TamdLlamaRMSNorm(ref.LlamaRMSNorm):
    """LlamaRMSNorm.forward, modeling_llama.py:62-67, on the row-reduction kernel."""

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        if not _on_gpu(hidden_states):
            return super().forward(hidden_states)  # (CPU tensors: the reference's own path, by contract)
        return ops.rmsnorm(hidden_states, self.weight, self.variance_epsilon)


class TamdLlamaMLP(ref.LlamaMLP):
    """LlamaMLP.forward, modeling_llama.py:174-176: one fused gate|up GEMM, SwiGLU kernel, down GEMM."""

    def _fused(self) -> FusedWeights:
        fw = self.__dict__.get("_tamd_gate_up")
        if fw is None:
            fw = FusedWeights([self.gate_proj, self.up_proj])
            self.__dict__["_tamd_gate_up"] = fw
        return fw

    def forward(self, x):
        w = self.gate_proj.weight
        if (not _on_gpu(x) or self.config.hidden_act not in ("silu", "swish") or self.gate_proj.bias is not None
                or self.down_proj.bias is not None or x.dtype not in (torch.bfloat16, torch.float16)
                or w.dtype != x.dtype or w.shape[0] % 8 or w.shape[1] % 8):
            note_fallback(self, x)
            return super().forward(x)
        gu = self._fused().linear(x)
        act = ops.swiglu(gu)
        return ops.linear(act, self.down_proj.weight)


class TamdLlamaAttention(ref.LlamaAttention):
    """LlamaAttention.forward, modeling_llama.py:243-281: fused QKV GEMM + rotary + flash attention + o_proj."""

    def _fused(self) -> FusedWeights:
        fw = self.__dict__.get("_tamd_qkv")
        if fw is None:
            fw = FusedWeights([self.q_proj, self.k_proj, self.v_proj])
            self.__dict__["_tamd_qkv"] = fw
        return fw

    def _fast_ok(self, hidden_states, past_key_values) -> bool:
        return (_on_gpu(hidden_states) and past_key_values is None and self.q_proj.bias is None
                and self.head_dim in (64, 128) and hidden_states.dtype in (torch.bfloat16, torch.float16)
                and self.config._attn_implementation == "tamd")

    def _cached_ok(self, hidden_states) -> bool:
        """A forward with a KV cache (prefill into it, or a decode step) on the kernels: inference only."""
        return (self._fast_ok(hidden_states, None) and self.o_proj.bias is None
                and not (torch.is_grad_enabled() and (hidden_states.requires_grad or self.q_proj.weight.requires_grad)))

    def cached_forward(self, hidden_states, position_embeddings, attention_mask, past_key_values, residual=None, **kwargs):
        """LlamaAttention.forward with `past_key_values` (modeling_llama.py:254-281): ONE fused q|k|v product (the
        weight-streaming kernel of csrc/gemv.hip at one token per sequence), the rotary kernel, the reference's own
        `Cache.update` (cache_utils.py: the cache stays the reference's object), the registered attention function (split-KV
        decode kernel for seq_q <= 16) and o_proj -- with `residual` added in its epilogue when the caller is our layer."""
        from ..attention import tamd_attention_forward

        b, s, _ = hidden_states.shape
        hq, hkv, d = self.config.num_attention_heads, self.config.num_key_value_heads, self.head_dim
        cos, sin = position_embeddings
        qkv = self._fused().linear(hidden_states)                       # [B, S, (Hq+2Hkv)*D], nobody else's: rotated in place
        ops.rope_inplace(qkv.view(b * s, qkv.shape[-1]), cos, sin, s, hq + hkv, d)
        q = qkv[..., : hq * d].view(b, s, hq, d).transpose(1, 2)        # the reference's [B, H, S, D] views
        k = qkv[..., hq * d: (hq + hkv) * d].view(b, s, hkv, d).transpose(1, 2)
        v = qkv[..., (hq + hkv) * d:].view(b, s, hkv, d).transpose(1, 2)
        k, v = past_key_values.update(k, v, self.layer_idx)
        o, _ = tamd_attention_forward(self, q, k, v, attention_mask, dropout=0.0, scaling=self.scaling, **kwargs)
        return ops.linear(o.reshape(b, s, hq * d), self.o_proj.weight, residual=residual)

    def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
        if past_key_values is not None and self._cached_ok(hidden_states):
            return self.cached_forward(hidden_states, position_embeddings, attention_mask, past_key_values, **kwargs), None
        if not self._fast_ok(hidden_states, past_key_values):
            # (a cache under autograd, biases, other head sizes: the reference module -- projections through the TamdLinear
            # children, cache update -- around the registered attention function, counted under its own reason)
            note_fallback(self, hidden_states, "kv_cache" if past_key_values is not None else "unsupported")
            return super().forward(hidden_states, position_embeddings=position_embeddings,
                                   attention_mask=attention_mask, past_key_values=past_key_values, **kwargs)
        b, s, _ = hidden_states.shape
        hq = self.config.num_attention_heads
        hkv = self.config.num_key_value_heads
        d = self.head_dim
        cos, sin = position_embeddings
        qkv = self._fused().linear(hidden_states)                       # [B, S, (Hq+2Hkv)*D]
        qkv = ops.rope(qkv, cos, sin, hq + hkv, d)
        q = qkv[..., : hq * d].view(b, s, hq, d)
        k = qkv[..., hq * d: (hq + hkv) * d].view(b, s, hkv, d)
        v = qkv[..., (hq + hkv) * d:].view(b, s, hkv, d)
        from ..attention import split_mask
        key_valid, q_start = split_mask(attention_mask, b, s)
        if kwargs.get("cu_seq_lens_q") is not None:
            from ..attention import varlen_q_start
            q_start = varlen_q_start(q_start, kwargs, b, s, s, bool(self.is_causal) and s > 1)
        o = ops.attention(q, k, v, float(self.scaling), bool(self.is_causal) and s > 1, key_valid,
                          dropout_p=self.attention_dropout if self.training else 0.0, q_start=q_start)
        out = ops.linear(o.view(b, s, hq * d), self.o_proj.weight)
        return out, None


class TamdLlamaDecoderLayer(ref.LlamaDecoderLayer):
    """LlamaDecoderLayer.forward, modeling_llama.py:295-324."""

    def _fused_ok(self, hidden_states, past_key_values) -> bool:
        attn, mlp = self.self_attn, self.mlp
        return (isinstance(attn, TamdLlamaAttention) and isinstance(mlp, TamdLlamaMLP)
                and attn._fast_ok(hidden_states, past_key_values)
                and not (attn.training and attn.attention_dropout > 0)  # the per-module path carries dropout
                and mlp.config.hidden_act in ("silu", "swish") and mlp.gate_proj.bias is None
                and attn.o_proj.bias is None
                and not _has_hooks(attn, mlp, self.input_layernorm, self.post_attention_layernorm)
                and _placement_ok(self, hidden_states.device, attn.q_proj.weight, mlp.down_proj.weight))

    def _cached_ok(self, hidden_states) -> bool:
        attn, mlp = self.self_attn, self.mlp
        return (isinstance(attn, TamdLlamaAttention) and isinstance(mlp, TamdLlamaMLP) and attn._cached_ok(hidden_states)
                and mlp.config.hidden_act in ("silu", "swish") and mlp.gate_proj.bias is None and mlp.down_proj.bias is None
                and mlp.gate_proj.weight.dtype == hidden_states.dtype
                and not _has_hooks(attn, mlp, self.input_layernorm, self.post_attention_layernorm)
                # (the cached path reads the children's weights directly: an accelerate `device_map` / offload wrapper on a
                # child would not run -- ADVICE r4; such layers take the reference module, whose children are called)
                and _placement_ok(self, hidden_states.device, attn.q_proj.weight, mlp.down_proj.weight))

    def cached_forward(self, hidden_states, attention_mask, past_key_values, position_embeddings, **kwargs):
        """The layer with a KV cache (modeling_llama.py:303-324), prefill or decode: both residual adds ride in the epilogues
        of o_proj / down_proj, SiLU(gate) * up inside the gate|up product; 10 launches per decode step and layer (the reference path: ~25)."""
        attn, mlp = self.self_attn, self.mlp
        x = ops.rmsnorm(hidden_states, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        h = attn.cached_forward(x, position_embeddings, attention_mask, past_key_values, residual=hidden_states, **kwargs)
        x = ops.rmsnorm(h, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon)
        wgu = mlp._fused().weight()
        x2 = x.view(-1, x.shape[-1])
        if ops.gemm_swiglu_supported(x2, wgu):  # SiLU(gate) * up inside the product (the streaming kernel at M = batch)
            act = ops.linear_swiglu(x2, wgu).view(*x.shape[:-1], -1)
        else:
            act = ops.swiglu(mlp._fused().linear(x))
        return ops.linear(act, mlp.down_proj.weight, residual=h)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                position_embeddings=None, **kwargs):
        if past_key_values is not None and self._cached_ok(hidden_states):
            return self.cached_forward(hidden_states, attention_mask, past_key_values, position_embeddings,
                                       position_ids=position_ids, **kwargs)
        if not self._fused_ok(hidden_states, past_key_values):
            # (the children are replacement classes: this level only loses the epilogue fusions)
            note_fallback(self, hidden_states, "kv_cache" if past_key_values is not None else "layer_unfused")
            return super().forward(hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                   past_key_values=past_key_values, use_cache=use_cache,
                                   position_embeddings=position_embeddings, **kwargs)
        attn, mlp = self.self_attn, self.mlp
        b, s, _ = hidden_states.shape
        cos, sin = position_embeddings
        stack = self.__dict__.get("_tamd_stack")  # (graph_stack.LlamaStackGraph, index of this layer), set by accelerate()
        if stack is not None and stack[1] > 0 and stack[0].passthrough(hidden_states):
            return hidden_states  # layer 0 replayed the whole stack as one HIP graph: this IS the final hidden state
        from ..attention import split_mask
        key_valid, q_start = split_mask(attention_mask, b, s)
        if kwargs.get("cu_seq_lens_q") is not None:  # the reference's varlen kwargs for a flattened batch
            from ..attention import varlen_q_start
            q_start = varlen_q_start(q_start, kwargs, b, s, s, bool(attn.is_causal) and s > 1)
        if stack is not None and stack[1] == 0 and not self.training:
            out = stack[0].run(hidden_states, cos, sin, key_valid, q_start)
            if out is not None:
                return out
        qkv, gu = attn._fused(), mlp._fused()
        return layer_ops.llama_layer(
            hidden_states, cos, sin, key_valid, q_start, self.input_layernorm.weight, qkv.weight(),
            attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight, attn.o_proj.weight,
            self.post_attention_layernorm.weight, gu.weight(), mlp.gate_proj.weight, mlp.up_proj.weight,
            mlp.down_proj.weight, eps=self.input_layernorm.variance_epsilon, hq=attn.config.num_attention_heads,
            hkv=attn.config.num_key_value_heads, d=attn.head_dim, scale=attn.scaling,
            causal=bool(attn.is_causal) and s > 1)


def fused_causal_lm_forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                            inputs_embeds=None, labels=None, use_cache=None, logits_to_keep=0, **kwargs):
    """LlamaForCausalLM.forward (modeling_llama.py: `logits = self.lm_head(...)` then `self.loss_function(...)`) with the
    lm_head GEMM and the causal-LM loss fused chunk by chunk.  Installed on the INSTANCE by
    `accelerate(model, fused_lm_head_loss=True)` -- the class, its name in `config.architectures` and the reference's
    per-class registries stay untouched.  Used when labels are given in training mode: the output carries the loss and
    `logits=None` (nothing of shape [tokens, vocab] is ever allocated); every other call is the reference's forward."""
    w = self.lm_head.weight
    fused = (labels is not None and self.training and self.lm_head.bias is None and w.shape[0] % 8 == 0
             and w.shape[1] % 8 == 0 and w.dtype in (torch.bfloat16, torch.float16) and _on_gpu(w)
             and isinstance(logits_to_keep, int) and logits_to_keep == 0)
    if not fused:
        if labels is not None and self.training:
            note_fallback(self, w, "lm_head_loss_unfused")
        return type(self).forward(self, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                  past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                  use_cache=use_cache, logits_to_keep=logits_to_keep, **kwargs)
    from transformers.modeling_outputs import CausalLMOutputWithPast

    outputs = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                         past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache, **kwargs)
    loss = ops.fused_linear_cross_entropy(outputs.last_hidden_state, w, labels,
                                          num_items_in_batch=kwargs.get("num_items_in_batch"))
    return CausalLMOutputWithPast(loss=loss, logits=None, past_key_values=outputs.past_key_values,
                                  hidden_states=outputs.hidden_states, attentions=outputs.attentions)


REPLACEMENTS = {
    ref.LlamaRMSNorm: TamdLlamaRMSNorm,
    ref.LlamaMLP: TamdLlamaMLP,
    ref.LlamaAttention: TamdLlamaAttention,
    ref.LlamaDecoderLayer: TamdLlamaDecoderLayer,
}
