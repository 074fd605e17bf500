"""MI355X execution path for the BERT encoder blocks (src/transformers/models/bert/modeling_bert.py).

Post-LN blocks with biases: `LayerNorm(dropout(dense(x)) + residual)`.  With dropout inactive (eval mode or p = 0)
the residual add rides in the GEMM epilogue; in train mode with hidden dropout the GEMM keeps the bias and one kernel
draws the dropout mask, adds the residual and normalises (`ops.dropout_add_layernorm`).  The masked-LM head (transform +
a 30522-wide tied decoder + CrossEntropyLoss) runs on the same GEMM through zero-padded weight rows and on the
cross-entropy kernels (`TamdBertLMPredictionHead`, `masked_lm_forward`).
"""
from __future__ import annotations

import torch
from transformers.models.bert import modeling_bert as ref

from .. import layer_ops, ops
from ..fused_params import FusedWeights, PaddedRows
from .common import _gpu, _has_hooks, _placement_ok, note_fallback


def _no_dropout(mod) -> bool:
    return (not mod.training) or mod.dropout.p == 0.0


class TamdBertSelfAttention(ref.BertSelfAttention):
    """BertSelfAttention.forward, modeling_bert.py:164-203: one fused QKV GEMM (+bias) and the flash kernel."""

    def _fused(self) -> FusedWeights:
        fw = self.__dict__.get("_tamd_qkv")
        if fw is None:
            fw = FusedWeights([self.query, self.key, self.value])
            self.__dict__["_tamd_qkv"] = fw
        return fw

    def forward(self, hidden_states, attention_mask=None, past_key_values=None, **kwargs):
        d = self.attention_head_size
        if not (_gpu(hidden_states) and past_key_values is None and d in (64, 128)
                and hidden_states.dtype in (torch.bfloat16, torch.float16)
                and self.config._attn_implementation == "tamd" and not kwargs.get("output_attentions", False)):
            note_fallback(self, hidden_states, "kv_cache" if past_key_values is not None else "unsupported")
            return super().forward(hidden_states, attention_mask=attention_mask, past_key_values=past_key_values,
                                   **kwargs)
        b, s, h = hidden_states.shape
        nh = self.num_attention_heads
        qkv = self._fused().linear(hidden_states)  # [B,S,3h]
        q = qkv[..., :h].view(b, s, nh, d)
        k = qkv[..., h:2 * h].view(b, s, nh, d)
        v = qkv[..., 2 * h:].view(b, s, nh, d)
        key_valid = None
        if attention_mask is not None:
            from ..attention import _key_valid_from_mask
            key_valid = _key_valid_from_mask(attention_mask, b, s)
        o = ops.attention(q, k, v, float(self.scaling), bool(self.is_causal) and s > 1, key_valid,
                          dropout_p=self.dropout.p if self.training else 0.0)
        return o.view(b, s, h), None


class _DenseResidualLN:
    """dense -> (+bias, +residual in the GEMM epilogue) -> LayerNorm: BertSelfOutput / BertOutput."""

    def _fast(self, hidden_states, input_tensor):
        if _no_dropout(self):
            y = ops.linear(hidden_states, self.dense.weight, self.dense.bias, residual=input_tensor)
            return ops.layernorm(y, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        # train mode with hidden dropout (the shipped configs): LayerNorm(dropout(dense(x)) + residual) -- the dropout
        # sits between the bias and the residual add, so the residual leaves the GEMM epilogue and joins the LayerNorm
        # kernel, which also draws the dropout mask (2 launches: GEMM+bias, dropout+add+LN; the reference path takes 5)
        y = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        return ops.dropout_add_layernorm(y, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                         self.LayerNorm.eps, self.dropout.p)

    def _ok(self, x):
        return _gpu(x) and x.dtype in (torch.bfloat16, torch.float16) and self.dense.bias is not None


class TamdBertSelfOutput(_DenseResidualLN, ref.BertSelfOutput):
    """modeling_bert.py:289-293"""

    def forward(self, hidden_states, input_tensor):
        if not self._ok(hidden_states):
            note_fallback(self, hidden_states)
            return ref.BertSelfOutput.forward(self, hidden_states, input_tensor)
        return self._fast(hidden_states, input_tensor)


class TamdBertOutput(_DenseResidualLN, ref.BertOutput):
    """modeling_bert.py:347-351"""

    def forward(self, hidden_states, input_tensor):
        if not self._ok(hidden_states):
            note_fallback(self, hidden_states)
            return ref.BertOutput.forward(self, hidden_states, input_tensor)
        return self._fast(hidden_states, input_tensor)


class TamdBertIntermediate(ref.BertIntermediate):
    """dense + GELU, modeling_bert.py:334-337: bias and activation live in the GEMM epilogue."""

    def forward(self, hidden_states):
        act = _act_name(self, "_tamd_act", self.intermediate_act_fn)
        if not (_gpu(hidden_states) and act in ops.ACT_CODES and ops.ACT_CODES[act] != ops.ACT_NONE
                and hidden_states.dtype in (torch.bfloat16, torch.float16) and self.dense.bias is not None):
            note_fallback(self, hidden_states)
            return super().forward(hidden_states)
        return ops.linear(hidden_states, self.dense.weight, self.dense.bias, act=ops.ACT_CODES[act])


class TamdBertLayer(ref.BertLayer):
    """BertLayer.forward, modeling_bert.py:374-416, as ONE dispatcher op / autograd node (torch.ops.tamd.bert_layer) for the
    encoder case: no cross-attention, no KV cache, no feed-forward chunking.  Everything else -- and any layer whose
    children are hooked or ask for attention weights -- takes the reference's forward over the replacement children."""

    def _fused_ok(self, hidden_states, past_key_values, kwargs) -> bool:
        att = self.attention
        sa, so, inter, out = att.self, att.output, self.intermediate, self.output
        return (type(sa) is TamdBertSelfAttention and type(so) is TamdBertSelfOutput
                and type(inter) is TamdBertIntermediate and type(out) is TamdBertOutput
                and _gpu(hidden_states) and hidden_states.dtype in (torch.bfloat16, torch.float16)
                and past_key_values is None and not self.is_decoder and self.chunk_size_feed_forward == 0
                and sa.attention_head_size in (64, 128) and sa.config._attn_implementation == "tamd"
                and not kwargs.get("output_attentions", False) and sa.query.bias is not None
                and so.dense.bias is not None and inter.dense.bias is not None and out.dense.bias is not None
                and sa.query.weight.dtype == hidden_states.dtype
                and ops.ACT_CODES.get(_act_name(inter, "_tamd_act", inter.intermediate_act_fn), ops.ACT_NONE) != ops.ACT_NONE
                and so.dropout.p == out.dropout.p
                and not _has_hooks(att, sa, so, inter, out, so.LayerNorm, out.LayerNorm)
                and _placement_ok(self, hidden_states.device, sa.query.weight, out.dense.weight))

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_values=None, **kwargs):
        if encoder_hidden_states is not None or not self._fused_ok(hidden_states, past_key_values, kwargs):
            # (the children are replacement classes themselves: this level only loses the epilogue fusions)
            note_fallback(self, hidden_states, "kv_cache" if past_key_values is not None else "layer_unfused")
            return super().forward(hidden_states, attention_mask=attention_mask,
                                   encoder_hidden_states=encoder_hidden_states,
                                   encoder_attention_mask=encoder_attention_mask, past_key_values=past_key_values, **kwargs)
        att = self.attention
        sa, so, inter, out = att.self, att.output, self.intermediate, self.output
        b, s, _ = hidden_states.shape
        key_valid = None
        if attention_mask is not None:
            from ..attention import _key_valid_from_mask
            key_valid = _key_valid_from_mask(attention_mask, b, s)
        fw = sa._fused()
        wqkv, bqkv = fw.weight(), fw.bias()
        train = self.training
        return layer_ops.bert_layer(
            hidden_states, key_valid, wqkv, bqkv,
            (sa.query.weight, sa.key.weight, sa.value.weight, sa.query.bias, sa.key.bias, sa.value.bias),
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, inter.dense.weight, inter.dense.bias,
            out.dense.weight, out.dense.bias, out.LayerNorm.weight, out.LayerNorm.bias, eps=so.LayerNorm.eps,
            heads=sa.num_attention_heads, d=sa.attention_head_size, scale=sa.scaling,
            act=ops.ACT_CODES[_act_name(inter, "_tamd_act", inter.intermediate_act_fn)],
            p_attn=sa.dropout.p if train else 0.0, p_hidden=so.dropout.p if train else 0.0)


class TamdBertEmbeddings(ref.BertEmbeddings):
    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None,
                past_key_values_length=0):
        w = self.word_embeddings.weight
        if not (input_ids is not None and inputs_embeds is None and _gpu(w)
                and w.shape[1] % 8 == 0 and w.shape[1] <= 4096
                and w.dtype in (torch.bfloat16, torch.float16, torch.float32)):
            note_fallback(self, w)
            return super().forward(input_ids=input_ids, token_type_ids=token_type_ids, position_ids=position_ids,
                                   inputs_embeds=inputs_embeds, past_key_values_length=past_key_values_length)
        b, s = input_ids.shape
        if position_ids is None:
            position_ids = self.position_ids[:, past_key_values_length: s + past_key_values_length]
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        position_ids = position_ids.expand(b, s)
        token_type_ids = token_type_ids.expand(b, s)
        out = ops.bert_embeddings(input_ids, token_type_ids, position_ids, w, self.token_type_embeddings.weight,
                                  self.position_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias,
                                  self.LayerNorm.eps, self.word_embeddings.padding_idx)
        return out if _no_dropout(self) else torch.nn.functional.dropout(out, self.dropout.p, True)


def _act_name(mod, attr: str, fn) -> str:
    """The reference modules keep only the activation callable: recover its ACT2FN name (cached on the module)."""
    act = mod.__dict__.get(attr)
    if act is None:
        from transformers.activations import ACT2FN

        act = next((k for k in ("gelu", "gelu_new", "quick_gelu", "silu", "gelu_pytorch_tanh")
                    if type(ACT2FN[k]) is type(fn)), "")
        mod.__dict__[attr] = act
    return act


class TamdBertPredictionHeadTransform(ref.BertPredictionHeadTransform):
    """dense -> activation -> LayerNorm, modeling_bert.py:466-480: bias + activation in the GEMM epilogue."""

    def forward(self, hidden_states):
        act = _act_name(self, "_tamd_act", self.transform_act_fn)
        if not (_gpu(hidden_states) and ops.ACT_CODES.get(act, ops.ACT_NONE) != ops.ACT_NONE
                and hidden_states.dtype in (torch.bfloat16, torch.float16) and self.dense.bias is not None
                and self.dense.weight.dtype == hidden_states.dtype):
            note_fallback(self, hidden_states)
            return super().forward(hidden_states)
        y = ops.linear(hidden_states, self.dense.weight, self.dense.bias, act=ops.ACT_CODES[act])
        return ops.layernorm(y, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)


class TamdBertLMPredictionHead(ref.BertLMPredictionHead):
    """transform -> decoder, modeling_bert.py:483-496.  The decoder is [vocab, hidden] tied to the word embeddings with
    vocab = 30522 for bert-base: not a multiple of 8, so it lives in zero-padded rows (fused_params.PaddedRows) and the
    scores come back as the [.., vocab] view of a [.., vocab_pad] buffer."""

    def _padded(self) -> PaddedRows:
        pr = self.__dict__.get("_tamd_padded")
        if pr is None or pr.linear is not self.decoder:
            pr = PaddedRows(self.decoder)
            self.__dict__["_tamd_padded"] = pr
        return pr

    def _fast_ok(self, h) -> bool:
        w = self.decoder.weight
        return (_gpu(h) and h.dtype in (torch.bfloat16, torch.float16) and w.dtype == h.dtype and w.shape[1] % 8 == 0
                and h.numel() > 0)

    def scores_and_loss(self, hidden_states, labels=None):
        """-> (masked-LM loss or None, prediction scores): `CrossEntropyLoss()` of modeling_bert.py:973-975 on the
        cross-entropy kernels (fp32 in registers) when labels are given."""
        h = self.transform(hidden_states)
        w_pad, b_pad = self._padded().buffers()
        return ops.padded_vocab_head(h, w_pad, b_pad, self.decoder.weight, self.decoder.bias, labels)

    def forward(self, hidden_states):
        if not self._fast_ok(hidden_states):
            note_fallback(self, hidden_states)
            return super().forward(hidden_states)
        return self.scores_and_loss(hidden_states)[1]


def masked_lm_forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                      inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None, **kwargs):
    """BertForMaskedLM.forward (modeling_bert.py:939-982) with the loss on the cross-entropy kernels.  Installed on the
    INSTANCE by `accelerate` (the class, `config.architectures` and the reference's registries stay untouched; `revert`
    removes it).  Same outputs as the reference: the loss AND the full prediction scores."""
    head = self.cls.predictions
    if labels is None or not isinstance(head, TamdBertLMPredictionHead):
        return type(self).forward(self, input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                                  position_ids=position_ids, inputs_embeds=inputs_embeds,
                                  encoder_hidden_states=encoder_hidden_states,
                                  encoder_attention_mask=encoder_attention_mask, labels=labels, **kwargs)
    from transformers.modeling_outputs import MaskedLMOutput

    return_dict = kwargs.pop("return_dict", None)
    outputs = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                        position_ids=position_ids, inputs_embeds=inputs_embeds,
                        encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=encoder_attention_mask,
                        return_dict=True, **kwargs)
    sequence_output = outputs[0]
    if not head._fast_ok(sequence_output):
        note_fallback(self, sequence_output, "mlm_loss")
        scores = self.cls(sequence_output)
        loss = torch.nn.functional.cross_entropy(scores.view(-1, self.config.vocab_size), labels.view(-1))
    else:
        loss, scores = head.scores_and_loss(sequence_output, labels)
    out = MaskedLMOutput(loss=loss, logits=scores, hidden_states=outputs.hidden_states, attentions=outputs.attentions)
    return out.to_tuple() if return_dict is False else out


REPLACEMENTS = {
    ref.BertLayer: TamdBertLayer,
    ref.BertPredictionHeadTransform: TamdBertPredictionHeadTransform,
    ref.BertLMPredictionHead: TamdBertLMPredictionHead,
    ref.BertSelfAttention: TamdBertSelfAttention,
    ref.BertSelfOutput: TamdBertSelfOutput,
    ref.BertOutput: TamdBertOutput,
    ref.BertIntermediate: TamdBertIntermediate,
    ref.BertEmbeddings: TamdBertEmbeddings,
}
