"""MI355X execution path for the CLIP vision/text encoder blocks (src/transformers/models/clip/modeling_clip.py),
used by LLaVA's vision tower (BASELINE config 5).  Pre-LN blocks with biases and `quick_gelu`.

`CLIPEncoderLayer` stays a hookable module with a tensor output (LLaVA reads hidden_states[-2] through the
output-capturing hooks, models/llava/modeling_llava.py:154-166), so fusion stops at the layer boundary."""
from __future__ import annotations

import torch
from transformers.models.clip import modeling_clip as ref

from .. import ops
from ..fused_params import FusedWeights
from .common import _gpu, _has_hooks, note_fallback


class TamdCLIPAttention(ref.CLIPAttention):
    """CLIPAttention.forward, modeling_clip.py:298-335: fused QKV GEMM (+bias), flash kernel, out_proj."""

    def _fused(self) -> FusedWeights:
        fw = self.__dict__.get("_tamd_qkv")
        if fw is None:
            fw = FusedWeights([self.q_proj, self.k_proj, self.v_proj])
            self.__dict__["_tamd_qkv"] = fw
        return fw

    def forward(self, hidden_states, attention_mask=None, **kwargs):
        d = self.head_dim
        if not (_gpu(hidden_states) and d in (64, 128) and hidden_states.dtype in (torch.bfloat16, torch.float16)
                and self.config._attn_implementation == "tamd"
                and not kwargs.get("output_attentions", False)):
            note_fallback(self, hidden_states)
            return super().forward(hidden_states, attention_mask=attention_mask, **kwargs)
        b, s, h = hidden_states.shape
        nh = self.num_heads
        qkv = self._fused().linear(hidden_states)
        q = qkv[..., :h].view(b, s, nh, d)
        k = qkv[..., h:2 * h].view(b, s, nh, d)
        v = qkv[..., 2 * h:].view(b, s, nh, d)
        key_valid = None
        if attention_mask is not None:
            from ..attention import _key_valid_from_mask
            key_valid = _key_valid_from_mask(attention_mask, b, s)
        causal = bool(kwargs.get("is_causal", self.is_causal)) and s > 1
        o = ops.attention(q, k, v, float(self.scale), causal, key_valid,
                          dropout_p=self.dropout if self.training else 0.0)
        return ops.linear(o.view(b, s, h), self.out_proj.weight, self.out_proj.bias), None


class TamdCLIPMLP(ref.CLIPMLP):
    """CLIPMLP.forward, modeling_clip.py:346-350: fc1 + bias + activation in one GEMM epilogue, then fc2."""

    def forward(self, hidden_states):
        act = self.config.hidden_act
        if not (_gpu(hidden_states) and isinstance(act, str) and ops.ACT_CODES.get(act, 0) != ops.ACT_NONE
                and hidden_states.dtype in (torch.bfloat16, torch.float16) and self.fc1.bias is not None):
            note_fallback(self, hidden_states)
            return super().forward(hidden_states)
        hmid = ops.linear(hidden_states, self.fc1.weight, self.fc1.bias, act=ops.ACT_CODES[act])
        return ops.linear(hmid, self.fc2.weight, self.fc2.bias)


class TamdCLIPEncoderLayer(ref.CLIPEncoderLayer):
    """CLIPEncoderLayer.forward, modeling_clip.py:362-383: residual adds folded into the out_proj / fc2 GEMMs."""

    def forward(self, hidden_states, attention_mask, **kwargs):
        attn, mlp = self.self_attn, self.mlp
        x = hidden_states
        act = mlp.config.hidden_act
        if not (isinstance(attn, TamdCLIPAttention) and isinstance(mlp, TamdCLIPMLP) and _gpu(x)
                and x.dtype in (torch.bfloat16, torch.float16) and attn.head_dim in (64, 128)
                and attn.config._attn_implementation == "tamd" and not (self.training and attn.dropout > 0)
                and isinstance(act, str) and ops.ACT_CODES.get(act, 0) != ops.ACT_NONE
                and not (_has_hooks(attn, mlp) or kwargs.get("output_attentions", False))):
            # (the children are replacement classes themselves: this level only loses the epilogue fusions, and a hooked
            # layer -- LLaVA reads hidden states through hooks on the LAYER, which do not land here -- is by design)
            note_fallback(self, x, "layer_unfused")
            return super().forward(hidden_states, attention_mask, **kwargs)
        b, s, h = x.shape
        nh, d = attn.num_heads, attn.head_dim
        y = ops.layernorm(x, self.layer_norm1.weight, self.layer_norm1.bias, self.layer_norm1.eps)
        qkv = attn._fused().linear(y)
        key_valid = None
        if attention_mask is not None:
            from ..attention import _key_valid_from_mask
            key_valid = _key_valid_from_mask(attention_mask, b, s)
        causal = bool(kwargs.get("is_causal", attn.is_causal)) and s > 1
        o = ops.attention(qkv[..., :h].view(b, s, nh, d), qkv[..., h:2 * h].view(b, s, nh, d),
                          qkv[..., 2 * h:].view(b, s, nh, d), float(attn.scale), causal, key_valid)
        x = ops.linear(o.view(b, s, h), attn.out_proj.weight, attn.out_proj.bias, residual=x)
        y = ops.layernorm(x, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps)
        hmid = ops.linear(y, mlp.fc1.weight, mlp.fc1.bias, act=ops.ACT_CODES[act])
        return ops.linear(hmid, mlp.fc2.weight, mlp.fc2.bias, residual=x)


class TamdCLIPVisionEmbeddings(ref.CLIPVisionEmbeddings):
    """CLIPVisionEmbeddings.forward, modeling_clip.py:209-218: the patch embedding is a convolution whose stride equals its
    kernel (14 x 14 patches of a 336 x 336 image, no bias, :148-154) -- i.e. a GEMM over non-overlapping patches:
        patches [B * 576, 3*14*14 = 588]  .  weight [1024, 588]^T
    with K zero-padded to a multiple of 64 (640) on both operands so that the MFMA kernels' 64-deep stages apply.  One
    permuting copy of the pixels replaces the convolution (MIOpen's `naive_conv_ab_nonpacked_fwd_nchw`: 0.34 ms of a 21 ms
    LLaVA-1.5-7B forward, profiles/r03r_llava_kernel_stats.csv); the class token, the concatenation and the position
    embedding stay the reference's own lines.  Training a tower (round 5): when a gradient is wanted the padded weight is built
    by a differentiable `pad` instead of taken from the cache, so dW arrives through the GEMM's own backward (dY^T . patches, cut
    back to the 588 real columns by autograd) and d(pixels), if anybody asks, through the copy's."""

    def _padded_weight(self):
        w = self.patch_embedding.weight
        key = (w.data_ptr(), w._version, w.dtype, w.device)
        cached = self.__dict__.get("_tamd_patch_w")
        if cached is None or cached[0] != key:
            n, k = w.shape[0], w[0].numel()
            kp = -(-k // 64) * 64
            wp = w.new_zeros(n, kp)
            wp[:, :k].copy_(w.detach().reshape(n, k))
            cached = (key, wp)
            self.__dict__["_tamd_patch_w"] = cached
        return cached[1]

    def forward(self, pixel_values, interpolate_pos_encoding=False):
        w = self.patch_embedding.weight
        p = self.patch_size
        b, c, height, width = pixel_values.shape
        fast = (_gpu(pixel_values) and w.dtype in (torch.bfloat16, torch.float16) and self.patch_embedding.bias is None
                and height % p == 0 and width % p == 0 and w.shape[0] % 8 == 0
                and tuple(self.patch_embedding.stride) == (p, p) and tuple(self.patch_embedding.kernel_size) == (p, p)
                and tuple(self.patch_embedding.padding) == (0, 0) and self.patch_embedding.groups == 1
                and not (not interpolate_pos_encoding and (height != self.image_size or width != self.image_size)))
        if not fast:
            note_fallback(self, pixel_values)
            return super().forward(pixel_values, interpolate_pos_encoding=interpolate_pos_encoding)
        gh, gw = height // p, width // p
        k = c * p * p
        if torch.is_grad_enabled() and w.requires_grad:  # a differentiable padded copy (1024 x 640: 1.3 MB)
            wp = torch.nn.functional.pad(w.reshape(w.shape[0], k), (0, -(-k // 64) * 64 - k))
        else:
            wp = self._padded_weight()
        patches = pixel_values.new_zeros((b * gh * gw, wp.shape[1]), dtype=w.dtype)
        # [B, C, gh, p, gw, p] -> [B, gh, gw, C, p, p]: one copy (with the dtype cast of `pixel_values.to(target_dtype)`)
        patches[:, :k].view(b, gh, gw, c, p, p).copy_(pixel_values.view(b, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5))
        patch_embeds = ops.linear(patches, wp).view(b, gh * gw, w.shape[0])  # = conv(...).flatten(2).transpose(1, 2)
        class_embeds = self.class_embedding.expand(b, 1, -1)
        embeddings = torch.cat([class_embeds, patch_embeds], dim=1)
        if interpolate_pos_encoding:
            return embeddings + self.interpolate_pos_encoding(embeddings, height, width)
        return embeddings + self.position_embedding(self.position_ids)


REPLACEMENTS = {
    ref.CLIPVisionEmbeddings: TamdCLIPVisionEmbeddings,
    ref.CLIPAttention: TamdCLIPAttention,
    ref.CLIPMLP: TamdCLIPMLP,
    ref.CLIPEncoderLayer: TamdCLIPEncoderLayer,
}
