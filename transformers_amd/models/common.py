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


class TamdLinear(nn.Linear):
    def forward(self, x):
        w = self.weight
        if (_gpu(x) and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and x.numel() > 0
                and ops.gemm_supported(x.numel() // x.shape[-1], w.shape[0], w.shape[1], x.dtype)):
            return ops.linear(x, w, self.bias)
        return super().forward(x)


class TamdEmbedding(nn.Embedding):
    def forward(self, ids):
        w = self.weight
        if (_gpu(w) and w.dtype in (torch.bfloat16, torch.float16, torch.float32) and self.max_norm is None
                and not self.sparse and not self.scale_grad_by_freq and w.shape[1] % 8 == 0
                and ids.dtype in (torch.int64, torch.int32)):
            return ops.embedding(ids, w, self.padding_idx)
        return super().forward(ids)


class TamdLayerNorm(nn.LayerNorm):
    def forward(self, x):
        if (_gpu(x) and self.elementwise_affine and len(self.normalized_shape) == 1
                and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.shape[-1] % 8 == 0
                and x.shape[-1] <= 8192 and self.weight.dtype == x.dtype):
            return ops.layernorm(x, self.weight, self.bias, self.eps)
        return super().forward(x)


REPLACEMENTS = {nn.Linear: TamdLinear, nn.Embedding: TamdEmbedding, nn.LayerNorm: TamdLayerNorm}
