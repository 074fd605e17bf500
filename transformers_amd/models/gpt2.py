"""GPT-2 (src/transformers/models/gpt2/modeling_gpt2.py, src/transformers/pytorch_utils.py:117-121).

BASELINE config 1 is GPT-2 on CPU through AutoModelForCausalLM: that path must stay the reference's own
(CPU tensors never enter our kernels).  On a GPU the only GPT-2 specific piece is `Conv1D`, whose weight is
stored [in, out]: exactly the k-major B operand of the MFMA GEMM, so no transpose is ever made."""
from __future__ import annotations

import torch
from transformers.pytorch_utils import Conv1D

from .. import ops
from .common import _gpu, note_fallback


class TamdConv1D(Conv1D):
    def forward(self, x):
        w = self.weight
        if (_gpu(x) and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype
                and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0 and x.numel() > 0):
            return ops.conv1d(x, w, self.bias)  # torch.ops.tamd.conv1d
        note_fallback(self, x)
        return super().forward(x)


REPLACEMENTS = {Conv1D: TamdConv1D}
