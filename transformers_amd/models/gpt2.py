"""GPT-2 (src/transformers/models/gpt2/modeling_gpt2.py, src/transformers/pytorch_utils.py:117-121).

BASELINE config 1 is GPT-2 on CPU through AutoModelForCausalLM: that path must stay the reference's own
(CPU tensors never enter our kernels).  On a GPU the only GPT-2 specific piece is `Conv1D`, whose weight is
stored [in, out]: exactly the k-major B operand of the MFMA GEMM, so no transpose is ever made."""
from __future__ import annotations

import torch
from transformers.pytorch_utils import Conv1D

from .. import ops
from .common import _gpu


class Conv1DFn(torch.autograd.Function):
    """y = x @ W + b with W [in, out]  (pytorch_utils.py:117-121: torch.addmm(bias, x, weight))."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = ops._c(x).view(-1, x.shape[-1])
        y = ops.raw_gemm(x2, w, b_kn=True, bias=b, epilogue=ops.EPI_BIAS if b is not None else ops.EPI_NONE)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], w.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = ops._c(dy).view(-1, w.shape[1])
        dx = ops.raw_gemm(dy2, w).view(ctx.x_shape)                 # dX = dY . W^T  (W is [N=in, K=out])
        dw = ops.raw_gemm(x2, dy2, a_km=True, b_kn=True)            # dW[in,out] = X^T . dY
        db = ops.raw_colsum(dy2) if ctx.has_bias else None
        return dx, dw, db


class TamdConv1D(Conv1D):
    def forward(self, x):
        w = self.weight
        if (_gpu(x) and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype
                and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0 and x.numel() > 0):
            return Conv1DFn.apply(x, w, self.bias)
        return super().forward(x)


REPLACEMENTS = {Conv1D: TamdConv1D}
