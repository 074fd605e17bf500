"""Fused projection weights without touching parameter names.

A fused QKV (or gate|up) GEMM wants ONE [sum(out_i), in] weight, but `state_dict` keys (`q_proj.weight`, ...),
DDP bucket order, tied weights and `save_pretrained` must keep seeing the reference's separate parameters
(SURVEY.md §7 "Parameter-layout vs drop-in", §8b invariant 3).  So each parameter keeps its identity and
shape and its `.data` becomes a row-slice VIEW of one fused buffer: optimizers, `load_state_dict`, DDP
broadcast all write through to the fused storage; the GEMM reads the fused buffer; the backward computes
one fused dW and hands each parameter its row-slice.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import ops


class FusedWeights:
    """Row-concatenation of several nn.Linear weights (and biases) kept coherent with the parameters."""

    def __init__(self, linears: List[nn.Linear]):
        self.linears = list(linears)
        self._w: Optional[torch.Tensor] = None
        self._b: Optional[torch.Tensor] = None
        self.has_bias = self.linears[0].bias is not None

    def _coherent(self) -> bool:
        if self._w is None:
            return False
        off = 0
        w0 = self.linears[0].weight
        if self._w.device != w0.device or self._w.dtype != w0.dtype:
            return False
        esz = self._w.element_size()
        row = self._w.shape[1] * esz
        for lin in self.linears:
            if lin.weight.data_ptr() != self._w.data_ptr() + off * row or not lin.weight.is_contiguous():
                return False
            off += lin.weight.shape[0]
        if self.has_bias:
            off = 0
            for lin in self.linears:
                if lin.bias.data_ptr() != self._b.data_ptr() + off * esz:
                    return False
                off += lin.bias.shape[0]
        return True

    @torch.no_grad()
    def refuse(self) -> None:
        self._w = torch.cat([lin.weight.data for lin in self.linears], dim=0).contiguous()
        off = 0
        for lin in self.linears:
            n = lin.weight.shape[0]
            lin.weight.data = self._w[off:off + n]
            off += n
        if self.has_bias:
            self._b = torch.cat([lin.bias.data for lin in self.linears], dim=0).contiguous()
            off = 0
            for lin in self.linears:
                n = lin.bias.shape[0]
                lin.bias.data = self._b[off:off + n]
                off += n

    def weight(self) -> torch.Tensor:
        if not self._coherent():
            self.refuse()
        return self._w

    def bias(self) -> Optional[torch.Tensor]:
        if not self.has_bias:
            return None
        self.weight()
        return self._b

    def linear(self, x: torch.Tensor) -> torch.Tensor:
        wf = self.weight()
        members = [lin.weight for lin in self.linears]
        if self.has_bias:
            members += [lin.bias for lin in self.linears]
        return ops.fused_linear(x, wf, self.bias(), members)


class PaddedRows:
    """An nn.Linear whose out_features is not a multiple of 8 (BERT's MLM decoder: 30522 x 768, tied to the word
    embeddings; models/bert/modeling_bert.py:490) kept in row-PADDED storage: weight.data / bias.data become views of the
    first `n` rows of zero-padded buffers [n_pad, in] / [n_pad] (n_pad = n rounded up to 64), so the MFMA GEMM's 16-byte
    row stores and the K-padded backward GEMMs run on whole rows.  Same invariants as FusedWeights: parameter identity,
    shape, names, tying and state_dict are untouched; optimizers / load_state_dict / DDP broadcast write through."""

    ALIGN = 64

    def __init__(self, linear: nn.Linear):
        self.linear = linear
        self._w: Optional[torch.Tensor] = None
        self._b: Optional[torch.Tensor] = None

    def _coherent(self) -> bool:
        w, b = self.linear.weight, self.linear.bias
        if self._w is None or self._w.device != w.device or self._w.dtype != w.dtype:
            return False
        if w.data_ptr() != self._w.data_ptr() or not w.is_contiguous():
            return False
        return b is None or (self._b is not None and b.data_ptr() == self._b.data_ptr())

    @torch.no_grad()
    def repad(self) -> None:
        w, b = self.linear.weight, self.linear.bias
        n, k = w.shape
        n_pad = -(-n // self.ALIGN) * self.ALIGN
        self._w = torch.zeros(n_pad, k, dtype=w.dtype, device=w.device)
        self._w[:n].copy_(w.data)
        w.data = self._w[:n]
        if b is not None:
            self._b = torch.zeros(n_pad, dtype=b.dtype, device=b.device)
            self._b[:n].copy_(b.data)
            b.data = self._b[:n]

    def buffers(self):
        """-> (w_pad [n_pad, in], b_pad [n_pad] or None), re-established if the parameters were moved or re-assigned."""
        if not self._coherent():
            self.repad()
        return self._w, (self._b if self.linear.bias is not None else None)
