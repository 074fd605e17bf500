"""Fused AdamW on the HIP kernels.

`TamdAdamW` is a drop-in for `torch.optim.AdamW` (the optimizer `Trainer` builds by default,
src/transformers/trainer.py:1783-1799): same constructor arguments, same `state_dict` layout (`step`, `exp_avg`,
`exp_avg_sq` per parameter), so checkpoints move between the two.  One kernel launch per parameter streams p, g, m, v
once; arithmetic in fp32, every stored tensor rounded once (the semantics of torch's `fused=True`).

    Trainer(model=model, args=args, optimizers=(TamdAdamW(model.parameters(), lr=2e-5, weight_decay=0.01), None))
"""
from __future__ import annotations

import torch

from . import ops


class TamdAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, fp32_moments=False,
                 maximize=False):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"invalid AdamW hyper-parameters: lr={lr} betas={betas} eps={eps} wd={weight_decay}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fp32_moments=fp32_moments,
                        maximize=maximize)
        super().__init__(params, defaults)

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict casts every floating-point state tensor to its parameter's dtype: with
        `fp32_moments=True` and bf16 parameters a resumed run would silently continue on bf16 moments.  Re-install the
        checkpoint's moments in the dtype this optimizer is configured for."""
        super().load_state_dict(state_dict)
        saved = state_dict["state"]
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        params = [p for g in self.param_groups for p in g["params"]]
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        for i, p in zip(ids, params):
            if i not in saved or p not in self.state:
                continue
            want = torch.float32 if group_of[id(p)]["fp32_moments"] else p.dtype
            for key in ("exp_avg", "exp_avg_sq"):
                src = saved[i].get(key)
                if torch.is_tensor(src) and self.state[p][key].dtype != want:
                    self.state[p][key] = src.detach().to(device=p.device, dtype=want).clone()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                lr = lr.item()
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("TamdAdamW does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    mdt = torch.float32 if group["fp32_moments"] else p.dtype
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)  # same key/type as torch.optim.AdamW
                    st["exp_avg"] = torch.zeros_like(p, dtype=mdt, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=mdt, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():  # (fused-weight views are row slices of a contiguous buffer: contiguous)
                    raise RuntimeError("TamdAdamW needs contiguous parameters")
                torch.ops.tamd.adamw_step_(p, g, st["exp_avg"], st["exp_avg_sq"], float(lr), float(b1), float(b2),
                                           float(group["eps"]), float(group["weight_decay"]), int(st["step"].item()),
                                           -1.0 if group["maximize"] else 1.0)
        return loss
