"""Fused AdamW + global gradient-norm clip on the HIP kernels.

`TamdAdamW` is a drop-in for `torch.optim.AdamW` (the optimizer `Trainer` builds by default,
src/transformers/trainer.py:1783-1799): same constructor arguments, same `state_dict` layout (`step`, `exp_avg`,
`exp_avg_sq` per parameter), so checkpoints move between the two.  The whole parameter set of a (param group, dtype) is
ONE launch (csrc/optim.hip `mt_adamw_kernel`: a device-side table of pointers, one workgroup per 64 Ki-element chunk):
the 291 tensors of Llama-3-8B are one launch, not 291.  p, g, m, v are streamed once; arithmetic in fp32, every stored
tensor rounded once (the semantics of torch's `fused=True`).

`max_grad_norm=` folds `Trainer`'s clipping (training_args.py:856 default 1.0; trainer.py:2538-2548 ->
torch.nn.utils.clip_grad_norm_) into the step: one pass over the gradients for the norm (fp32 partial per chunk, a
one-workgroup finish that leaves `norm` and `coef = min(1, max_norm / (norm + 1e-6))` in device memory -- no host
synchronisation), and the coefficient is applied to the gradient in registers inside the AdamW launch: the clipped gradient
is never written.

    opt = TamdAdamW(model.parameters(), lr=2e-5, weight_decay=0.01, max_grad_norm=1.0)
    Trainer(model=model, args=TrainingArguments(max_grad_norm=0.0, ...), optimizers=(opt, None))

`clip_grad_norm_` below is the stand-alone drop-in for `torch.nn.utils.clip_grad_norm_` (same return value and side
effect); `install_trainer_clip()` routes `accelerate.Accelerator.clip_grad_norm_` -- what an unchanged `Trainer` calls --
through it.
"""
from __future__ import annotations

import functools
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _cabi

MT_CHUNK = 65536  # include/tamd.h TAMD_MT_CHUNK
_DTYPE_CODE = {torch.bfloat16: _cabi.TAMD_BF16, torch.float16: _cabi.TAMD_F16, torch.float32: _cabi.TAMD_F32}


class MtTable:
    """The device-side table of include/tamd.h "multi-tensor step" for one list of tensors of one dtype on one device:
    int64 words [p | g | m | v | numel | first chunk], rebuilt (one small host-to-device copy) only when a pointer moved.
    Under zero-copy DDP (ddp.py) the gradients are views of the reducer's buckets and never move."""

    def __init__(self):
        self._key: Optional[tuple] = None
        self.table: Optional[torch.Tensor] = None
        self.n = 0
        self.chunks = 0
        self._keep: tuple = ()

    def update(self, ps, gs, ms, vs) -> "MtTable":
        n = len(gs)
        cols = [[t.data_ptr() for t in col] if col is not None else [0] * n for col in (ps, gs, ms, vs)]
        numel = [g.numel() for g in gs]
        key = (gs[0].device, tuple(cols[0]), tuple(cols[1]), tuple(cols[2]), tuple(cols[3]), tuple(numel))
        if key != self._key:
            start, c = [], 0
            for k in numel:
                start.append(c)
                c += -(-k // MT_CHUNK)
            words = cols[0] + cols[1] + cols[2] + cols[3] + numel + start + [c]
            self.table = torch.tensor(words, dtype=torch.int64).to(gs[0].device)
            self._key, self.n, self.chunks = key, n, c
        self._keep = (ps, gs, ms, vs)  # the table holds raw pointers: keep their owners alive until the next update
        return self


def _grads_by_device_dtype(grads: Iterable[torch.Tensor]) -> Dict[Tuple[torch.device, torch.dtype], List[torch.Tensor]]:
    out: Dict[Tuple[torch.device, torch.dtype], List[torch.Tensor]] = {}
    for g in grads:
        if g.is_sparse:
            raise RuntimeError("tamd: sparse gradients are not supported")
        if g.dtype not in _DTYPE_CODE:
            raise RuntimeError(f"tamd: gradient dtype {g.dtype} is not supported (bf16 / fp16 / fp32)")
        out.setdefault((g.device, g.dtype), []).append(g)
    return out


class _NormState:
    """Tables and scratch of one caller of `grad_norm`, reused from step to step."""

    def __init__(self):
        self.tables: Dict[Tuple[torch.device, torch.dtype], MtTable] = {}
        self.partials: Dict[torch.device, torch.Tensor] = {}


def grad_norm(grads: List[torch.Tensor], max_norm: float, state: Optional[_NormState] = None):
    """L2 norm over every tensor of `grads` and the clip coefficient, both left in device memory.

    Returns (out, tables): `out` fp32 [2] on the first gradient's device -- out[0] the norm, out[1]
    min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0 or infinite) -- and this call's tables by (device, dtype) for
    a following `mt_scale_`.  `grads` must be contiguous (a non-contiguous gradient cannot be scaled in place through a copy).
    """
    state = state if state is not None else _NormState()
    groups = _grads_by_device_dtype(grads)
    dev0 = grads[0].device
    per_dev: Dict[torch.device, List[Tuple[MtTable, torch.dtype]]] = {}
    for (dev, dt), gs in groups.items():
        for g in gs:
            if not g.is_contiguous():
                raise RuntimeError("tamd: grad_norm needs contiguous gradients")
        tab = state.tables.setdefault((dev, dt), MtTable()).update(None, gs, None, None)
        per_dev.setdefault(dev, []).append((tab, dt))
    pieces = []
    for dev, tabs in per_dev.items():
        total = sum(t.chunks for t, _ in tabs)
        buf = state.partials.get(dev)
        if buf is None or buf.numel() < total:
            buf = state.partials[dev] = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
        off = 0
        for tab, dt in tabs:
            if tab.chunks:
                torch.ops.tamd.mt_sumsq(tab.table, tab.n, tab.chunks, buf[off:off + tab.chunks], _DTYPE_CODE[dt])
            off += tab.chunks
        pieces.append(buf[:total])
    partials = pieces[0] if len(pieces) == 1 else torch.cat([p.to(dev0) for p in pieces])
    out = torch.empty(2, dtype=torch.float32, device=dev0)
    mn = float(max_norm)
    torch.ops.tamd.mt_norm_finish(partials, out, mn if mn == mn and mn != float("inf") else 0.0)
    for key in [k for k in state.tables if k not in groups]:  # a dtype without gradients this step: drop its stale table
        del state.tables[key]
    return out, {k: state.tables[k] for k in groups}


_clip_state = _NormState()


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm: float, norm_type: float = 2.0, error_if_nonfinite: bool = False,
                    foreach=None) -> torch.Tensor:
    """Drop-in for `torch.nn.utils.clip_grad_norm_` (what `Trainer._clip_grad_norm` reaches through accelerate,
    src/transformers/trainer.py:2538-2548): returns the total L2 norm of the gradients (0-dim tensor, device memory) and
    scales them in place by min(1, max_norm / (norm + 1e-6)).  One read of the gradients for the norm; the scaling pass
    exits at once on the device when nothing has to be clipped.  `max_norm=inf` only measures (`Trainer._get_grad_norm`).
    The norm is accumulated in fp32 over the stored gradients (torch rounds every per-tensor norm to the gradient dtype
    first: its bf16 result carries 3 significant digits)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    if float(norm_type) != 2.0:
        raise ValueError("tamd: clip_grad_norm_ implements the L2 norm (norm_type=2), what Trainer uses")
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    out, tables = grad_norm(grads, max_norm, _clip_state)
    if error_if_nonfinite and not bool(torch.isfinite(out[0])):
        raise RuntimeError("The total norm for gradients from `parameters` is non-finite, so it cannot be clipped.")
    if float(max_norm) != float("inf") and max_norm > 0:
        for (dev, dt), tab in tables.items():
            if tab.chunks:
                coef = out[1:2] if dev == out.device else out[1:2].to(dev)
                torch.ops.tamd.mt_scale_(tab.table, tab.n, tab.chunks, coef, _DTYPE_CODE[dt])
    return out[0]


def install_trainer_clip() -> bool:
    """An unchanged `Trainer` clips through `accelerate.Accelerator.clip_grad_norm_` (trainer.py:2538-2548).  Route its
    plain-PyTorch branch (no FSDP / DeepSpeed / XLA; L2 norm; GPU gradients) through `clip_grad_norm_` above; everything
    else reaches accelerate's own implementation.  Idempotent; returns False when accelerate is not importable."""
    try:
        import accelerate
        from accelerate.utils import DistributedType
    except Exception:
        return False
    cls = accelerate.Accelerator
    orig = cls.clip_grad_norm_
    if getattr(orig, "_tamd_clip", False):
        return True
    plain = {DistributedType.NO, DistributedType.MULTI_GPU}

    @functools.wraps(orig)
    def clip(self, parameters, max_norm, norm_type=2):
        from . import ops

        if self.distributed_type in plain and float(norm_type) == 2.0:
            parameters = list(parameters)
            grads = [p.grad for p in parameters if p.grad is not None]
            on_kernels = grads and all((g.is_cuda or ops.backend_is_emulated()) and g.dtype in _DTYPE_CODE and
                                       g.is_contiguous() and not g.is_sparse for g in grads)
            if on_kernels:
                self.unscale_gradients()
                return clip_grad_norm_(parameters, max_norm)
        return orig(self, parameters, max_norm, norm_type=norm_type)

    clip._tamd_clip = True
    clip._tamd_orig = orig
    cls.clip_grad_norm_ = clip
    return True


class TamdAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, fp32_moments=False,
                 maximize=False, max_grad_norm: Optional[float] = None):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"invalid AdamW hyper-parameters: lr={lr} betas={betas} eps={eps} wd={weight_decay}")
        if max_grad_norm is not None and max_grad_norm < 0.0:
            raise ValueError(f"invalid max_grad_norm={max_grad_norm}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fp32_moments=fp32_moments,
                        maximize=maximize)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm  # None / 0: no clipping.  One norm over ALL param groups, as Trainer's
        self.grad_norm: Optional[torch.Tensor] = None  # the pre-clip norm of the last step (device memory, 0-dim)
        self._tables: Dict[tuple, MtTable] = {}
        self._norm_state = _NormState()

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
        # launches: every parameter of a (group, device, dtype, moment dtype, step count) in one table
        launches: Dict[tuple, Tuple[list, list, list, list]] = {}
        all_grads: List[torch.Tensor] = []
        for gi, group in enumerate(self.param_groups):
            fresh, steps, entries = [], [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("TamdAdamW does not support sparse gradients")
                if p.dtype not in _DTYPE_CODE:
                    raise RuntimeError(f"TamdAdamW: parameter dtype {p.dtype} is not supported (bf16 / fp16 / fp32)")
                if not p.is_contiguous():  # (fused-weight views are row slices of a contiguous buffer: contiguous)
                    raise RuntimeError("TamdAdamW needs contiguous parameters")
                st = self.state[p]
                if len(st) == 0:
                    mdt = torch.float32 if group["fp32_moments"] else p.dtype
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)  # same key/type as torch.optim.AdamW
                    st["exp_avg"] = torch.zeros_like(p, dtype=mdt, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=mdt, memory_format=torch.contiguous_format)
                    fresh.append(st)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                steps.append(st["step"])
                entries.append((p, g, st))
            if not entries:
                continue
            cpu_steps = [s for s in steps if not s.is_cuda]
            if len(cpu_steps) == len(steps):
                torch._foreach_add_(cpu_steps, 1.0)
                counts = torch.stack(cpu_steps).tolist()  # one read for the group (host memory: no synchronisation)
            else:  # a checkpoint of torch's fused / capturable AdamW keeps `step` on the GPU
                for s in steps:
                    s += 1
                counts = [float(s) for s in steps]
            for (p, g, st), t in zip(entries, counts):
                key = (gi, p.device, p.dtype, st["exp_avg"].dtype, int(t))
                cols = launches.setdefault(key, ([], [], [], []))
                cols[0].append(p)
                cols[1].append(g)
                cols[2].append(st["exp_avg"])
                cols[3].append(st["exp_avg_sq"])
                all_grads.append(g)
        if not launches:
            return loss
        coef_by_dev: Dict[torch.device, torch.Tensor] = {}
        if self.max_grad_norm:
            out, _ = grad_norm(all_grads, self.max_grad_norm, self._norm_state)
            self.grad_norm = out[0]
            coef_by_dev[out.device] = out[1:2]
        live = set()
        for key, (ps, gs, ms, vs) in launches.items():
            gi, dev, dt, mdt, t = key
            group = self.param_groups[gi]
            lr = group["lr"]
            if isinstance(lr, torch.Tensor):
                lr = lr.item()
            b1, b2 = group["betas"]
            tkey = key[:4]
            if tkey in live:  # two step counts inside one group (parameters added later): a second table for the stragglers
                tkey = key
            live.add(tkey)
            tab = self._tables.setdefault(tkey, MtTable()).update(ps, gs, ms, vs)
            coef = None
            if coef_by_dev:
                coef = coef_by_dev.get(dev)
                if coef is None:
                    coef = coef_by_dev[dev] = next(iter(coef_by_dev.values())).to(dev)
            torch.ops.tamd.mt_adamw_step_(tab.table, tab.n, tab.chunks, float(lr), float(b1), float(b2),
                                          float(group["eps"]), float(group["weight_decay"]), t,
                                          -1.0 if group["maximize"] else 1.0, coef, _DTYPE_CODE[dt], _DTYPE_CODE[mdt])
        for k in [k for k in self._tables if k not in live]:
            del self._tables[k]
        return loss
