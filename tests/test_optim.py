"""Fused AdamW against torch.optim.AdamW -- the optimizer `Trainer` builds by default
(src/transformers/trainer.py:1783-1799) -- and against the oracle's restatement with explicit storage roundings."""
import copy
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import transformers_amd
from transformers_amd import ops

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
import oracle as orc  # noqa: E402

HYP = dict(lr=2e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)


def _bf16_ulp_close(a, b, frac=0.01, atol=0.0):
    """a, b bf16 tensors: equal except for at most `frac` of the elements, and those by one bf16 ulp (+ `atol`: near a
    zero crossing of p the ulp of p says nothing about the rounding of the update that was subtracted)."""
    a, b = a.float().cpu(), b.float().cpu()
    diff = (a - b).abs()
    ulp = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-30 + atol
    assert (diff <= ulp).all(), (diff / ulp).max().item()
    assert (diff > 0).float().mean().item() <= frac


def test_adamw_fp32_matches_torch(env):
    torch.manual_seed(0)
    shapes = [(256, 64), (1000,), (8, 12, 4)]
    ref_p = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone().to(env.device)) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, foreach=False, **HYP)
    our = transformers_amd.TamdAdamW(our_p, **HYP)
    for _ in range(4):
        for rp, op in zip(ref_p, our_p):
            g = torch.randn_like(rp)
            rp.grad, op.grad = g, g.clone().to(env.device)
        ref.step()
        our.step()
    for rp, op in zip(ref_p, our_p):
        assert torch.allclose(op.detach().cpu(), rp.detach(), rtol=3e-6, atol=1e-7)
        assert torch.allclose(our.state[op]["exp_avg_sq"].cpu(), ref.state[rp]["exp_avg_sq"], rtol=3e-6, atol=1e-12)
    # same state_dict layout: a checkpoint written by one loads into the other
    sd = our.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ref2 = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone().cpu()) for p in our_p], foreach=False, **HYP)
    ref2.load_state_dict(sd)
    assert float(ref2.state_dict()["state"][0]["step"]) == 4.0


@pytest.mark.parametrize("fp32_moments", [False, True])
def test_adamw_bf16_matches_oracle_roundings(env, fp32_moments):
    torch.manual_seed(1)
    n = 4096 if env.big else 512
    p0 = torch.randn(n).bfloat16()
    w = torch.nn.Parameter(p0.clone().to(env.device))
    opt = transformers_amd.TamdAdamW([w], fp32_moments=fp32_moments, **HYP)
    p, m, v = p0.float().numpy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    rs = (lambda x: x) if fp32_moments else orc.rnd_bf16
    for t in range(3):
        g = (torch.randn(n) * 0.3).bfloat16()
        w.grad = g.clone().to(env.device)
        opt.step()
        p, m, v = orc.adamw_step(p, g.float().numpy(), m, v, HYP["lr"], *HYP["betas"], HYP["eps"],
                                 HYP["weight_decay"], t + 1, round_p=orc.rnd_bf16, round_state=rs)
    st = opt.state[w]
    assert st["exp_avg"].dtype == (torch.float32 if fp32_moments else torch.bfloat16)
    _bf16_ulp_close(w.detach(), torch.from_numpy(p).bfloat16())
    if fp32_moments:
        assert torch.allclose(st["exp_avg"].cpu(), torch.from_numpy(m), rtol=2e-6, atol=1e-9)
        assert torch.allclose(st["exp_avg_sq"].cpu(), torch.from_numpy(v), rtol=2e-6, atol=1e-12)
    else:
        _bf16_ulp_close(st["exp_avg"], torch.from_numpy(m).bfloat16())
        _bf16_ulp_close(st["exp_avg_sq"], torch.from_numpy(v).bfloat16())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adamw_small_and_ragged_parameters(env, dtype):
    """Parameters whose element count is not a multiple of the 16-byte vector (CLIP's scalar `logit_scale`, a 2- or
    3-label classifier bias, odd hidden sizes): the vector body plus a one-element-per-thread tail, same update."""
    torch.manual_seed(3)
    shapes = [(), (3,), (6,), (37,), (5, 13)]
    ref_p = [torch.nn.Parameter(torch.randn(s).to(dtype).float()) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone().to(dtype).to(env.device)) for p in ref_p]
    ref = torch.optim.AdamW(ref_p, foreach=False, **HYP)
    our = transformers_amd.TamdAdamW(our_p, fp32_moments=True, **HYP)
    for _ in range(3):
        for rp, op in zip(ref_p, our_p):
            g = torch.randn(rp.shape).to(dtype)
            rp.grad, op.grad = g.float(), g.clone().to(env.device)
        ref.step()
        our.step()
        if dtype != torch.float32:  # the reference run keeps fp32 parameters: round them where ours are stored
            with torch.no_grad():
                for rp in ref_p:
                    rp.copy_(rp.to(dtype).float())
    for rp, op in zip(ref_p, our_p):
        if dtype == torch.float32:
            assert torch.allclose(op.detach().cpu(), rp.detach(), rtol=3e-6, atol=1e-7)
        else:
            _bf16_ulp_close(op.detach(), rp.detach().to(dtype), frac=1.0)
        assert torch.allclose(our.state[op]["exp_avg"].cpu(), ref.state[rp]["exp_avg"], rtol=1e-5, atol=1e-8)


def test_adamw_resume_keeps_fp32_moments(env):
    """Optimizer.load_state_dict casts state to the parameter dtype; fp32 moments must survive a resume."""
    torch.manual_seed(4)
    w = torch.nn.Parameter(torch.randn(64).bfloat16().to(env.device))
    opt = transformers_amd.TamdAdamW([w], fp32_moments=True, **HYP)
    w.grad = torch.randn(64).bfloat16().to(env.device)
    opt.step()
    sd = copy.deepcopy(opt.state_dict())  # (a checkpoint: load_state_dict shares same-dtype tensors such as `step`)
    w2 = torch.nn.Parameter(w.detach().clone())
    opt2 = transformers_amd.TamdAdamW([w2], fp32_moments=True, **HYP)
    opt2.load_state_dict(sd)
    st = opt2.state[w2]
    assert st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].dtype == torch.float32
    assert torch.equal(st["exp_avg"], opt.state[w]["exp_avg"]) and torch.equal(st["exp_avg_sq"], opt.state[w]["exp_avg_sq"])
    g = torch.randn(64).bfloat16().to(env.device)
    w.grad, w2.grad = g, g.clone()
    opt.step()
    opt2.step()
    assert torch.equal(w.detach(), w2.detach())
    # bf16 moments stay bf16
    opt3 = transformers_amd.TamdAdamW([torch.nn.Parameter(w.detach().clone())], **HYP)
    p3 = opt3.param_groups[0]["params"][0]
    p3.grad = g.clone()
    opt3.step()
    opt4 = transformers_amd.TamdAdamW([torch.nn.Parameter(w.detach().clone())], **HYP)
    opt4.load_state_dict(copy.deepcopy(opt3.state_dict()))
    assert opt4.state[opt4.param_groups[0]["params"][0]]["exp_avg"].dtype == torch.bfloat16


@pytest.mark.gpu
def test_adamw_bf16_vs_torch_fused_on_gpu():
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    p0 = torch.randn(1 << 20).bfloat16()
    a = torch.nn.Parameter(p0.clone().to(dev))
    b = torch.nn.Parameter(p0.clone().to(dev))
    ours = transformers_amd.TamdAdamW([a], **HYP)
    ref = torch.optim.AdamW([b], fused=True, **HYP)
    for _ in range(3):
        g = (torch.randn(1 << 20) * 0.3).bfloat16().to(dev)
        a.grad, b.grad = g.clone(), g.clone()
        ours.step()
        ref.step()
    # torch's fused kernel applies the decay as p - lr*wd*p and keeps its own operation order: differences are one
    # rounding of the update term (lr * O(1) * 2^-8 ~ 1e-5) or one bf16 ulp of p
    _bf16_ulp_close(a.detach(), b.detach(), frac=0.05, atol=2e-5)


# ---------------------------------------------------------------------------------- multi-tensor step + gradient clipping
def _mixed_params(device, dtype, big):
    """Tensors that exercise every path of the multi-tensor kernels: one spanning several 64 Ki-element chunks with a ragged
    end, a scalar, a ragged vector, and a view whose storage is not 16-byte aligned (one element into a buffer)."""
    torch.manual_seed(7)
    sizes = [(2 * 65536 + 37,) if big else (65536 + 37,), (), (37,), (8, 12, 4)]
    ps = [torch.randn(s).to(dtype) for s in sizes]
    buf = torch.randn(1 + 1000).to(dtype).to(device)
    out = [torch.nn.Parameter(p.clone().to(device)) for p in ps]
    out.append(torch.nn.Parameter(buf[1:]))  # contiguous, misaligned
    ps.append(buf[1:].detach().cpu().clone())
    return ps, out


def test_clip_grad_norm_matches_torch(env):
    from transformers_amd import optim

    ref_t, ours = _mixed_params(env.device, torch.float32, env.big)
    ref = [torch.nn.Parameter(t.clone()) for t in ref_t]
    torch.manual_seed(8)
    for max_norm, scale in ((1.0, 0.5), (1.0, 1e-5), (float("inf"), 0.5)):
        for rp, op in zip(ref, ours):
            g = torch.randn(rp.shape) * scale
            rp.grad, op.grad = g.clone(), g.clone().to(env.device)
        want = torch.nn.utils.clip_grad_norm_(ref, max_norm, foreach=False)
        got = optim.clip_grad_norm_(ours, max_norm)
        assert got.dim() == 0 and abs(float(got) - float(want)) <= 3e-6 * float(want)
        for rp, op in zip(ref, ours):
            assert torch.allclose(op.grad.cpu(), rp.grad, rtol=3e-6, atol=0), (max_norm, scale)
        if scale == 1e-5:  # nothing to clip: the gradients are untouched, bit for bit
            assert float(want) < 1.0
    with pytest.raises(ValueError):
        optim.clip_grad_norm_(ours, 1.0, norm_type=1.0)


@pytest.mark.parametrize("dtype,fp32_moments", [(torch.float32, False), (torch.bfloat16, True), (torch.bfloat16, False)])
def test_multi_tensor_step_equals_per_tensor_kernel(env, dtype, fp32_moments):
    """One launch over the table == one tamd_adamw_step per tensor (the round-2 kernel, golden-tested), bit for bit --
    including the in-register clip coefficient (grad_scale of the per-tensor kernel)."""
    _, ps = _mixed_params(env.device, dtype, env.big)
    twins = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    twins[-1] = torch.nn.Parameter(torch.cat([ps[-1].new_zeros(1), ps[-1].detach()])[1:])  # same misalignment
    opt = transformers_amd.TamdAdamW(ps, fp32_moments=fp32_moments, max_grad_norm=0.7, **HYP)
    mdt = torch.float32 if fp32_moments else dtype
    ms = [torch.zeros_like(p, dtype=mdt) for p in twins]
    vs = [torch.zeros_like(p, dtype=mdt) for p in twins]
    torch.manual_seed(9)
    for t in range(2):
        gs = [(torch.randn(p.shape) * 0.3).to(dtype).to(env.device) for p in ps]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()
        norm, coef = orc.clip_grad_norm([g.float().cpu().numpy() for g in gs], 0.7)
        assert abs(float(opt.grad_norm) - float(norm)) <= 2e-6 * float(norm)
        dev_coef = float(coef)
        for p, g, m, v in zip(twins, gs, ms, vs):
            ops.raw_adamw_step_(p.data.view(-1), g.view(-1), m.view(-1), v.view(-1), lr=HYP["lr"], beta1=HYP["betas"][0],
                                beta2=HYP["betas"][1], eps=HYP["eps"], weight_decay=HYP["weight_decay"], step=t + 1,
                                grad_scale=dev_coef)
    assert len(opt._tables) == 1
    for p, q, m, v in zip(ps, twins, ms, vs):
        if dtype == torch.float32:  # (the coefficient went through the host as a double on the twin side)
            assert torch.allclose(p.detach().cpu(), q.detach().cpu(), rtol=2e-6, atol=1e-8)
            assert torch.allclose(opt.state[p]["exp_avg"].cpu(), m.cpu(), rtol=2e-6, atol=1e-8)
        else:
            _bf16_ulp_close(p.detach(), q.detach(), frac=0.01, atol=1e-6)
            assert opt.state[p]["exp_avg"].dtype == mdt


def test_multi_tensor_step_without_clip_is_bit_identical(env):
    """grad_scale 1: the table launch and the per-tensor kernel run the same arithmetic on the same bits."""
    _, ps = _mixed_params(env.device, torch.bfloat16, False)
    twins = [p.detach().clone() for p in ps]
    twins[-1] = torch.cat([ps[-1].new_zeros(1), ps[-1].detach()])[1:]
    opt = transformers_amd.TamdAdamW(ps, **HYP)
    ms = [torch.zeros_like(p) for p in twins]
    vs = [torch.zeros_like(p) for p in twins]
    torch.manual_seed(10)
    for t in range(2):
        for p, q, m, v in zip(ps, twins, ms, vs):
            g = (torch.randn(p.shape) * 0.3).bfloat16().to(env.device)
            p.grad = g.clone()
            ops.raw_adamw_step_(q.view(-1), g.view(-1), m.view(-1), v.view(-1), lr=HYP["lr"], beta1=HYP["betas"][0],
                                beta2=HYP["betas"][1], eps=HYP["eps"], weight_decay=HYP["weight_decay"], step=t + 1)
        opt.step()
    for p, q, m, v in zip(ps, twins, ms, vs):
        assert torch.equal(p.detach(), q)
        assert torch.equal(opt.state[p]["exp_avg"], m) and torch.equal(opt.state[p]["exp_avg_sq"], v)


def test_param_groups_and_dtypes_launch_counts(env):
    """Trainer builds two groups (decay / no decay, trainer.py:1747-1760); a model may mix bf16 weights with fp32 norms:
    one table per (group, dtype), ONE norm over all of them."""
    torch.manual_seed(11)
    a = [torch.nn.Parameter(torch.randn(300).bfloat16().to(env.device)) for _ in range(3)]
    b = [torch.nn.Parameter(torch.randn(50).to(env.device)) for _ in range(2)]
    c = [torch.nn.Parameter(torch.randn(20).bfloat16().to(env.device))]
    opt = transformers_amd.TamdAdamW([{"params": a + b, "weight_decay": 0.1}, {"params": c, "weight_decay": 0.0}],
                                     lr=1e-2, max_grad_norm=1.0)
    ref_p = [torch.nn.Parameter(p.detach().float().cpu()) for p in a + b + c]
    ref = torch.optim.AdamW([{"params": ref_p[:5], "weight_decay": 0.1}, {"params": ref_p[5:], "weight_decay": 0.0}],
                            lr=1e-2, foreach=False)
    for _ in range(2):
        for p, rp in zip(a + b + c, ref_p):
            g = torch.randn(p.shape).to(p.dtype)
            p.grad, rp.grad = g.clone().to(env.device), g.float()
        want = torch.nn.utils.clip_grad_norm_(ref_p, 1.0, foreach=False)
        ref.step()
        opt.step()
        assert abs(float(opt.grad_norm) - float(want)) <= 3e-6 * float(want)
        with torch.no_grad():
            for p, rp in zip(a + b + c, ref_p):
                rp.copy_(rp.to(p.dtype).float())
    assert len(opt._tables) == 3
    for p, rp in zip(a + b + c, ref_p):
        if p.dtype == torch.float32:
            assert torch.allclose(p.detach().cpu(), rp.detach(), rtol=1e-5, atol=1e-7)
        else:
            _bf16_ulp_close(p.detach(), rp.detach().bfloat16(), frac=1.0, atol=1e-4)
    # a parameter without a gradient is skipped; the table follows
    a[0].grad = None
    for p in a[1:] + b + c:
        p.grad = torch.ones_like(p)
    before = a[0].detach().clone()
    opt.step()
    assert torch.equal(a[0].detach(), before)


def test_accelerator_clip_routes_through_the_kernels(env):
    """`Trainer._clip_grad_norm` calls accelerate's `clip_grad_norm_` (trainer.py:2538-2548): after
    `install_trainer_clip()` its plain-PyTorch branch runs the kernels, and returns / does what torch's does."""
    accelerate = pytest.importorskip("accelerate")
    from transformers_amd import optim

    assert optim.install_trainer_clip() and optim.install_trainer_clip()  # idempotent
    acc = accelerate.Accelerator(cpu=env.name != "hip")
    torch.manual_seed(12)
    ps = [torch.nn.Parameter(torch.randn(100, 7).to(env.device)) for _ in range(3)]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    for p, rp in zip(ps, ref):
        g = torch.randn(100, 7)
        p.grad, rp.grad = g.clone().to(env.device), g.clone()
    calls = []
    real = optim.clip_grad_norm_
    optim.clip_grad_norm_ = lambda *a, **k: calls.append(1) or real(*a, **k)
    try:
        got = acc.clip_grad_norm_(iter(ps), 0.5)
    finally:
        optim.clip_grad_norm_ = real
    want = torch.nn.utils.clip_grad_norm_(ref, 0.5, foreach=False)
    assert calls == [1] and abs(float(got) - float(want)) <= 3e-6 * float(want)
    for p, rp in zip(ps, ref):
        assert torch.allclose(p.grad.cpu(), rp.grad, rtol=3e-6, atol=0)
