"""The product path goes through `torch.ops.tamd.*` (boundary B3: torch.ops over the C-ABI library).

* every op of the namespace has a schema, a fake (Meta) implementation and -- for the differentiable ones -- an
  autograd formula registered with torch.library (precedent: src/transformers/integrations/moe.py:245-257):
  `torch.library.opcheck` verifies schema / fake / autograd registration on the kernels themselves;
* gradients of the differentiable ops agree with autograd through an fp32 restatement of the reference formulas
  (gradcheck-style, at bf16 tolerances);
* the model code under transformers_amd/models/ contains no direct kernel-launcher (`raw_*`) calls.
"""
import re
from pathlib import Path

import pytest
import torch

from conftest import rel_err
from transformers_amd import layer_ops, ops  # noqa: F401

T = torch.ops.tamd
ROOT = Path(__file__).resolve().parent.parent
CHECKS = ("test_schema", "test_autograd_registration", "test_faketensor")


def _bf(*shape, dev, scale=1.0, grad=False):
    t = (torch.randn(*shape) * scale).bfloat16().to(dev)
    return t.requires_grad_(True) if grad else t


def test_models_call_kernels_only_through_torch_ops():
    for f in sorted((ROOT / "transformers_amd" / "models").glob("*.py")) + [ROOT / "transformers_amd" / "attention.py",
                                                                           ROOT / "transformers_amd" / "patch.py",
                                                                           ROOT / "transformers_amd" / "fused_params.py"]:
        src = f.read_text()
        assert not re.search(r"\braw_[a-z_]+\(", src), f"{f.name} launches kernels directly"
        assert "autograd.Function" not in src, f"{f.name} defines its own autograd node instead of a torch.ops op"
        assert "ctypes" not in src and "_cabi.TamdLib" not in src, f.name


def test_namespace_is_complete():
    """One dispatcher op per C-ABI compute entry point, plus the differentiable ops the models use."""
    kernel_level = ["rmsnorm_fwd", "rmsnorm_bwd", "layernorm_fwd", "layernorm_bwd", "rope_", "embedding_fwd",
                    "embedding_bwd", "bert_embeddings_fwd", "swiglu_fwd", "swiglu_bwd", "bias_act_fwd", "bias_act_bwd",
                    "add", "colsum", "transpose", "cross_entropy_fwd", "cross_entropy_bwd", "adamw_step_", "gemm",
                    "gemm_out", "attn_fwd", "attn_bwd"]
    differentiable = ["rmsnorm", "add_rmsnorm", "layernorm", "add_layernorm", "linear", "fused_linear", "conv1d", "rope",
                      "attention", "swiglu", "bias_act", "embedding", "bert_embeddings", "cross_entropy_sum",
                      "linear_cross_entropy", "llama_layer", "bert_layer", "padded_vocab_head"]
    for name in kernel_level + differentiable:
        op = getattr(T, name).default
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "CUDA"), name
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "Meta"), name
    for name in differentiable:
        assert torch._C._dispatch_has_kernel_for_dispatch_key(getattr(T, name).default.name(), "Autograd"), name


def test_fake_implementations_give_shapes_without_data():
    """Meta tensors flow through the ops (what torch's tracing / shape tooling uses): no kernel runs."""
    m = torch.device("meta")
    x = torch.empty(4, 64, 256, dtype=torch.bfloat16, device=m)
    w = torch.empty(512, 256, dtype=torch.bfloat16, device=m)
    y, pre = T.linear(x, w, None, None, 0, False)
    assert y.shape == (4, 64, 512) and pre.numel() == 0
    q = torch.empty(2, 128, 8, 64, dtype=torch.bfloat16, device=m)
    kv = torch.empty(2, 128, 2, 64, dtype=torch.bfloat16, device=m)
    o, lse = T.attention(q, kv, kv, None, 0.125, True, 0.0, 0, None, True)
    assert o.shape == q.shape and lse.shape == (2, 8, 128) and lse.dtype == torch.float32
    assert T.gemm(torch.empty(64, 128, dtype=torch.bfloat16, device=m), w[:, :64], True, False).shape == (128, 512)
    assert T.swiglu(torch.empty(3, 5, 1024, dtype=torch.bfloat16, device=m)).shape == (3, 5, 512)


def test_opcheck_differentiable_ops(env):
    """torch.library.opcheck on the kernels: the schema is truthful (no hidden mutation / aliasing), the fake
    implementation matches the real output metadata, the autograd formula is registered the supported way."""
    from torch.library import opcheck

    dev = env.device
    torch.manual_seed(0)
    x, w = _bf(2, 24, 128, dev=dev, grad=True), _bf(128, dev=dev, grad=True)
    opcheck(T.rmsnorm, (x, w, 1e-5), test_utils=CHECKS)
    opcheck(T.add_rmsnorm, (x, _bf(2, 24, 128, dev=dev), w, 1e-5), test_utils=CHECKS)
    opcheck(T.layernorm, (x, w, _bf(128, dev=dev, grad=True), 1e-5), test_utils=CHECKS)
    wl, bl = _bf(192, 128, dev=dev, scale=0.05, grad=True), _bf(192, dev=dev, grad=True)
    opcheck(T.linear, (x, wl, bl, None, ops.ACT_GELU_ERF, True), test_utils=CHECKS)
    opcheck(T.linear, (x, wl, None, None, ops.ACT_NONE, True), test_utils=CHECKS)
    opcheck(T.gemm, (x.detach().view(-1, 128), wl.detach()), test_utils=("test_schema", "test_faketensor"))
    opcheck(T.swiglu, (_bf(3, 16, 256, dev=dev, grad=True),), test_utils=CHECKS)
    qkv = _bf(2, 40, 6 * 64, dev=dev, grad=True)
    q, k, v = (qkv[..., :256].view(2, 40, 4, 64), qkv[..., 256:320].view(2, 40, 1, 64),
               qkv[..., 320:].view(2, 40, 1, 64))
    opcheck(T.attention, (q, k, v, None, 0.125, True, 0.0, 0, None, True), test_utils=CHECKS)
    cos, sin = _bf(1, 40, 64, dev=dev), _bf(1, 40, 64, dev=dev)
    opcheck(T.rope, (qkv, cos, sin, 5, 64), test_utils=CHECKS)
    ids = torch.randint(0, 50, (3, 9)).to(dev)
    opcheck(T.embedding, (ids, _bf(50, 64, dev=dev, grad=True), -1), test_utils=CHECKS)
    logits = _bf(12, 200, dev=dev, grad=True)
    opcheck(T.cross_entropy_sum, (logits, torch.randint(0, 200, (12,)).to(dev), -100), test_utils=CHECKS)
    # the padded vocabulary head (an output that is a row-strided view of an internal [M, Vp] buffer)
    wp = torch.zeros(256, 128, dtype=torch.bfloat16, device=dev)
    wp[:202] = _bf(202, 128, dev=dev, scale=0.05)
    bp = torch.zeros(256, dtype=torch.bfloat16, device=dev)
    wv, bv = wp[:202].detach().requires_grad_(True), bp[:202].detach().requires_grad_(True)
    hh = _bf(24, 128, dev=dev, grad=True)
    opcheck(T.padded_vocab_head, (hh, wp, bp, wv, bv, torch.randint(0, 202, (24,)).to(dev), -100, True), test_utils=CHECKS)
    opcheck(T.padded_vocab_head, (hh, wp, bp, wv, bv, None, -100, True), test_utils=CHECKS)
    # in-place kernel-level op: the schema declares the mutation
    opcheck(T.rope_, (qkv.detach().clone().view(80, 384), cos, sin, 40, 5, 64), test_utils=("test_schema",))


def test_torch_ops_gradients_match_fp32_autograd(env):
    """Call the dispatcher ops directly (not the Python wrappers) and compare outputs and gradients with torch
    autograd through the fp32 formulas of the reference (modeling_llama.py:62-67, :174-176, :191-213)."""
    dev = env.device
    torch.manual_seed(1)
    t, h, n = (512, 1024, 2048) if env.big else (40, 128, 192)
    x, w, b = _bf(t, h, dev=dev, grad=True), _bf(n, h, dev=dev, scale=h ** -0.5, grad=True), _bf(n, dev=dev, grad=True)
    res = _bf(t, n, dev=dev, grad=True)
    y, _ = T.linear(x, w, b, res, 0, True)
    xr, wr, br, rr = (a.detach().float().requires_grad_(True) for a in (x, w, b, res))
    yr = torch.nn.functional.linear(xr, wr, br) + rr
    assert rel_err(y, yr) < 4e-3
    g = torch.randn_like(yr)
    y.backward(g.to(y.dtype))
    yr.backward(g.to(y.dtype).float())
    for a, r in ((x, xr), (w, wr), (b, br), (res, rr)):
        assert rel_err(a.grad, r.grad) < 0.0034
    # gemm with the k-major operand modes is its own transpose
    a2, b2 = _bf(t, n, dev=dev), _bf(t, h, dev=dev)
    dw = T.gemm(a2, b2, True, True)                                      # [n, h] = a2^T . b2
    assert rel_err(dw, a2.float().t() @ b2.float()) < 0.0034
    # attention: GQA, causal, fp32 softmax
    bsz, s, hq, hkv, d = (2, 512, 8, 2, 128) if env.big else (2, 48, 4, 2, 64)
    q, k, v = _bf(bsz, s, hq, d, dev=dev, grad=True), _bf(bsz, s, hkv, d, dev=dev, grad=True), _bf(bsz, s, hkv, d,
                                                                                                     dev=dev, grad=True)
    o, lse = T.attention(q, k, v, None, d ** -0.5, True, 0.0, 0, None, True)
    qf, kf, vf = (a.detach().float().requires_grad_(True) for a in (q, k, v))
    kk, vv = (a.repeat_interleave(hq // hkv, dim=2) for a in (kf, vf))
    sc = torch.einsum("bqhd,bkhd->bhqk", qf, kk) * d ** -0.5
    sc = sc.masked_fill(torch.ones(s, s, dtype=torch.bool, device=sc.device).triu(1), float("-inf"))
    of = torch.einsum("bhqk,bkhd->bqhd", sc.softmax(-1), vv)
    assert rel_err(o, of) < 4e-3
    assert rel_err(lse, sc.logsumexp(-1)) < 4e-4   # (q enters the kernel multiplied by scale*log2(e) and re-rounded)
    go = torch.randn_like(of)
    o.backward(go.to(o.dtype))
    of.backward(go.to(o.dtype).float())
    for a, r in ((q, qf), (k, kf), (v, vf)):
        assert rel_err(a.grad, r.grad) < 0.0054
    # rmsnorm + swiglu chain through the ops
    gu = _bf(t, 2 * h, dev=dev, grad=True)
    wn = (torch.rand(h) + 0.5).bfloat16().to(dev).requires_grad_(True)
    out = T.rmsnorm(T.swiglu(gu), wn, 1e-5)[0]
    guf, wnf = gu.detach().float().requires_grad_(True), wn.detach().float().requires_grad_(True)
    a_ = torch.nn.functional.silu(guf[:, :h]) * guf[:, h:]
    outf = wnf * (a_ * torch.rsqrt(a_.pow(2).mean(-1, keepdim=True) + 1e-5))
    assert rel_err(out, outf) < 0.0068
    out.backward(torch.ones_like(out))
    outf.backward(torch.ones_like(outf))
    assert rel_err(gu.grad, guf.grad) < 7e-3 and rel_err(wn.grad, wnf.grad) < 7e-3


def test_llama_layer_op_twice_differentiable_graph(env):
    """the fused layer's backward must not overwrite what the forward saved -- a retained graph
    differentiated twice gives the same gradients both times."""
    import copy

    import transformers_amd
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(2)
    cfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=1,
                      num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=64,
                      attn_implementation="eager")
    m = transformers_amd.accelerate(LlamaForCausalLM(cfg).bfloat16().to(env.device)).train()
    ids = torch.randint(0, 128, (2, 24)).to(env.device)
    loss = m(input_ids=ids, labels=ids, use_cache=False).loss
    loss.backward(retain_graph=True)
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    loss.backward()
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, g1[n]), n


def test_llama_layer_saved_swiglu_product_matches_rematerialised(env, monkeypatch):
    """The SiLU*up product kept from the forward (layer_ops._SAVE_ACT, the default: memory for HBM traffic) and the one the
    SwiGLU backward re-materialises are the same bits: every gradient of a two-layer model is identical either way."""
    import transformers_amd
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(3)
    cfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=64,
                      attn_implementation="eager")
    m = transformers_amd.accelerate(LlamaForCausalLM(cfg).bfloat16().to(env.device)).train()
    ids = torch.randint(0, 128, (2, 24)).to(env.device)
    grads = []
    for save in (True, False):
        monkeypatch.setattr(layer_ops, "_SAVE_ACT", save)
        m.zero_grad(set_to_none=True)
        m(input_ids=ids, labels=ids, use_cache=False).loss.backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


def test_swiglu_bwd_form_is_measured_on_the_device(env):
    """Which of the two bit-identical forms of the SiLU*up backward runs (the dX GEMM's way out or GEMM + kernel) is measured
    on the operands of the first backward of each shape (torch_binding.cpp swiglu_bwd_fused; round 2's slow regime of the fused
    form, profiles/r02_regression_note.md): on the GPU the record carries both timings and the faster form; the CPU execution
    model has no clock and takes the fused form unrecorded."""
    import os

    import transformers_amd
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers_amd import _native

    torch.manual_seed(5)
    cfg = LlamaConfig(vocab_size=128, hidden_size=192, intermediate_size=320, num_hidden_layers=1, num_attention_heads=3,
                      num_key_value_heads=1, head_dim=64, max_position_embeddings=64, attn_implementation="eager")
    m = transformers_amd.accelerate(LlamaForCausalLM(cfg).bfloat16().to(env.device)).train()
    ids = torch.randint(0, 128, (2, 40)).to(env.device)
    before = len(_native.swiglu_bwd_choices())
    m(input_ids=ids, labels=ids, use_cache=False).loss.backward()
    recs = [r for r in _native.swiglu_bwd_choices() if (r["tokens"], r["intermediate"], r["hidden"]) == (80, 320, 192)]
    if env.big and not os.environ.get("TAMD_FUSE_SWIGLU_BWD"):
        assert len(recs) == 1 and recs[0]["fused_ms"] > 0 and recs[0]["two_kernels_ms"] > 0, recs
        faster = "dX GEMM way out" if recs[0]["fused_ms"] <= recs[0]["two_kernels_ms"] else "GEMM + swiglu_bwd kernel"
        assert recs[0]["form"] == faster
        m.zero_grad(set_to_none=True)
        m(input_ids=ids, labels=ids, use_cache=False).loss.backward()  # decided once per shape
        assert len(_native.swiglu_bwd_choices()) == before + 1
    elif not env.big:
        assert not recs


def test_padded_vocab_head_matches_fp32_autograd(env):
    """BERT's MLM head at a vocabulary that is not a multiple of 8 (modeling_bert.py:483-496, 970-975): scores, loss and
    the gradients of hidden / tied weight / bias against torch autograd in fp32 -- through the loss, through a custom
    function of the scores, and through both at once."""
    from transformers_amd.fused_params import PaddedRows

    dev = env.device
    torch.manual_seed(7)
    m, k, v = (1024, 768, 30522) if env.big else (40, 128, 203)
    lin = torch.nn.Linear(k, v).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        lin.bias.normal_(0, 0.5)
    pr = PaddedRows(lin)
    w_pad, b_pad = pr.buffers()
    assert w_pad.shape[0] % 64 == 0 and lin.weight.data_ptr() == w_pad.data_ptr() and lin.weight.shape == (v, k)
    assert not w_pad[v:].any() and not b_pad[v:].any()
    h = _bf(2, m // 2, k, dev=dev, grad=True)
    labels = torch.randint(0, v, (2, m // 2))
    labels[0, ::3] = -100
    gl = (torch.randn(2, m // 2, v) * 1e-3).bfloat16()
    hr, wr, br = (a.detach().float().cpu().requires_grad_(True) for a in (h, lin.weight, lin.bias))
    for use_loss, use_scores in ((True, False), (False, True), (True, True)):
        for t in (h, lin.weight, lin.bias, hr, wr, br):
            t.grad = None
        loss, scores = ops.padded_vocab_head(h, w_pad, b_pad, lin.weight, lin.bias, labels.to(dev) if use_loss else None)
        sr = torch.nn.functional.linear(hr, wr, br)
        assert scores.shape == (2, m // 2, v) and rel_err(scores, sr) < 4e-3
        obj, objr = 0.0, 0.0
        if use_loss:
            lr_ = torch.nn.functional.cross_entropy(sr.view(-1, v), labels.view(-1))
            assert abs(loss.item() - lr_.item()) < 2e-3 * abs(lr_.item())
            obj, objr = obj + loss * 3, objr + lr_ * 3
        else:
            assert loss is None
        if use_scores:
            obj, objr = obj + (scores.float() * gl.to(dev).float()).sum(), objr + (sr * gl.float()).sum()
        obj.backward()
        objr.backward()
        for a, r in ((h, hr), (lin.weight, wr), (lin.bias, br)):
            assert a.grad.shape == r.grad.shape and rel_err(a.grad, r.grad) < 8e-3
    # the parameters are still views of the padded buffers, whose padding stayed zero
    assert pr._coherent() and not w_pad[v:].any() and not b_pad[v:].any()
