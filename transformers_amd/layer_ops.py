"""Layer-level dispatcher ops: a whole reference block as ONE `torch.ops.tamd.*` call (one autograd node).

    tamd::llama_layer / tamd::llama_layer_bwd    LlamaDecoderLayer.forward, models/llama/modeling_llama.py:295-324
    tamd::bert_layer  / tamd::bert_layer_bwd     BertLayer.forward (encoder layer), models/bert/modeling_bert.py:374-416

The forward and backward bodies -- sequences of C-ABI launches on the caller's stream -- are compiled
(csrc/torch_binding.cpp: op_llama_layer, op_llama_layer_bwd, op_bert_layer, op_bert_layer_bwd; the environment switches
TAMD_FUSE_ROPE_BWD / TAMD_SAVE_SWIGLU_ACT are read there).  This module registers their fake (Meta)
implementations and autograd formulas and holds the wrappers the model classes in `transformers_amd/models/` call.
"""
from __future__ import annotations

import os

import torch

from . import ddp, ops
from .ops import register

T = torch.ops.tamd
_SAVE_ACT = os.environ.get("TAMD_SAVE_SWIGLU_ACT", "1") != "0"  # (mirrors torch_binding.cpp: shape of the fake output)


def _llama_layer_fake(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wq, wk, wv, wo, w_ln2, wgu, wg, wu, wd, eps, hq,
                      hkv, d, scale, causal, train):
    b, s, hd = h_in.shape
    t = b * s
    h_out = h_in.new_empty(b, s, hd)
    if not train:
        return (h_out,) + tuple(h_in.new_empty(0) for _ in range(10))
    f32 = dict(dtype=torch.float32)
    return (h_out, h_in.new_empty(t, hd), h_in.new_empty(t, wqkv.shape[0]), h_in.new_empty(b, s, hq, d),
            h_in.new_empty(b, hq, s, **f32), h_in.new_empty(t, hd), h_in.new_empty(t, hd),
            h_in.new_empty(t, wgu.shape[0]), h_in.new_empty(t, **f32), h_in.new_empty(t, **f32),
            h_in.new_empty(t, wgu.shape[0] // 2) if _SAVE_ACT else h_in.new_empty(0))


def _llama_layer_bwd_fake(d_hout, h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wo, w_ln2, wgu, wd, rstd1, xn, qkv,
                          o, lse, h_mid, rstd2, xn2, gu, act_saved, hq, hkv, d, scale, causal, dst_q=None, dst_k=None,
                          dst_v=None, dst_o=None, dst_g=None, dst_u=None, dst_d=None):
    e = (lambda w: h_in.new_empty(0)) if dst_q is not None else torch.empty_like  # (with destinations: written there)
    return (torch.empty_like(h_in, memory_format=torch.contiguous_format), torch.empty_like(w_ln1),
            e(wqkv), e(wo), torch.empty_like(w_ln2), e(wgu), e(wd))


def _llama_layer_setup(ctx, inputs, output):
    (h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, _wq, _wk, _wv, wo, w_ln2, wgu, _wg, _wu, wd, _eps, hq, hkv, d,
     scale, causal, train) = inputs
    _h_out, xn, qkv, o, lse, h_mid, xn2, gu, rstd1, rstd2, act = output
    ctx.save_for_backward(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wo, w_ln2, wgu, wd, rstd1, xn, qkv, o, lse,
                          h_mid, rstd2, xn2, gu, act)
    ctx.meta = (hq, hkv, d, scale, causal, train)
    ctx.weights = (_wq, _wk, _wv, wo, _wg, _wu, wd)  # the parameters themselves: ddp.destinations looks their bucket views up
    ctx.set_materialize_grads(False)


def _llama_layer_backward(ctx, d_hout, *_aux):
    none = (None,) * 23
    if d_hout is None:
        return none
    hq, hkv, d, scale, causal, train = ctx.meta
    if not train:
        raise ops.TamdError("llama_layer was run with train=False but is being differentiated")
    # under DistributedDataParallel with ddp.enable_zero_copy: the dW GEMMs write straight into the all-reduce buckets
    dst = ddp.destinations(ctx.weights) if ddp._VIEWS else None
    if dst is not None:
        d_hin, dw_ln1, _, _, dw_ln2, _, _ = T.llama_layer_bwd(d_hout, *ctx.saved_tensors, hq, hkv, d, scale, causal, *dst)
        gq, gk, gv, go, gg, gu_, gd = (t.detach() for t in dst)  # fresh aliases: AccumulateGrad adopts them without a copy
        return (d_hin, None, None, None, None, dw_ln1, None, gq, gk, gv, go, dw_ln2, None, gg, gu_, gd) + none[16:]
    d_hin, dw_ln1, dwqkv, dwo, dw_ln2, dwgu, dwd = T.llama_layer_bwd(d_hout, *ctx.saved_tensors, hq, hkv, d, scale,
                                                                      causal)
    nq, nk = hq * d, hkv * d
    inter = dwgu.shape[0] // 2
    #       h_in   cos   sin   kv    qs    w_ln1   wqkv  wq          wk                 wv               wo
    return (d_hin, None, None, None, None, dw_ln1, None, dwqkv[:nq], dwqkv[nq:nq + nk], dwqkv[nq + nk:], dwo,
            dw_ln2, None, dwgu[:inter], dwgu[inter:], dwd) + none[16:]


register("llama_layer", _llama_layer_fake, _llama_layer_backward, _llama_layer_setup)
register("llama_layer_bwd", _llama_layer_bwd_fake)


def llama_layer(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wq, wk, wv, wo, w_ln2, wgu, wg, wu, wd, *, eps, hq,
                hkv, d, scale, causal):
    """LlamaDecoderLayer.forward as one op.  `wqkv` / `wgu` are the fused buffers the member parameters `wq, wk, wv` /
    `wg, wu` are row-slice views of (fused_params.py): the kernels read the fused buffers, the gradients go to the
    members."""
    train = ops._wants_grad(h_in, w_ln1, wq, wk, wv, wo, w_ln2, wg, wu, wd)
    return T.llama_layer(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wq, wk, wv, wo, w_ln2, wgu, wg, wu, wd,
                         float(eps), int(hq), int(hkv), int(d), float(scale), bool(causal), train)[0]


# ============================================================================================ BertLayer as one op
# (what the op computes: csrc/torch_binding.cpp op_bert_layer / op_bert_layer_bwd)
def _bert_layer_fake(h_in, key_valid, wqkv, bqkv, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w,
                     ln2_b, eps, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train, seeds_dev=None):
    b, s, hd = h_in.shape
    t = b * s
    out = h_in.new_empty(b, s, hd)
    if not train:
        return (out,) + tuple(h_in.new_empty(0) for _ in range(12))
    f32 = dict(dtype=torch.float32)
    return (out, h_in.new_empty(t, 3 * hd), h_in.new_empty(b, s, heads, d), h_in.new_empty(b, heads, s, **f32),
            h_in.new_empty(t, hd), h_in.new_empty(t, **f32), h_in.new_empty(t, **f32), h_in.new_empty(t, hd),
            h_in.new_empty(t, wi.shape[0]), h_in.new_empty(t, wi.shape[0]), h_in.new_empty(t, hd),
            h_in.new_empty(t, **f32), h_in.new_empty(t, **f32))


def _bert_layer_bwd_fake(d_out, h_in, key_valid, wqkv, wo, ln1_w, wi, wo2, ln2_w, qkv, o, lse, y1, mean1, rstd1, h1, pre,
                         inter, y2, mean2, rstd2, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, seeds_dev=None):
    vec = lambda w: w.new_empty(w.shape[0])  # noqa: E731
    return (torch.empty_like(h_in, memory_format=torch.contiguous_format), torch.empty_like(wqkv), vec(wqkv),
            torch.empty_like(wo), vec(wo), torch.empty_like(ln1_w), torch.empty_like(ln1_w), torch.empty_like(wi), vec(wi),
            torch.empty_like(wo2), vec(wo2), torch.empty_like(ln2_w), torch.empty_like(ln2_w))


def _bert_layer_setup(ctx, inputs, output):
    (h_in, key_valid, wqkv, _bqkv, _wq, _wk, _wv, _bq, _bk, _bv, wo, _bo, ln1_w, _ln1_b, wi, _bi, wo2, _bo2, ln2_w, _ln2_b,
     _eps, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train, seeds_dev) = inputs
    _out, qkv, o, lse, y1, mean1, rstd1, h1, pre, inter, y2, mean2, rstd2 = output
    ctx.save_for_backward(h_in, key_valid, wqkv, wo, ln1_w, wi, wo2, ln2_w, qkv, o, lse, y1, mean1, rstd1, h1, pre, inter,
                          y2, mean2, rstd2, seeds_dev)
    ctx.meta = (heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train)
    ctx.set_materialize_grads(False)


def _bert_layer_backward(ctx, d_out, *_aux):
    none = (None,) * 32
    if d_out is None:
        return none
    heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train = ctx.meta
    if not train:
        raise ops.TamdError("bert_layer was run with train=False but is being differentiated")
    *saved, seeds_dev = ctx.saved_tensors
    (d_x, dwqkv, dbqkv, dwo, dbo, dw_ln1, db_ln1, dwi, dbi, dwo2, dbo2, dw_ln2, db_ln2) = T.bert_layer_bwd(
        d_out, *saved, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, seeds_dev)
    hd = dwqkv.shape[1]
    #       h_in kv    wqkv  bqkv  wq          wk                wv              bq          bk                bv
    return (d_x, None, None, None, dwqkv[:hd], dwqkv[hd:2 * hd], dwqkv[2 * hd:], dbqkv[:hd], dbqkv[hd:2 * hd], dbqkv[2 * hd:],
            dwo, dbo, dw_ln1, db_ln1, dwi, dbi, dwo2, dbo2, dw_ln2, db_ln2) + none[20:]


register("bert_layer", _bert_layer_fake, _bert_layer_backward, _bert_layer_setup)
register("bert_layer_bwd", _bert_layer_bwd_fake)


def bert_layer(h_in, key_valid, wqkv, bqkv, members, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w, ln2_b, *, eps, heads, d,
               scale, act, p_attn, p_hidden):
    """BertLayer.forward (encoder layer) as one op.  `wqkv` / `bqkv` are the fused buffers the query / key / value
    parameters `members` = (wq, wk, wv, bq, bk, bv) are row-slice views of (fused_params.py).  Dropout seeds come from
    torch's generators (ops.dropout_seeds): `torch.manual_seed` repeats them, checkpointing regenerates them, a step captured
    in a HIP graph draws fresh ones on every replay."""
    wq, wk, wv, bq, bk, bv = members
    train = ops._wants_grad(h_in, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w, ln2_b)
    seed_attn = seed1 = seed2 = 0
    seeds_dev = None
    if p_attn > 0.0 or p_hidden > 0.0:
        # eager: one host seed per ACTIVE site, drawn in the order attention, hidden 1, hidden 2; while a graph is being
        # captured: three device words (site = index)
        n = (1 if p_attn > 0.0 else 0) + (2 if p_hidden > 0.0 else 0)
        host, seeds_dev = ops.dropout_seeds(3 if ops.capturing(h_in.device) else n, h_in.device)
        if seeds_dev is None:
            it = iter(host)
            seed_attn = next(it) if p_attn > 0.0 else 0
            seed1, seed2 = (next(it), next(it)) if p_hidden > 0.0 else (0, 0)
    return T.bert_layer(h_in, key_valid, wqkv, bqkv, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w,
                        ln2_b, float(eps), int(heads), int(d), float(scale), int(act), float(p_attn), float(p_hidden),
                        int(seed_attn), int(seed1), int(seed2), train, seeds_dev)[0]
