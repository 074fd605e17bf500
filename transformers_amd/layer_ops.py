"""Layer-level dispatcher ops: a whole reference block as ONE `torch.ops.tamd.*` call (one autograd node).

    tamd::llama_layer / tamd::llama_layer_bwd    LlamaDecoderLayer.forward, models/llama/modeling_llama.py:295-324

The implementations below are the op kernels: sequences of C-ABI launches (ops.raw_*) on the caller's stream.  The
model classes in `transformers_amd/models/` only call `torch.ops.tamd.llama_layer(...)`.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .ops import EPI_RESIDUAL, define_op

T = torch.ops.tamd
# Rotary embedding in the epilogues (both bit-identical to the two-kernel paths; A/B switches for measurements).
# Backward (transposed rotary on dq / dk inside the attention backward): ON -- saves the 118-us rotary kernel for
# ~50-70 us of epilogue.  Forward (tamd_gemm_rope): OFF until the rewritten way out is measured -- across eight profiled
# runs of round 2 the fused q|k|v GEMM took 1377-1518 us where GEMM + rope_kernel took 1285-1372 us (normalised by the
# gate|up GEMM of the same run: 0.245-0.267 against 0.225-0.234): its first way out divided a 64-bit index per row
# segment and loaded cos / sin inside the row loop; the rewrite (csrc/gemm.hip gemm_epilogue_rows, a third fewer
# instructions) is what TAMD_FUSE_ROPE_FWD=1 selects.  profiles/r02_gemm_variants.md section 7
_FUSE_ROPE_FWD = os.environ.get("TAMD_FUSE_ROPE_FWD", "0") == "1"
_FUSE_ROPE_BWD = os.environ.get("TAMD_FUSE_ROPE_BWD", "1") != "0"
# (Round 2 also carried the SwiGLU backward as an epilogue of the d_act GEMM: 0.2 ms per layer faster than GEMM +
# swiglu_bwd_kernel in its good regime, 1.9 ms slower in its bad one -- profiles/r02_regression_note.md -- and, with one
# workgroup per CU, an epilogue that streams four large arrays has nothing to overlap with: removed in round 3.)
# The SiLU*up product `act` [T, I] (the down projection's input, needed again for its weight gradient) is KEPT for the
# backward instead of re-materialised there: the gate|up GEMM's epilogue writes it anyway, so keeping it costs no time
# and T*I*2 bytes per layer (0.94 GB at the Llama-3-8B shape: 30 GB over 32 layers on a 288 GB part, peak 141 -> 171 GB),
# and the HBM-bound SwiGLU backward kernel writes 4.70 instead of 5.64 GB.  TAMD_SAVE_SWIGLU_ACT=0 re-materialises.
_SAVE_ACT = os.environ.get("TAMD_SAVE_SWIGLU_ACT", "1") != "0"


def _split_qkv(qkv, b, s, hq, hkv, d):
    q = qkv[:, : hq * d].view(b, s, hq, d)
    k = qkv[:, hq * d: (hq + hkv) * d].view(b, s, hkv, d)
    v = qkv[:, (hq + hkv) * d:].view(b, s, hkv, d)
    return q, k, v


# forward : rmsnorm -> QKV GEMM, rotary (kernel, or the GEMM's epilogue) -> attention -> o_proj GEMM(+residual)
#           -> rmsnorm -> gate|up GEMM with the SwiGLU epilogue -> down GEMM(+residual)
def _llama_layer_impl(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wq, wk, wv, wo, w_ln2, wgu, wg, wu, wd, eps,
                      hq, hkv, d, scale, causal, train):
    b, s, hd = h_in.shape
    t = b * s
    x = ops._c(h_in).view(t, hd)
    xn, _, rstd1 = ops.raw_rmsnorm_fwd(x, w_ln1, eps)
    if _FUSE_ROPE_FWD and ops.gemm_rope_supported(xn, wqkv, cos, d):  # apply_rotary_pos_emb in the q|k|v GEMM epilogue
        qkv = ops.raw_gemm_rope(xn, wqkv, cos, sin, s, hq + hkv, d)
    else:
        qkv = ops.raw_gemm(xn, wqkv)
        ops.raw_rope_(qkv, cos, sin, s, hq + hkv, d)
    q, k, v = _split_qkv(qkv, b, s, hq, hkv, d)
    o, lse = ops.raw_attn_fwd(q, k, v, scale, causal, key_valid, need_lse=train, q_start=q_start)
    h_mid = ops.raw_gemm(o.view(t, hq * d), wo, residual=x, epilogue=EPI_RESIDUAL)
    xn2, _, rstd2 = ops.raw_rmsnorm_fwd(h_mid, w_ln2, eps)
    if ops.gemm_swiglu_supported(xn2, wgu):  # SiLU*up in the gate|up GEMM epilogue; gate|up itself only if saved
        gu, act = ops.raw_gemm_swiglu(xn2, wgu, need_gu=train)
    else:
        gu = ops.raw_gemm(xn2, wgu)
        act = ops.raw_swiglu_fwd(gu)
    h_out = ops.raw_gemm(act, wd, residual=h_mid, epilogue=EPI_RESIDUAL).view(b, s, hd)
    if not train:
        e = h_out.new_empty(0)
        return (h_out, e, e.clone(), e.clone(), e.clone(), e.clone(), e.clone(), e.clone(), e.clone(), e.clone(),
                e.clone())
    return h_out, xn, qkv, o, lse, h_mid, xn2, gu, rstd1, rstd2, (act if _SAVE_ACT else h_out.new_empty(0))


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


# backward: the derivatives of SURVEY.md section 8a in reverse; every weight gradient is a k-major GEMM on the saved
# activations, the SiLU*up product comes from the forward (`act_saved`, _SAVE_ACT) or is re-materialised (an empty
# `act_saved`), and the residual-stream gradient is folded into the RMSNorm backward kernels (`dres`).  Nothing saved
# by the forward is written (a retained graph can be differentiated twice).
def _llama_layer_bwd_impl(d_hout, h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wo, w_ln2, wgu, wd, rstd1, xn, qkv, o,
                          lse, h_mid, rstd2, xn2, gu, act_saved, hq, hkv, d, scale, causal):
    b, s, hd = h_in.shape
    t = b * s
    x = ops._c(h_in).view(t, hd)
    dh = ops._c(d_hout).view(t, hd)
    # ---- MLP
    d_act = ops.raw_gemm(dh, wd, b_kn=True)                                  # [T, I]
    if act_saved.numel():
        d_gu, act = ops.raw_swiglu_bwd(gu, d_act, want_act=False)[0], act_saved
    else:
        d_gu, act = ops.raw_swiglu_bwd(gu, d_act, want_act=True)
    del d_act
    dwd = ops.raw_gemm(dh, act, a_km=True, b_kn=True)                        # [hd, I]
    del act
    d_xn2 = ops.raw_gemm(d_gu, wgu, b_kn=True)                               # [T, hd]
    dwgu = ops.raw_gemm(d_gu, xn2, a_km=True, b_kn=True)                     # [2I, hd]
    del d_gu
    d_hmid, dw_ln2 = ops.raw_rmsnorm_bwd(d_xn2, h_mid, w_ln2, rstd2, dres=dh)
    del d_xn2
    # ---- attention
    d_o = ops.raw_gemm(d_hmid, wo, b_kn=True)                                # [T, Hq*D]
    dwo = ops.raw_gemm(d_hmid, o.view(t, hq * d), a_km=True, b_kn=True)
    d_qkv = torch.empty_like(qkv)
    q, k, v = _split_qkv(qkv, b, s, hq, hkv, d)
    dq, dk, dv = _split_qkv(d_qkv, b, s, hq, hkv, d)
    fused_rope = _FUSE_ROPE_BWD and ops.attn_bwd_rope_supported(q, k, cos, d)  # the transposed rotary on dq / dk inside the kernels
    ops.raw_attn_bwd(q, k, v, o, lse, d_o.view(b, s, hq, d), scale, causal, key_valid, dq=dq, dk=dk, dv=dv,
                     q_start=q_start, rope=(cos, sin) if fused_rope else None)
    del d_o
    if not fused_rope:
        ops.raw_rope_(d_qkv, cos, sin, s, hq + hkv, d, conj=True)
    d_xn = ops.raw_gemm(d_qkv, wqkv, b_kn=True)
    dwqkv = ops.raw_gemm(d_qkv, xn, a_km=True, b_kn=True)                    # [(Hq+2Hkv)D, hd]
    d_hin, dw_ln1 = ops.raw_rmsnorm_bwd(d_xn, x, w_ln1, rstd1, dres=d_hmid)
    return d_hin.view(b, s, hd), dw_ln1, dwqkv, dwo, dw_ln2, dwgu, dwd


def _llama_layer_bwd_fake(d_hout, h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wo, w_ln2, wgu, wd, rstd1, xn, qkv,
                          o, lse, h_mid, rstd2, xn2, gu, act_saved, hq, hkv, d, scale, causal):
    return (torch.empty_like(h_in, memory_format=torch.contiguous_format), torch.empty_like(w_ln1),
            torch.empty_like(wqkv), torch.empty_like(wo), torch.empty_like(w_ln2), torch.empty_like(wgu),
            torch.empty_like(wd))


def _llama_layer_setup(ctx, inputs, output):
    (h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, _wq, _wk, _wv, wo, w_ln2, wgu, _wg, _wu, wd, _eps, hq, hkv, d,
     scale, causal, train) = inputs
    _h_out, xn, qkv, o, lse, h_mid, xn2, gu, rstd1, rstd2, act = output
    ctx.save_for_backward(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wo, w_ln2, wgu, wd, rstd1, xn, qkv, o, lse,
                          h_mid, rstd2, xn2, gu, act)
    ctx.meta = (hq, hkv, d, scale, causal, train)
    ctx.set_materialize_grads(False)


def _llama_layer_backward(ctx, d_hout, *_aux):
    none = (None,) * 23
    if d_hout is None:
        return none
    hq, hkv, d, scale, causal, train = ctx.meta
    if not train:
        raise ops.TamdError("llama_layer was run with train=False but is being differentiated")
    d_hin, dw_ln1, dwqkv, dwo, dw_ln2, dwgu, dwd = T.llama_layer_bwd(d_hout, *ctx.saved_tensors, hq, hkv, d, scale,
                                                                      causal)
    nq, nk = hq * d, hkv * d
    inter = dwgu.shape[0] // 2
    #       h_in   cos   sin   kv    qs    w_ln1   wqkv  wq          wk                 wv               wo
    return (d_hin, None, None, None, None, dw_ln1, None, dwqkv[:nq], dwqkv[nq:nq + nk], dwqkv[nq + nk:], dwo,
            dw_ln2, None, dwgu[:inter], dwgu[inter:], dwd) + none[16:]


define_op("llama_layer(Tensor h_in, Tensor cos, Tensor sin, Tensor? key_valid, Tensor? q_start, Tensor w_ln1, "
          "Tensor wqkv, Tensor wq, Tensor wk, Tensor wv, Tensor wo, Tensor w_ln2, Tensor wgu, Tensor wg, Tensor wu, "
          "Tensor wd, float eps, int hq, int hkv, int d, float scale, bool causal, bool train) -> "
          "(Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
          _llama_layer_impl, _llama_layer_fake, _llama_layer_backward, _llama_layer_setup)
define_op("llama_layer_bwd(Tensor d_hout, Tensor h_in, Tensor cos, Tensor sin, Tensor? key_valid, Tensor? q_start, "
          "Tensor w_ln1, Tensor wqkv, Tensor wo, Tensor w_ln2, Tensor wgu, Tensor wd, Tensor rstd1, Tensor xn, "
          "Tensor qkv, Tensor o, Tensor lse, Tensor h_mid, Tensor rstd2, Tensor xn2, Tensor gu, Tensor act_saved, int hq, "
          "int hkv, int d, float scale, bool causal) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
          _llama_layer_bwd_impl, _llama_layer_bwd_fake)


def llama_layer(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wq, wk, wv, wo, w_ln2, wgu, wg, wu, wd, *, eps, hq,
                hkv, d, scale, causal):
    """LlamaDecoderLayer.forward as one op.  `wqkv` / `wgu` are the fused buffers the member parameters `wq, wk, wv` /
    `wg, wu` are row-slice views of (fused_params.py): the kernels read the fused buffers, the gradients go to the
    members."""
    train = ops._wants_grad(h_in, w_ln1, wq, wk, wv, wo, w_ln2, wg, wu, wd)
    return T.llama_layer(h_in, cos, sin, key_valid, q_start, w_ln1, wqkv, wq, wk, wv, wo, w_ln2, wgu, wg, wu, wd,
                         float(eps), int(hq), int(hkv), int(d), float(scale), bool(causal), train)[0]


# ============================================================================================ BertLayer as one op
# tamd::bert_layer / tamd::bert_layer_bwd    BertLayer.forward (encoder layer: BertAttention -> BertIntermediate ->
# BertOutput), models/bert/modeling_bert.py:164-203, 289-293, 334-351, 374-416.  Post-LN blocks with biases:
#     y1 = dropout(attn_out . Wo^T + bo) + x ;   h1 = LayerNorm1(y1)
#     y2 = dropout(act(h1 . Wi^T + bi) . Wo2^T + bo2) + h1 ;   out = LayerNorm2(y2)
# forward : fused q|k|v GEMM(+bias) -> attention (dropout inside) -> dense GEMM -> [dropout+]add+LayerNorm -> GEMM(+bias[,
#           +act]) -> dense GEMM -> [dropout+]add+LayerNorm.  Without hidden dropout the residual adds ride in the dense
#           GEMMs' epilogues; with it the add joins the dropout + LayerNorm kernel.
# backward: the same chain reversed; the two places where a tensor feeds both a projection and a residual (x, h1) get their
#           gradient sum from the residual epilogue of the dX GEMM -- as separate ops autograd adds them (49 `at::add`
#           launches per bert-base step, 4 % of it) and rebuilds d_qkv from three slices (36 fills + copies).
def _bert_layer_impl(h_in, key_valid, wqkv, bqkv, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w,
                     ln2_b, eps, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train):
    b, s, hd = h_in.shape
    t = b * s
    x = ops._c(h_in).view(t, hd)
    qkv = ops.raw_gemm(x, wqkv, bias=bqkv, epilogue=ops.EPI_BIAS)
    q, k, v = _split_qkv(qkv, b, s, heads, heads, d)
    o, lse = ops.raw_attn_fwd(q, k, v, scale, False, key_valid, need_lse=train, dropout_p=p_attn, seed=seed_attn)
    o2 = o.view(t, hd)

    def dense_add_ln(inp, w, bias, res, ln_w, ln_b, seed):
        if p_hidden > 0.0:
            a = ops.raw_gemm(inp, w, bias=bias, epilogue=ops.EPI_BIAS)
            return ops.raw_layernorm_dropout_fwd(a, ln_w, ln_b, eps, res, p_hidden, seed)   # (y, pre-norm sum, mean, rstd)
        y = ops.raw_gemm(inp, w, bias=bias, residual=res, epilogue=EPI_RESIDUAL)
        out, _, mean, rstd = ops.raw_layernorm_fwd(y, ln_w, ln_b, eps)
        return out, y, mean, rstd

    h1, y1, mean1, rstd1 = dense_add_ln(o2, wo, bo, x, ln1_w, ln1_b, seed1)
    if train:  # the pre-activation is what the activation's backward needs
        pre = ops.raw_gemm(h1, wi, bias=bi, epilogue=ops.EPI_BIAS)
        inter = ops.raw_bias_act_fwd(pre, None, act)
    else:
        pre = h_in.new_empty(0)
        inter = ops.raw_gemm(h1, wi, bias=bi, epilogue=ops.EPI_BIAS_ACT, act=act)
    out, y2, mean2, rstd2 = dense_add_ln(inter, wo2, bo2, h1, ln2_w, ln2_b, seed2)
    out = out.view(b, s, hd)
    if not train:
        e = h_in.new_empty(0)
        return (out,) + tuple(e.clone() for _ in range(12))
    return out, qkv, o, lse, y1, mean1, rstd1, h1, pre, inter, y2, mean2, rstd2


def _bert_layer_fake(h_in, key_valid, wqkv, bqkv, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w,
                     ln2_b, eps, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train):
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


def _bert_layer_bwd_impl(d_out, h_in, key_valid, wqkv, wo, ln1_w, wi, wo2, ln2_w, qkv, o, lse, y1, mean1, rstd1, h1, pre,
                         inter, y2, mean2, rstd2, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2):
    b, s, hd = h_in.shape
    t = b * s
    x = ops._c(h_in).view(t, hd)
    dy = ops._c(d_out).view(t, hd)

    def ln_bwd(g, y, ln_w, mean, rstd, seed):  # -> (gradient of the residual input, of the dense output, dw, db)
        if p_hidden > 0.0:
            return ops.raw_layernorm_dropout_bwd(g, y, ln_w, mean, rstd, p_hidden, seed)
        dx, dw, db = ops.raw_layernorm_bwd(g, y, ln_w, mean, rstd)
        return dx, dx, dw, db

    # ---- BertOutput / BertIntermediate
    d_h1_res, d_b, dw_ln2, db_ln2 = ln_bwd(dy, y2, ln2_w, mean2, rstd2, seed2)
    dbo2 = ops.raw_colsum(d_b)
    dwo2 = ops.raw_gemm(d_b, inter, a_km=True, b_kn=True)                    # [hd, I]
    d_inter = ops.raw_gemm(d_b, wo2, b_kn=True)                              # [T, I]
    d_pre = ops.raw_bias_act_bwd(pre, None, d_inter, act)
    del d_inter
    dbi = ops.raw_colsum(d_pre)
    dwi = ops.raw_gemm(d_pre, h1, a_km=True, b_kn=True)                      # [I, hd]
    d_h1 = ops.raw_gemm(d_pre, wi, b_kn=True, residual=d_h1_res, epilogue=EPI_RESIDUAL)  # + the residual path's gradient
    del d_pre
    # ---- BertSelfOutput / BertSelfAttention
    d_x_res, d_a, dw_ln1, db_ln1 = ln_bwd(d_h1, y1, ln1_w, mean1, rstd1, seed1)
    dbo = ops.raw_colsum(d_a)
    dwo = ops.raw_gemm(d_a, o.view(t, hd), a_km=True, b_kn=True)
    d_o = ops.raw_gemm(d_a, wo, b_kn=True)
    d_qkv = torch.empty_like(qkv)
    q, k, v = _split_qkv(qkv, b, s, heads, heads, d)
    dq, dk, dv = _split_qkv(d_qkv, b, s, heads, heads, d)
    ops.raw_attn_bwd(q, k, v, o, lse, d_o.view(b, s, heads, d), scale, False, key_valid, dq=dq, dk=dk, dv=dv,
                     dropout_p=p_attn, seed=seed_attn)
    del d_o
    dbqkv = ops.raw_colsum(d_qkv)
    dwqkv = ops.raw_gemm(d_qkv, x, a_km=True, b_kn=True)                     # [3 hd, hd]
    d_x = ops.raw_gemm(d_qkv, wqkv, b_kn=True, residual=d_x_res, epilogue=EPI_RESIDUAL)
    return (d_x.view(b, s, hd), dwqkv, dbqkv, dwo, dbo, dw_ln1, db_ln1, dwi, dbi, dwo2, dbo2, dw_ln2, db_ln2)


def _bert_layer_bwd_fake(d_out, h_in, key_valid, wqkv, wo, ln1_w, wi, wo2, ln2_w, qkv, o, lse, y1, mean1, rstd1, h1, pre,
                         inter, y2, mean2, rstd2, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2):
    vec = lambda w: w.new_empty(w.shape[0])  # noqa: E731
    return (torch.empty_like(h_in, memory_format=torch.contiguous_format), torch.empty_like(wqkv), vec(wqkv),
            torch.empty_like(wo), vec(wo), torch.empty_like(ln1_w), torch.empty_like(ln1_w), torch.empty_like(wi), vec(wi),
            torch.empty_like(wo2), vec(wo2), torch.empty_like(ln2_w), torch.empty_like(ln2_w))


def _bert_layer_setup(ctx, inputs, output):
    (h_in, key_valid, wqkv, _bqkv, _wq, _wk, _wv, _bq, _bk, _bv, wo, _bo, ln1_w, _ln1_b, wi, _bi, wo2, _bo2, ln2_w, _ln2_b,
     _eps, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train) = inputs
    _out, qkv, o, lse, y1, mean1, rstd1, h1, pre, inter, y2, mean2, rstd2 = output
    ctx.save_for_backward(h_in, key_valid, wqkv, wo, ln1_w, wi, wo2, ln2_w, qkv, o, lse, y1, mean1, rstd1, h1, pre, inter,
                          y2, mean2, rstd2)
    ctx.meta = (heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train)
    ctx.set_materialize_grads(False)


def _bert_layer_backward(ctx, d_out, *_aux):
    none = (None,) * 31
    if d_out is None:
        return none
    heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2, train = ctx.meta
    if not train:
        raise ops.TamdError("bert_layer was run with train=False but is being differentiated")
    (d_x, dwqkv, dbqkv, dwo, dbo, dw_ln1, db_ln1, dwi, dbi, dwo2, dbo2, dw_ln2, db_ln2) = T.bert_layer_bwd(
        d_out, *ctx.saved_tensors, heads, d, scale, act, p_attn, p_hidden, seed_attn, seed1, seed2)
    hd = dwqkv.shape[1]
    #       h_in kv    wqkv  bqkv  wq          wk                wv              bq          bk                bv
    return (d_x, None, None, None, dwqkv[:hd], dwqkv[hd:2 * hd], dwqkv[2 * hd:], dbqkv[:hd], dbqkv[hd:2 * hd], dbqkv[2 * hd:],
            dwo, dbo, dw_ln1, db_ln1, dwi, dbi, dwo2, dbo2, dw_ln2, db_ln2) + none[20:]


define_op("bert_layer(Tensor h_in, Tensor? key_valid, Tensor wqkv, Tensor bqkv, Tensor wq, Tensor wk, Tensor wv, Tensor bq, "
          "Tensor bk, Tensor bv, Tensor wo, Tensor bo, Tensor ln1_w, Tensor ln1_b, Tensor wi, Tensor bi, Tensor wo2, "
          "Tensor bo2, Tensor ln2_w, Tensor ln2_b, float eps, int heads, int d, float scale, int act, float p_attn, "
          "float p_hidden, int seed_attn, int seed1, int seed2, bool train) -> (Tensor, Tensor, Tensor, Tensor, Tensor, "
          "Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
          _bert_layer_impl, _bert_layer_fake, _bert_layer_backward, _bert_layer_setup)
define_op("bert_layer_bwd(Tensor d_out, Tensor h_in, Tensor? key_valid, Tensor wqkv, Tensor wo, Tensor ln1_w, Tensor wi, "
          "Tensor wo2, Tensor ln2_w, Tensor qkv, Tensor o, Tensor lse, Tensor y1, Tensor mean1, Tensor rstd1, Tensor h1, "
          "Tensor pre, Tensor inter, Tensor y2, Tensor mean2, Tensor rstd2, int heads, int d, float scale, int act, "
          "float p_attn, float p_hidden, int seed_attn, int seed1, int seed2) -> (Tensor, Tensor, Tensor, Tensor, Tensor, "
          "Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)",
          _bert_layer_bwd_impl, _bert_layer_bwd_fake)


def bert_layer(h_in, key_valid, wqkv, bqkv, members, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w, ln2_b, *, eps, heads, d,
               scale, act, p_attn, p_hidden):
    """BertLayer.forward (encoder layer) as one op.  `wqkv` / `bqkv` are the fused buffers the query / key / value
    parameters `members` = (wq, wk, wv, bq, bk, bv) are row-slice views of (fused_params.py).  Dropout seeds come from
    torch's CPU generator (ops.dropout_seed): `torch.manual_seed` repeats them, checkpointing regenerates them."""
    wq, wk, wv, bq, bk, bv = members
    train = ops._wants_grad(h_in, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w, ln2_b)
    seed_attn = ops.dropout_seed() if p_attn > 0.0 else 0
    seed1, seed2 = (ops.dropout_seed(), ops.dropout_seed()) if p_hidden > 0.0 else (0, 0)
    return T.bert_layer(h_in, key_valid, wqkv, bqkv, wq, wk, wv, bq, bk, bv, wo, bo, ln1_w, ln1_b, wi, bi, wo2, bo2, ln2_w,
                        ln2_b, float(eps), int(heads), int(d), float(scale), int(act), float(p_attn), float(p_hidden),
                        int(seed_attn), int(seed1), int(seed2), train)[0]
