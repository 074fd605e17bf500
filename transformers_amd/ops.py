"""torch-level entry points of the MI355X kernels.

Three layers, all thin:

1. ``raw_*``   -- one Python function per C-ABI entry point (include/tamd.h): argument checks,
                 output allocation with torch's caching allocator, launch on the *current* HIP
                 stream of the calling thread.  No autograd.
2. ``torch.ops.tamd.*`` -- the same functions registered with ``torch.library`` (the reference's own
                 precedent: src/transformers/integrations/moe.py:245-257), with Meta ("fake")
                 implementations so they compose with torch's tooling.
3. ``*Fn``     -- ``torch.autograd.Function``s implementing the backward contract of SURVEY.md §8a.

The HIP library is mandatory: there is no CPU or eager fallback in this module.  If libtamd.so is
missing, or a tensor is not on a GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Optional

import torch

from . import _cabi
from ._cabi import (ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, ACT_QUICK_GELU, ACT_SILU, EPI_ACCUM, EPI_BIAS,
                    EPI_BIAS_ACT, EPI_NONE, EPI_RESIDUAL, GEMM_A_KM, GEMM_B_KN, TamdError)

_DTYPE_CODE = {torch.bfloat16: _cabi.TAMD_BF16, torch.float16: _cabi.TAMD_F16, torch.float32: _cabi.TAMD_F32}
ACT_CODES = {"none": ACT_NONE, "gelu": ACT_GELU_ERF, "gelu_new": ACT_GELU_TANH, "gelu_pytorch_tanh": ACT_GELU_TANH,
             "quick_gelu": ACT_QUICK_GELU, "silu": ACT_SILU, "swish": ACT_SILU}


# --------------------------------------------------------------------------- backend
class HipBackend:
    """libtamd.so + the calling thread's current HIP stream."""

    name = "hip"

    def __init__(self):
        path = _cabi.default_library_path()
        if not path.exists():
            raise TamdError(
                f"{path} not found: build it with `python -m transformers_amd.build` "
                "(the MI355X path has no CPU/eager fallback)")
        self.lib = _cabi.TamdLib(path)

    def check_tensor(self, t: torch.Tensor) -> None:
        if not t.is_cuda:
            raise TamdError(f"tamd op received a {t.device} tensor; the HIP kernels need GPU memory")

    def stream(self, t: torch.Tensor):
        # raw handle of torch's current stream on the tensor's device (no Stream object: this runs once per kernel)
        idx = t.device.index
        if _RAW_STREAM is not None:
            return _RAW_STREAM(idx if idx is not None else torch.cuda.current_device())
        return torch.cuda.current_stream(t.device).cuda_stream


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_backend = None
_backend_lock = threading.Lock()


def backend():
    global _backend
    if _backend is None:
        with _backend_lock:
            if _backend is None:
                _backend = HipBackend()
    return _backend


def _set_backend(b):
    """Test hook (tests/hipemu installs the CPU execution model of the same kernels here)."""
    global _backend
    old, _backend = _backend, b
    return old


def backend_is_emulated() -> bool:
    """True only while tests/hipemu has installed the CPU execution model of the kernels."""
    return _backend is not None and _backend.name != "hip"


def _code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise TamdError(f"unsupported dtype {t.dtype}") from None


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()  # ctypes converts the int for `c_void_p` argtypes


def _prep(*tensors):
    be = backend()
    for t in tensors:
        if t is not None:
            be.check_tensor(t)
    return be


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- raw ops
def raw_rmsnorm_fwd(x, w, eps, residual=None):
    """-> (y, h, rstd); h is x+residual (or x itself when residual is None)."""
    cols = x.shape[-1]
    x2 = _c(x).view(-1, cols)
    r2 = None if residual is None else _c(residual).view(-1, cols)
    be = _prep(x2, w, r2)
    w = _c(w)
    y = torch.empty_like(x2)
    h = torch.empty_like(x2) if r2 is not None else x2
    rstd = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    be.lib.check(be.lib.tamd_rmsnorm_fwd(_p(x2), _p(r2), _p(w), _p(y), _p(h) if r2 is not None else None,
                                         _p(rstd), x2.shape[0], cols, float(eps), _code(x2), be.stream(x2)),
                 "tamd_rmsnorm_fwd")
    return y.view(x.shape), h.view(x.shape), rstd


def raw_rmsnorm_bwd(dy, h, w, rstd, dres=None):
    cols = h.shape[-1]
    dy2, h2 = _c(dy).view(-1, cols), _c(h).view(-1, cols)
    dr2 = None if dres is None else _c(dres).view(-1, cols)
    be = _prep(dy2, h2, w, dr2)
    w = _c(w)
    rows = h2.shape[0]
    dx = torch.empty_like(h2)
    dw = torch.empty_like(w)
    nbytes = be.lib.tamd_norm_bwd_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
    be.lib.check(be.lib.tamd_rmsnorm_bwd(_p(dy2), _p(h2), _p(w), _p(rstd), _p(dr2), _p(dx), _p(dw), _p(ws), nbytes,
                                         rows, cols, _code(h2), be.stream(h2)), "tamd_rmsnorm_bwd")
    return dx.view(h.shape), dw


def raw_layernorm_fwd(x, w, b, eps, residual=None):
    cols = x.shape[-1]
    x2 = _c(x).view(-1, cols)
    r2 = None if residual is None else _c(residual).view(-1, cols)
    be = _prep(x2, w, b, r2)
    w = _c(w)
    b = None if b is None else _c(b)
    y = torch.empty_like(x2)
    h = torch.empty_like(x2) if r2 is not None else x2
    mean = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    be.lib.check(be.lib.tamd_layernorm_fwd(_p(x2), _p(r2), _p(w), _p(b), _p(y),
                                           _p(h) if r2 is not None else None, _p(mean), _p(rstd), x2.shape[0], cols,
                                           float(eps), _code(x2), be.stream(x2)), "tamd_layernorm_fwd")
    return y.view(x.shape), h.view(x.shape), mean, rstd


def raw_layernorm_bwd(dy, h, w, mean, rstd, dres=None, need_db=True):
    cols = h.shape[-1]
    dy2, h2 = _c(dy).view(-1, cols), _c(h).view(-1, cols)
    dr2 = None if dres is None else _c(dres).view(-1, cols)
    be = _prep(dy2, h2, w, dr2)
    w = _c(w)
    rows = h2.shape[0]
    dx = torch.empty_like(h2)
    dw = torch.empty_like(w)
    db = torch.empty_like(w) if need_db else None
    nbytes = be.lib.tamd_norm_bwd_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
    be.lib.check(be.lib.tamd_layernorm_bwd(_p(dy2), _p(h2), _p(w), _p(mean), _p(rstd), _p(dr2), _p(dx), _p(dw),
                                           _p(db), _p(ws), nbytes, rows, cols, _code(h2), be.stream(h2)),
                 "tamd_layernorm_bwd")
    return dx.view(h.shape), dw, db


def raw_rope_(x2d, cos, sin, seq, nheads, head_dim, conj=False):
    """In-place rotary on the first `nheads` heads of every row of x2d [tokens, row_stride]."""
    be = _prep(x2d, cos, sin)
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    cos, sin = _c(cos), _c(sin)
    if cos.dtype != x2d.dtype:
        cos, sin = cos.to(x2d.dtype), sin.to(x2d.dtype)
    cos_batch = cos.shape[0] if cos.dim() == 3 else 1
    be.lib.check(be.lib.tamd_rope_inplace(_p(x2d), _p(cos), _p(sin), x2d.shape[0], seq, x2d.stride(0), nheads,
                                          head_dim, cos_batch, int(conj), _code(x2d), be.stream(x2d)),
                 "tamd_rope_inplace")
    return x2d


def raw_embedding_fwd(ids, table):
    be = _prep(ids, table)
    ids_c = _c(ids)
    if ids_c.dtype != torch.int64:
        ids_c = ids_c.long()
    table = _c(table)
    out = torch.empty(*ids.shape, table.shape[1], dtype=table.dtype, device=table.device)
    be.lib.check(be.lib.tamd_embedding_fwd(_p(ids_c), _p(table), _p(out), ids_c.numel(), table.shape[0],
                                           table.shape[1], None, _code(table), be.stream(table)),
                 "tamd_embedding_fwd")
    return out


def raw_embedding_bwd(ids, dout, vocab, padding_idx=-1):
    be = _prep(ids, dout)
    dim = dout.shape[-1]
    flat = _c(ids).view(-1).long()
    sorted_ids, perm = torch.sort(flat, stable=True)  # index plumbing on torch; accumulation is ours
    dtable = torch.zeros(vocab, dim, dtype=dout.dtype, device=dout.device)
    d2 = _c(dout).view(-1, dim)
    ws_bytes = 2 * (-(-max(flat.numel(), 1) // 32)) * dim * 4  # = tamd_embedding_bwd_workspace_bytes (32-token segments)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dout.device)
    be.lib.check(be.lib.tamd_embedding_bwd(_p(sorted_ids), _p(perm), _p(d2), _p(dtable), _p(ws), ws_bytes,
                                           flat.numel(), vocab, dim,
                                           -1 if padding_idx is None else int(padding_idx), _code(d2),
                                           be.stream(d2)), "tamd_embedding_bwd")
    return dtable


def raw_bert_embeddings_fwd(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, keep_pre_ln):
    be = _prep(input_ids, word, typ, pos, ln_w, ln_b)
    n = input_ids.numel()
    dim = word.shape[1]
    out = torch.empty(*input_ids.shape, dim, dtype=word.dtype, device=word.device)
    pre = torch.empty_like(out) if keep_pre_ln else None
    mean = torch.empty(n, dtype=torch.float32, device=word.device)
    rstd = torch.empty_like(mean)
    # named locals: the converted operands must outlive the launch call
    iid, tid, pid = _c(input_ids).long(), _c(token_type_ids).long(), _c(position_ids).long()
    word, typ, pos, ln_w, ln_b = _c(word), _c(typ), _c(pos), _c(ln_w), _c(ln_b)
    be.lib.check(be.lib.tamd_bert_embeddings_fwd(
        _p(iid), _p(tid), _p(pid), _p(word), _p(typ), _p(pos), _p(ln_w), _p(ln_b), _p(out), _p(pre), _p(mean),
        _p(rstd), n, dim,
        word.shape[0], typ.shape[0], pos.shape[0], float(eps), _code(word), be.stream(word)),
        "tamd_bert_embeddings_fwd")
    return out, pre, mean, rstd


def raw_swiglu_fwd(gu):
    """gu [T, 2I] = [gate | up]  ->  act [T, I]"""
    be = _prep(gu)
    t, two_i = gu.shape
    inter = two_i // 2
    act = torch.empty(t, inter, dtype=gu.dtype, device=gu.device)
    up = gu[:, inter:]
    be.lib.check(be.lib.tamd_swiglu_fwd(_p(gu), _p(up), _p(act), t, inter, gu.stride(0), act.stride(0), _code(gu),
                                        be.stream(gu)), "tamd_swiglu_fwd")
    return act


def raw_swiglu_bwd(gu, dact, want_act=False, inplace=False):
    be = _prep(gu, dact)
    t, two_i = gu.shape
    inter = two_i // 2
    dgu = gu if inplace else torch.empty_like(gu)
    act = torch.empty_like(dact) if want_act else None
    dact = _c(dact)
    be.lib.check(be.lib.tamd_swiglu_bwd(_p(gu), _p(gu[:, inter:]), _p(dact), _p(dgu), _p(dgu[:, inter:]), _p(act), t,
                                        inter, gu.stride(0), dact.stride(0), _code(gu), be.stream(gu)),
                 "tamd_swiglu_bwd")
    return dgu, act


def raw_bias_act_fwd(x, bias, act):
    x2 = _c(x).view(-1, x.shape[-1])
    be = _prep(x2, bias)
    y = torch.empty_like(x2)
    be.lib.check(be.lib.tamd_bias_act_fwd(_p(x2), _p(bias), _p(y), x2.shape[0], x2.shape[1], act, _code(x2),
                                          be.stream(x2)), "tamd_bias_act_fwd")
    return y.view(x.shape)


def raw_bias_act_bwd(x, bias, dy, act):
    x2, dy2 = _c(x).view(-1, x.shape[-1]), _c(dy).view(-1, x.shape[-1])
    be = _prep(x2, bias, dy2)
    dx = torch.empty_like(x2)
    be.lib.check(be.lib.tamd_bias_act_bwd(_p(x2), _p(bias), _p(dy2), _p(dx), x2.shape[0], x2.shape[1], act,
                                          _code(x2), be.stream(x2)), "tamd_bias_act_bwd")
    return dx.view(x.shape)


def raw_add(a, b):
    a, b = _c(a), _c(b)
    be = _prep(a, b)
    out = torch.empty_like(a)
    be.lib.check(be.lib.tamd_add(_p(a), _p(b), _p(out), a.numel(), _code(a), be.stream(a)), "tamd_add")
    return out


def raw_adamw_step_(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """In-place fused AdamW step on one tensor (p, m, v updated); torch.optim.AdamW semantics, include/tamd.h."""
    be = _prep(p, g, m, v)
    for t in (p, g, m, v):
        if not t.is_contiguous():
            raise TamdError("adamw_step needs contiguous tensors (parameters, gradients and moments)")
    if g.dtype != p.dtype or m.dtype != v.dtype or m.dtype not in (p.dtype, torch.float32):
        raise TamdError(f"adamw_step dtypes: p/g {p.dtype}/{g.dtype}, m/v {m.dtype}/{v.dtype}")
    be.lib.check(be.lib.tamd_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                        float(eps), float(weight_decay), int(step), float(grad_scale), _code(p),
                                        _code(m), be.stream(p)), "tamd_adamw_step")


def raw_colsum(x2d):
    be = _prep(x2d)
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=x2d.dtype, device=x2d.device)
    nbytes = be.lib.tamd_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x2d.device)
    be.lib.check(be.lib.tamd_colsum(_p(x2d), _p(out), _p(ws), nbytes, rows, cols, x2d.stride(0), _code(x2d),
                                    be.stream(x2d)), "tamd_colsum")
    return out


def raw_transpose(x2d):
    be = _prep(x2d)
    rows, cols = x2d.shape
    out = torch.empty(cols, rows, dtype=x2d.dtype, device=x2d.device)
    be.lib.check(be.lib.tamd_transpose(_p(x2d), _p(out), rows, cols, x2d.stride(0), out.stride(0), _code(x2d),
                                       be.stream(x2d)), "tamd_transpose")
    return out


def raw_cross_entropy_fwd(logits2d, labels, ignore_index=-100):
    be = _prep(logits2d, labels)
    t, v = logits2d.shape
    lse = torch.empty(t, dtype=torch.float32, device=logits2d.device)
    row_loss = torch.empty_like(lse)
    be.lib.check(be.lib.tamd_cross_entropy_fwd(_p(logits2d), _p(labels), _p(lse), _p(row_loss), t, v,
                                               logits2d.stride(0), ignore_index, _code(logits2d),
                                               be.stream(logits2d)), "tamd_cross_entropy_fwd")
    return lse, row_loss


def raw_cross_entropy_bwd(logits2d, labels, lse, gscale, ignore_index=-100):
    be = _prep(logits2d, labels, lse, gscale)
    t, v = logits2d.shape
    dlogits = torch.empty_like(logits2d)
    be.lib.check(be.lib.tamd_cross_entropy_bwd(_p(logits2d), _p(labels), _p(lse), _p(gscale), _p(dlogits), t, v,
                                               logits2d.stride(0), ignore_index, _code(logits2d),
                                               be.stream(logits2d)), "tamd_cross_entropy_bwd")
    return dlogits


def gemm_supported(m, n, k, dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and k % 8 == 0 and n % 8 == 0 and m > 0


def gemm_workspace_bytes(m: int, n: int, k: int, epilogue: int) -> int:
    """Host-side mirror of tamd_gemm_workspace_bytes (csrc/gemm.hip gemm_choose_splits): one ctypes round trip per
    GEMM is ~10 us, which small-model steps (hundreds of 30-us kernels) cannot hide.  tests/test_kernels.py keeps the
    two in step."""
    if k % 64 or n % 4 or epilogue not in (EPI_NONE, EPI_ACCUM):
        return 0
    tiles, nst = -(-m // 256) * -(-n // 256), k // 64
    if tiles > 128 or nst < 32:
        return 0
    s = min(256 // tiles, nst // 8, 16)
    if s < 2:
        return 0
    sps = -(-nst // s)
    return -(-nst // sps) * m * n * 4


GEMM_SCHED = {None: 0, "pp": 1 << 8, "fl": 3 << 8}  # diagnostic schedule hints (include/tamd.h)


def raw_gemm(a, b, *, a_km=False, b_kn=False, bias=None, residual=None, epilogue=EPI_NONE, act=ACT_NONE, out=None,
             sched=None):
    """C[M,N] = epi(A . B^T).  a: [M,K] (or [K,M] if a_km); b: [N,K] (or [K,N] if b_kn)."""
    be = _prep(a, b, bias, residual, out)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    (k_a, m) = a.shape if a_km else (a.shape[1], a.shape[0])
    (k_b, n) = b.shape if b_kn else (b.shape[1], b.shape[0])
    if k_a != k_b:
        raise TamdError(f"gemm K mismatch: {tuple(a.shape)} x {tuple(b.shape)} (a_km={a_km}, b_kn={b_kn})")
    if out is None:
        out = torch.empty(m, n, dtype=a.dtype, device=a.device)
    flags = (GEMM_A_KM if a_km else 0) | (GEMM_B_KN if b_kn else 0) | GEMM_SCHED[sched]
    ldr = residual.stride(0) if residual is not None else 0
    # split-K for tile grids that cannot fill the GPU (weight gradients of narrow layers): needs an fp32 workspace
    ws_bytes = gemm_workspace_bytes(m, n, k_a, epilogue) if sched is None else 0
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
        be.lib.check(be.lib.tamd_gemm_ws(_p(a), _p(b), _p(out), _p(bias), _p(residual), m, n, k_a, a.stride(0),
                                         b.stride(0), out.stride(0), ldr, flags, epilogue, act, _code(a), _p(ws),
                                         ws_bytes, be.stream(a)), "tamd_gemm_ws")
    else:
        be.lib.check(be.lib.tamd_gemm(_p(a), _p(b), _p(out), _p(bias), _p(residual), m, n, k_a, a.stride(0),
                                      b.stride(0), out.stride(0), ldr, flags, epilogue, act, _code(a), be.stream(a)),
                     "tamd_gemm")
    return out


def _attn_params(q, k, v, o, lse, key_valid, scale, causal, dropout_p=0.0, seed=0, q_start=None):
    """q/k/v/o are [B, S, H, D] *views* (any batch/seq/head strides, D contiguous)."""
    p = _cabi.AttnParams()
    p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    p.lse = lse.data_ptr() if lse is not None else None
    p.key_valid = key_valid.data_ptr() if key_valid is not None else None
    p.batch, p.seq_q, p.heads_q, p.head_dim = q.shape
    p.seq_k, p.heads_kv = k.shape[1], k.shape[2]
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t.stride(3) != 1:
            raise TamdError("attention operands must have a contiguous head_dim")
        setattr(p, name + "_stride_b", t.stride(0))
        setattr(p, name + "_stride_s", t.stride(1))
        setattr(p, name + "_stride_h", t.stride(2))
    p.scale = float(scale)
    p.causal = int(bool(causal))
    p.dtype = _code(q)
    p.dropout_p = float(dropout_p)
    p.dropout_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    p.q_start = q_start.data_ptr() if q_start is not None else None
    return p


def _check_q_start(q_start, q, causal):
    """packed sequences: int32 [2, B, S] = (first token of each query's sequence, last token of each key's sequence),
    include/tamd.h; build it with `packed_q_start`."""
    if q_start is None:
        return None
    if not causal:
        raise TamdError("packed sequences (q_start) need causal attention")
    if q_start.dtype != torch.int32 or tuple(q_start.shape) != (2, q.shape[0], q.shape[1]):
        raise TamdError(f"q_start must be int32 [2, batch, seq], got {q_start.dtype} {tuple(q_start.shape)}")
    return _c(q_start)


def raw_attn_fwd(q, k, v, scale, causal, key_valid=None, need_lse=True, out=None, dropout_p=0.0, seed=0,
                 q_start=None):
    """q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (strided views fine) -> o [B,Sq,Hq,D] contiguous, lse [B,Hq,Sq] fp32."""
    be = _prep(q, k, v, key_valid, out)
    b, sq, hq, d = q.shape
    o = out if out is not None else torch.empty(b, sq, hq, d, dtype=q.dtype, device=q.device)
    lse = torch.empty(b, hq, sq, dtype=torch.float32, device=q.device) if need_lse else None
    if key_valid is not None:
        key_valid = _c(key_valid.to(torch.uint8))
    q_start = _check_q_start(q_start, q, causal)
    p = _attn_params(q, k, v, o, lse, key_valid, scale, causal, dropout_p, seed, q_start)
    be.lib.check(be.lib.tamd_attn_fwd(ctypes.byref(p), be.stream(q)), "tamd_attn_fwd")
    return o, lse


def raw_attn_bwd(q, k, v, o, lse, dout, scale, causal, key_valid=None, dq=None, dk=None, dv=None,
                 dropout_p=0.0, seed=0, q_start=None):
    """Gradients written into dq/dk/dv (views with the strides of q/k/v) or freshly allocated."""
    be = _prep(q, k, v, o, lse, dout, key_valid)
    if dout.stride() != o.stride():
        dout = dout.contiguous() if o.is_contiguous() else dout.clone(memory_format=torch.preserve_format)
    if dq is None:
        dq = torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device)
    if dk is None:
        dk = torch.empty_strided(k.shape, k.stride(), dtype=k.dtype, device=k.device)
    if dv is None:
        dv = torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)
    assert dq.stride() == q.stride() and dk.stride() == k.stride() and dv.stride() == v.stride()
    if key_valid is not None:
        key_valid = _c(key_valid.to(torch.uint8))
    delta = torch.empty((2,) + tuple(lse.shape), dtype=torch.float32, device=lse.device)  # delta | lse*log2(e)
    bp = _cabi.AttnBwdParams()
    q_start = _check_q_start(q_start, q, causal)
    bp.fwd = _attn_params(q, k, v, o, lse, key_valid, scale, causal, dropout_p, seed, q_start)
    bp.dout, bp.dq, bp.dk, bp.dv, bp.delta = (dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                              delta.data_ptr())
    be.lib.check(be.lib.tamd_attn_bwd(ctypes.byref(bp), be.stream(q)), "tamd_attn_bwd")
    return dq, dk, dv


# --------------------------------------------------------------------------- autograd functions
class RMSNormFn(torch.autograd.Function):
    """y = LlamaRMSNorm(x).  Reference: models/llama/modeling_llama.py:62-67."""

    @staticmethod
    def forward(ctx, x, w, eps):
        y, h, rstd = raw_rmsnorm_fwd(x, w, eps, None)
        ctx.save_for_backward(h, w, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, rstd = ctx.saved_tensors
        dx, dw = raw_rmsnorm_bwd(dy, h, w, rstd)
        return dx, dw, None


class AddRMSNormFn(torch.autograd.Function):
    """h = x + residual; y = LlamaRMSNorm(h) -> (y, h): the residual add of LlamaDecoderLayer.forward
    (modeling_llama.py:317,323) fused into the norm that follows it."""

    @staticmethod
    def forward(ctx, x, residual, w, eps):
        y, h, rstd = raw_rmsnorm_fwd(x, w, eps, residual)
        ctx.save_for_backward(h, w, rstd)
        ctx.set_materialize_grads(False)
        return y, h

    @staticmethod
    def backward(ctx, dy, dh):
        h, w, rstd = ctx.saved_tensors
        if dy is None:
            return dh, dh, None, None
        dx, dw = raw_rmsnorm_bwd(dy, h, w, rstd, dres=dh)
        return dx, dx, dw, None


class LayerNormFn(torch.autograd.Function):
    """y = nn.LayerNorm(x).  Call sites: models/bert/modeling_bert.py:62,106; gpt2 :252-254; clip :358-360."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, h, mean, rstd = raw_layernorm_fwd(x, w, b, eps, None)
        ctx.save_for_backward(h, w, mean, rstd)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = raw_layernorm_bwd(dy, h, w, mean, rstd, need_db=ctx.has_b)
        return dx, dw, db, None


class AddLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + residual) -> (y, h): BertSelfOutput / BertOutput (modeling_bert.py:289-293, :347-351)."""

    @staticmethod
    def forward(ctx, x, residual, w, b, eps):
        y, h, mean, rstd = raw_layernorm_fwd(x, w, b, eps, residual)
        ctx.save_for_backward(h, w, mean, rstd)
        ctx.has_b = b is not None
        ctx.set_materialize_grads(False)
        return y, h

    @staticmethod
    def backward(ctx, dy, dh):
        h, w, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dh, dh, None, None, None
        dx, dw, db = raw_layernorm_bwd(dy, h, w, mean, rstd, dres=dh, need_db=ctx.has_b)
        return dx, dx, dw, db, None


def rmsnorm(x, w, eps, residual=None):
    """-> y  (or (y, h) with h = x + residual when a residual is given)."""
    if residual is None:
        return RMSNormFn.apply(x, w, eps)
    return AddRMSNormFn.apply(x, residual, w, eps)


def layernorm(x, w, b, eps, residual=None):
    if residual is None:
        return LayerNormFn.apply(x, w, b, eps)
    return AddLayerNormFn.apply(x, residual, w, b, eps)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) [+ residual] on the MFMA GEMM; dX and dW use the k-major operand modes
    (no HBM transposes).  nn.Linear call sites: see csrc/gemm.hip header."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, act):
        k = x.shape[-1]
        x2 = _c(x).view(-1, k)
        epi, r2 = EPI_NONE, None
        if residual is not None:
            epi, r2 = EPI_RESIDUAL, _c(residual).view(-1, w.shape[0])
        elif bias is not None and act != ACT_NONE:
            epi = EPI_BIAS_ACT
        elif bias is not None:
            epi = EPI_BIAS
        pre = None
        if act != ACT_NONE and (bias is None or residual is not None):
            raise TamdError("activation epilogue needs a bias and no residual")
        if epi == EPI_BIAS_ACT and any(ctx.needs_input_grad[:3]):
            # keep the pre-activation for the backward: GEMM+bias, then the activation kernel
            pre = raw_gemm(x2, w, bias=bias, epilogue=EPI_BIAS)
            y = raw_bias_act_fwd(pre, None, act)
        else:
            y = raw_gemm(x2, w, bias=bias, residual=r2, epilogue=epi, act=act)
        ctx.save_for_backward(x2, w, pre)
        ctx.has_bias, ctx.has_res, ctx.act = bias is not None, residual is not None, act
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        n = w.shape[0]
        dy2 = _c(dy).view(-1, n)
        dres = dy if ctx.has_res else None
        if pre is not None:
            dy2 = raw_bias_act_bwd(pre, None, dy2, ctx.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = raw_gemm(dy2, w, b_kn=True).view(ctx.x_shape)          # dX = dY . W
        if ctx.needs_input_grad[1]:
            dw = raw_gemm(dy2, x2, a_km=True, b_kn=True)                # dW = dY^T . X
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = raw_colsum(dy2)
        return dx, dw, db, dres, None


def linear(x, w, bias=None, residual=None, act=ACT_NONE):
    return LinearFn.apply(x, w, bias, residual, act)


class RopeFn(torch.autograd.Function):
    """Rotary embedding applied to the first `nheads` heads of a [B, S, row] projection output
    (models/llama/modeling_llama.py:130-160).  Out of place at this level (autograd needs the input intact)."""

    @staticmethod
    def forward(ctx, x, cos, sin, nheads, head_dim):
        b, s, row = x.shape
        y = x.clone()
        raw_rope_(y.view(b * s, row), cos, sin, s, nheads, head_dim, conj=False)
        ctx.save_for_backward(cos, sin)
        ctx.meta = (nheads, head_dim)
        return y

    @staticmethod
    def backward(ctx, dy):
        cos, sin = ctx.saved_tensors
        nheads, head_dim = ctx.meta
        b, s, row = dy.shape
        dx = dy.clone()
        raw_rope_(dx.view(b * s, row), cos, sin, s, nheads, head_dim, conj=True)
        return dx, None, None, None, None


class AttentionFn(torch.autograd.Function):
    """softmax(scale QK^T + mask) V on [B,S,H,D] views.  Reference: eager_attention_forward,
    models/llama/modeling_llama.py:191-213 and siblings."""

    @staticmethod
    def forward(ctx, q, k, v, key_valid, scale, causal, dropout_p=0.0, seed=0, q_start=None):
        need = any(ctx.needs_input_grad[:3])
        o, lse = raw_attn_fwd(q, k, v, scale, causal, key_valid, need_lse=need, dropout_p=dropout_p, seed=seed,
                              q_start=q_start)
        if need:
            ctx.save_for_backward(q, k, v, o, lse, key_valid, q_start)
        ctx.meta = (scale, causal, dropout_p, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, key_valid, q_start = ctx.saved_tensors
        scale, causal, dropout_p, seed = ctx.meta
        dq, dk, dv = raw_attn_bwd(q, k, v, o, lse, do, scale, causal, key_valid, dropout_p=dropout_p, seed=seed,
                                  q_start=q_start)
        return dq, dk, dv, None, None, None, None, None, None


def dropout_seed() -> int:
    """One 63-bit seed per attention call from torch's CPU generator: `torch.manual_seed` makes runs repeatable,
    and activation checkpointing (which restores the CPU RNG state before recomputing) regenerates the same mask."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def dropout_keep_mask(seed: int, batch: int, heads: int, seq_q: int, seq_k: int, p: float) -> torch.Tensor:
    """The kernels' keep mask [B,H,Sq,Sk] (bool), rebuilt on the host with the exported hash -- for tests/debugging."""
    import numpy as np

    idx = np.arange(batch * heads * seq_q * seq_k, dtype=np.uint64)
    lo, hi = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    slo, shi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = (lo ^ slo) * np.uint32(0x9E3779B1)
        x ^= x >> np.uint32(15)
        x += (hi * np.uint32(0x85EBCA77)) ^ shi
        x *= np.uint32(0xC2B2AE3D)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0x27D4EB2F)
        x ^= x >> np.uint32(16)
    thr = np.uint32(min(4294967295.0, float(np.float32(p)) * 4294967296.0))
    return torch.from_numpy((x >= thr).reshape(batch, heads, seq_q, seq_k))


def attention(q, k, v, scale, causal, key_valid=None, dropout_p=0.0, seed=None, q_start=None):
    if dropout_p > 0.0 and seed is None:
        seed = dropout_seed()
    return AttentionFn.apply(q, k, v, key_valid, scale, causal, float(dropout_p), int(seed or 0), q_start)


def packed_q_start(seq_ids: torch.Tensor) -> torch.Tensor:
    """[B, S] sequence ids of a packed batch (equal ids = same sequence, masking_utils.py:728-757) -> int32 [2, B, S]:
    plane 0 the index of the first token of each token's sequence, plane 1 the index of its last token (the two
    bounds the kernels take, include/tamd.h).  Device-side, no synchronisation."""
    s = seq_ids.shape[-1]
    pos = torch.arange(s, device=seq_ids.device).expand_as(seq_ids)
    first = torch.ones_like(seq_ids, dtype=torch.bool)
    first[..., 1:] = seq_ids[..., 1:] != seq_ids[..., :-1]
    last = torch.ones_like(first)
    last[..., :-1] = first[..., 1:]
    start = torch.where(first, pos, torch.zeros_like(pos)).cummax(-1).values
    end = torch.where(last, pos, torch.full_like(pos, s - 1)).flip(-1).cummin(-1).values.flip(-1)
    return torch.stack((start, end)).to(torch.int32).contiguous()


class SwiGLUFn(torch.autograd.Function):
    """act = silu(gate) * up on a fused [T, 2I] projection output (modeling_llama.py:174-176)."""

    @staticmethod
    def forward(ctx, gu):
        shape = gu.shape
        gu2 = _c(gu).view(-1, shape[-1])
        ctx.save_for_backward(gu2)
        ctx.shape = shape
        return raw_swiglu_fwd(gu2).view(*shape[:-1], shape[-1] // 2)

    @staticmethod
    def backward(ctx, dact):
        (gu2,) = ctx.saved_tensors
        dgu, _ = raw_swiglu_bwd(gu2, _c(dact).view(gu2.shape[0], -1))
        return dgu.view(ctx.shape)


class BiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, act):
        ctx.save_for_backward(x, bias)
        ctx.act = act
        return raw_bias_act_fwd(x, bias, act)

    @staticmethod
    def backward(ctx, dy):
        x, bias = ctx.saved_tensors
        dx = raw_bias_act_bwd(x, bias, dy, ctx.act)
        db = raw_colsum(dx.view(-1, dx.shape[-1])) if bias is not None else None
        return dx, db, None


class EmbeddingFn(torch.autograd.Function):
    """nn.Embedding (models/llama/modeling_llama.py:381): bit-exact gather, sorted scatter-add backward."""

    @staticmethod
    def forward(ctx, ids, table, padding_idx):
        ctx.save_for_backward(ids)
        ctx.meta = (table.shape[0], padding_idx)
        return raw_embedding_fwd(ids, table)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        vocab, padding_idx = ctx.meta
        return None, raw_embedding_bwd(ids, dout, vocab, padding_idx), None


def embedding(ids, table, padding_idx=None):
    return EmbeddingFn.apply(ids, table, padding_idx)


class CrossEntropyFn(torch.autograd.Function):
    """fixed_cross_entropy on `logits.float()` (loss/loss_utils.py:32-46) without materialising fp32 logits.
    Returns the SUM of per-token losses; the caller divides (mean over valid labels or num_items_in_batch)."""

    @staticmethod
    def forward(ctx, logits2d, labels, ignore_index):
        lse, row_loss = raw_cross_entropy_fwd(logits2d, labels, ignore_index)
        ctx.save_for_backward(logits2d, labels, lse)
        ctx.ignore_index = ignore_index
        return row_loss.sum()

    @staticmethod
    def backward(ctx, g):
        logits2d, labels, lse = ctx.saved_tensors
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        return raw_cross_entropy_bwd(logits2d, labels, lse, gs, ctx.ignore_index), None, None


def causal_lm_loss(logits, labels, vocab_size, num_items_in_batch=None, ignore_index=-100, shift_labels=None,
                   **_unused):
    """Drop-in for ForCausalLMLoss (loss/loss_utils.py:49-71): same shifting, same reductions."""
    if shift_labels is None:
        labels = torch.nn.functional.pad(labels, (0, 1), value=ignore_index)
        shift_labels = labels[..., 1:].contiguous()
    logits2d = logits.reshape(-1, vocab_size)
    shift_labels = shift_labels.reshape(-1).to(logits.device)
    total = CrossEntropyFn.apply(logits2d, shift_labels, ignore_index)
    if num_items_in_batch is not None:
        if torch.is_tensor(num_items_in_batch):
            num_items_in_batch = num_items_in_batch.to(total.device)
        return total / num_items_in_batch
    n_valid = (shift_labels != ignore_index).sum()
    return total / n_valid


class FusedLinearCrossEntropyFn(torch.autograd.Function):
    """lm_head + causal-LM loss without ever holding the [tokens, vocab] logits (reference:
    `logits = self.lm_head(hidden)` then ForCausalLMLoss, modeling_llama.py / loss/loss_utils.py:49-71).

    Tokens are processed in chunks: logits_c = h_c W^T (MFMA GEMM) -> cross-entropy forward (lse, per-token loss) ->
    dlogits_c, already scaled by 1/normaliser -> dh_c = dlogits_c W and dW += dlogits_c^T h_c (accumulate epilogue).
    The same three GEMMs as the unfused path, no recomputation; the gradients are produced in the forward and only
    multiplied by the upstream scalar in the backward.  Peak extra memory: one chunk of logits instead of 2 x [T, V].
    dW accumulates in the storage dtype across chunks (<= 8 roundings at the default chunking)."""

    @staticmethod
    def forward(ctx, h2d, w, labels, normaliser, ignore_index, chunk):
        t, v = h2d.shape[0], w.shape[0]
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gs = (1.0 / normaliser.to(torch.float32)).reshape(1).contiguous()
        loss = torch.zeros((), dtype=torch.float32, device=h2d.device)
        dh = torch.empty_like(h2d) if need_h else None
        dw = torch.empty_like(w) if need_w else None
        first = True
        for c0 in range(0, t, chunk):
            c1 = min(c0 + chunk, t)
            hc, lc = h2d[c0:c1], labels[c0:c1]
            logits = raw_gemm(hc, w)
            lse, row_loss = raw_cross_entropy_fwd(logits, lc, ignore_index)
            loss = loss + row_loss.sum()
            if need_h or need_w:
                dlog = raw_cross_entropy_bwd(logits, lc, lse, gs, ignore_index)
                del logits
                if need_h:
                    raw_gemm(dlog, w, b_kn=True, out=dh[c0:c1])
                if need_w:
                    raw_gemm(dlog, hc, a_km=True, b_kn=True, epilogue=EPI_NONE if first else EPI_ACCUM, out=dw)
                del dlog
            first = False
        ctx.save_for_backward(dh, dw)
        return loss * gs[0]

    @staticmethod
    def backward(ctx, g):
        dh, dw = ctx.saved_tensors
        g = g.detach()
        return (None if dh is None else dh * g.to(dh.dtype), None if dw is None else dw * g.to(dw.dtype), None, None,
                None, None)


def fused_linear_cross_entropy(hidden, weight, labels, num_items_in_batch=None, ignore_index=-100, shift=True,
                               chunk_tokens=None):
    """Causal-LM loss of `hidden @ weight.T` against `labels` (shifted by one like ForCausalLMLoss unless shift=False)
    without materialising the logits.  hidden [..., h], weight [V, h] (V % 8 == 0), labels [...] int64."""
    hd = hidden.shape[-1]
    if shift:
        labels = torch.nn.functional.pad(labels, (0, 1), value=ignore_index)[..., 1:]
    labels = labels.reshape(-1).to(hidden.device).contiguous()
    h2d = _c(hidden).view(-1, hd)
    t = h2d.shape[0]
    if chunk_tokens is None:  # at most 8 chunks, at least 2048 tokens each (multiple of 256 rows: whole GEMM tiles)
        chunk_tokens = max(2048, -(-t // 8))
        chunk_tokens = -(-chunk_tokens // 256) * 256
    if num_items_in_batch is not None:
        norm = torch.as_tensor(num_items_in_batch, device=hidden.device)
    else:
        norm = (labels != ignore_index).sum()
    return FusedLinearCrossEntropyFn.apply(h2d, weight, labels, norm, ignore_index, int(chunk_tokens))


# --------------------------------------------------------------------------- torch.ops registration
_LIB = torch.library.Library("tamd", "DEF")
_registered = False


def _register():
    """torch.ops.tamd.<name>: the raw kernels as dispatcher ops (CUDA key = HIP on ROCm) + Meta shapes."""
    global _registered
    if _registered:
        return
    _registered = True
    defs = [
        ("rmsnorm_fwd(Tensor x, Tensor w, float eps, Tensor? residual=None) -> (Tensor, Tensor, Tensor)",
         raw_rmsnorm_fwd,
         lambda x, w, eps, residual=None: (torch.empty_like(x), torch.empty_like(x),
                                           x.new_empty(x.numel() // x.shape[-1], dtype=torch.float32))),
        ("rmsnorm_bwd(Tensor dy, Tensor h, Tensor w, Tensor rstd, Tensor? dres=None) -> (Tensor, Tensor)",
         raw_rmsnorm_bwd, lambda dy, h, w, rstd, dres=None: (torch.empty_like(h), torch.empty_like(w))),
        ("layernorm_fwd(Tensor x, Tensor w, Tensor? b, float eps, Tensor? residual=None) -> "
         "(Tensor, Tensor, Tensor, Tensor)", raw_layernorm_fwd,
         lambda x, w, b, eps, residual=None: (torch.empty_like(x), torch.empty_like(x),
                                              x.new_empty(x.numel() // x.shape[-1], dtype=torch.float32),
                                              x.new_empty(x.numel() // x.shape[-1], dtype=torch.float32))),
        ("swiglu_fwd(Tensor gu) -> Tensor", raw_swiglu_fwd,
         lambda gu: gu.new_empty(gu.shape[0], gu.shape[1] // 2)),
        ("embedding_fwd(Tensor ids, Tensor table) -> Tensor", raw_embedding_fwd,
         lambda ids, table: table.new_empty(*ids.shape, table.shape[1])),
        ("gemm(Tensor a, Tensor b, bool a_km=False, bool b_kn=False, Tensor? bias=None, Tensor? residual=None, "
         "int epilogue=0, int act=0) -> Tensor",
         lambda a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0: raw_gemm(
             a, b, a_km=a_km, b_kn=b_kn, bias=bias, residual=residual, epilogue=epilogue, act=act),
         lambda a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0: a.new_empty(
             a.shape[1] if a_km else a.shape[0], b.shape[1] if b_kn else b.shape[0])),
        ("attn_fwd(Tensor q, Tensor k, Tensor v, float scale, bool causal, Tensor? key_valid=None) -> "
         "(Tensor, Tensor)",
         lambda q, k, v, scale, causal, key_valid=None: raw_attn_fwd(q, k, v, scale, causal, key_valid),
         lambda q, k, v, scale, causal, key_valid=None: (
             q.new_empty(q.shape), q.new_empty(q.shape[0], q.shape[2], q.shape[1], dtype=torch.float32))),
    ]
    for schema, impl, meta in defs:
        _LIB.define(schema)
        name = schema.split("(")[0]
        _LIB.impl(name, impl, "CUDA")
        _LIB.impl(name, meta, "Meta")


_register()
