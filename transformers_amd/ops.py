"""torch-level entry points of the MI355X kernels.

Three layers, all thin:

1. ``raw_*``   -- one Python function per C-ABI entry point (include/tamd.h): argument checks,
                 output allocation with torch's caching allocator, launch on the *current* HIP
                 stream of the calling thread.  No autograd.  These are the IMPLEMENTATIONS of ...
2. ``torch.ops.tamd.*`` -- ... the dispatcher ops (``torch.library``; the reference's own precedent is
                 src/transformers/integrations/moe.py:245-257): kernel-level ops (``gemm``, ``attn_fwd``,
                 ``rmsnorm_bwd`` ...) with fake (Meta) implementations, and differentiable ops (``linear``,
                 ``attention``, ``rmsnorm``, ``llama_layer`` ...) whose backward -- the contract of SURVEY.md
                 section 8a -- is attached with ``torch.library.register_autograd``.
3. wrappers    -- ``ops.linear(...)``, ``ops.attention(...)`` ...: Python conveniences (default arguments, the
                 "does anything need a gradient" flag) around ``torch.ops.tamd.*``.  The model code under
                 ``transformers_amd/models/`` uses only these and ``torch.ops.tamd.*``.

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
            return _RAW_STREAM(idx if idx is not None else _current_device())
        return torch.cuda.current_stream(t.device).cuda_stream


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_CUDA_OK = None  # torch.cuda.is_available(), asked once (it re-counts the devices on every call)


def _current_device() -> int:
    """torch.cuda.current_device() without its Python layers once CUDA is initialised (this runs several times per
    kernel launch; a small-model step is hundreds of 10-us launches and its host side is what bounds it)."""
    if _RAW_GET_DEVICE is not None and torch.cuda.is_initialized():
        return _RAW_GET_DEVICE()
    return torch.cuda.current_device()
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
    """Backend + operand checks: every operand on a GPU, all on ONE device; the launch runs with that device current
    (a tensor of another device than torch's current one -- device_map pipelines, several GPUs in one process --
    would otherwise be launched in the wrong device context)."""
    be = backend()
    dev = None
    for t in tensors:
        if t is not None:
            be.check_tensor(t)
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise TamdError(f"tamd op operands live on different devices: {dev} and {t.device}")
    if dev is not None and dev.type == "cuda" and dev.index != _current_device():
        torch.cuda.set_device(dev)  # restored by `_device_guard` around the raw_* call
    return be


def _device_guard(fn):
    """Run a raw_* launcher with the operands' device current, restoring torch's current device afterwards."""
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        global _CUDA_OK
        if _CUDA_OK is None:
            _CUDA_OK = torch.cuda.is_available()
        if not _CUDA_OK:
            return fn(*args, **kwargs)
        cur = _current_device()
        try:
            return fn(*args, **kwargs)
        finally:
            if _current_device() != cur:
                torch.cuda.set_device(cur)

    return guarded


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- raw ops
@_device_guard
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


@_device_guard
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


@_device_guard
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


@_device_guard
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


@_device_guard
def raw_layernorm_dropout_fwd(x, w, b, eps, residual, dropout_p, seed):
    """h = dropout(x, p) + residual; y = LayerNorm(h)  ->  (y, h, mean, rstd).  Keep mask: the counter-based hash of
    (seed, flat element index), `hidden_dropout_keep_mask` on the host."""
    cols = x.shape[-1]
    x2, r2 = _c(x).view(-1, cols), _c(residual).view(-1, cols)
    be = _prep(x2, w, b, r2)
    w = _c(w)
    b = None if b is None else _c(b)
    y, h = torch.empty_like(x2), torch.empty_like(x2)
    mean = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    be.lib.check(be.lib.tamd_layernorm_dropout_fwd(_p(x2), _p(r2), _p(w), _p(b), _p(y), _p(h), _p(mean), _p(rstd),
                                                   x2.shape[0], cols, float(eps), float(dropout_p),
                                                   int(seed) & 0xFFFFFFFFFFFFFFFF, _code(x2), be.stream(x2)),
                 "tamd_layernorm_dropout_fwd")
    return y.view(x.shape), h.view(x.shape), mean, rstd


@_device_guard
def raw_layernorm_dropout_bwd(dy, h, w, mean, rstd, dropout_p, seed, dres=None, need_db=True):
    """-> (dx = gradient of the residual input, dx_drop = gradient of the dropped-out input, dw, db)."""
    cols = h.shape[-1]
    dy2, h2 = _c(dy).view(-1, cols), _c(h).view(-1, cols)
    dr2 = None if dres is None else _c(dres).view(-1, cols)
    be = _prep(dy2, h2, w, dr2)
    w = _c(w)
    rows = h2.shape[0]
    dx, dxd = torch.empty_like(h2), torch.empty_like(h2)
    dw = torch.empty_like(w)
    db = torch.empty_like(w) if need_db else None
    nbytes = be.lib.tamd_norm_bwd_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=h.device)
    be.lib.check(be.lib.tamd_layernorm_dropout_bwd(_p(dy2), _p(h2), _p(w), _p(mean), _p(rstd), _p(dr2), _p(dx), _p(dxd),
                                                   _p(dw), _p(db), _p(ws), nbytes, rows, cols, float(dropout_p),
                                                   int(seed) & 0xFFFFFFFFFFFFFFFF, _code(h2), be.stream(h2)),
                 "tamd_layernorm_dropout_bwd")
    return dx.view(h.shape), dxd.view(h.shape), dw, db


@_device_guard
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


@_device_guard
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


@_device_guard
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


@_device_guard
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


@_device_guard
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


@_device_guard
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


@_device_guard
def raw_bias_act_fwd(x, bias, act):
    x2 = _c(x).view(-1, x.shape[-1])
    be = _prep(x2, bias)
    y = torch.empty_like(x2)
    be.lib.check(be.lib.tamd_bias_act_fwd(_p(x2), _p(bias), _p(y), x2.shape[0], x2.shape[1], act, _code(x2),
                                          be.stream(x2)), "tamd_bias_act_fwd")
    return y.view(x.shape)


@_device_guard
def raw_bias_act_bwd(x, bias, dy, act):
    x2, dy2 = _c(x).view(-1, x.shape[-1]), _c(dy).view(-1, x.shape[-1])
    be = _prep(x2, bias, dy2)
    dx = torch.empty_like(x2)
    be.lib.check(be.lib.tamd_bias_act_bwd(_p(x2), _p(bias), _p(dy2), _p(dx), x2.shape[0], x2.shape[1], act,
                                          _code(x2), be.stream(x2)), "tamd_bias_act_bwd")
    return dx.view(x.shape)


@_device_guard
def raw_add(a, b):
    a, b = _c(a), _c(b)
    be = _prep(a, b)
    out = torch.empty_like(a)
    be.lib.check(be.lib.tamd_add(_p(a), _p(b), _p(out), a.numel(), _code(a), be.stream(a)), "tamd_add")
    return out


@_device_guard
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


@_device_guard
def raw_colsum(x2d):
    be = _prep(x2d)
    rows, cols = x2d.shape
    out = torch.empty(cols, dtype=x2d.dtype, device=x2d.device)
    nbytes = be.lib.tamd_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x2d.device)
    be.lib.check(be.lib.tamd_colsum(_p(x2d), _p(out), _p(ws), nbytes, rows, cols, x2d.stride(0), _code(x2d),
                                    be.stream(x2d)), "tamd_colsum")
    return out


@_device_guard
def raw_transpose(x2d):
    be = _prep(x2d)
    rows, cols = x2d.shape
    out = torch.empty(cols, rows, dtype=x2d.dtype, device=x2d.device)
    be.lib.check(be.lib.tamd_transpose(_p(x2d), _p(out), rows, cols, x2d.stride(0), out.stride(0), _code(x2d),
                                       be.stream(x2d)), "tamd_transpose")
    return out


@_device_guard
def raw_cross_entropy_fwd(logits2d, labels, ignore_index=-100):
    be = _prep(logits2d, labels)
    t, v = logits2d.shape
    lse = torch.empty(t, dtype=torch.float32, device=logits2d.device)
    row_loss = torch.empty_like(lse)
    be.lib.check(be.lib.tamd_cross_entropy_fwd(_p(logits2d), _p(labels), _p(lse), _p(row_loss), t, v,
                                               logits2d.stride(0), ignore_index, _code(logits2d),
                                               be.stream(logits2d)), "tamd_cross_entropy_fwd")
    return lse, row_loss


@_device_guard
def raw_cross_entropy_bwd(logits2d, labels, lse, gscale, ignore_index=-100, padded=False):
    """-> dlogits [t, v], a view of a fresh [t, ld] buffer (ld = the row stride of logits2d) whose padding columns
    v .. ld-1 the kernel zeroes; `padded=True` returns that whole buffer (a K-padded GEMM operand)."""
    be = _prep(logits2d, labels, lse, gscale)
    t, v = logits2d.shape
    ld = logits2d.stride(0)
    if logits2d.stride(1) != 1 or ld < v:
        raise TamdError("cross_entropy_bwd needs row-major logits")
    buf = torch.empty(t, ld, dtype=logits2d.dtype, device=logits2d.device)
    be.lib.check(be.lib.tamd_cross_entropy_bwd(_p(logits2d), _p(labels), _p(lse), _p(gscale), _p(buf), t, v,
                                               ld, ignore_index, _code(logits2d),
                                               be.stream(logits2d)), "tamd_cross_entropy_bwd")
    return buf if (padded or ld == v) else buf[:, :v]


def gemm_supported(m, n, k, dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and k % 8 == 0 and n % 8 == 0 and m > 0


def gemm_workspace_bytes(m: int, n: int, k: int, epilogue: int) -> int:
    """Host-side mirror of tamd_gemm_workspace_bytes (csrc/gemm.hip gemm_choose_splits): one ctypes round trip per
    GEMM is ~10 us, which small-model steps (hundreds of 30-us kernels) cannot hide.  tests/test_kernels.py keeps the
    two in step."""
    if k % 64 or n % 4 or epilogue not in (EPI_NONE, EPI_ACCUM):
        return 0
    tiles, nst = -(-m // 256) * -(-n // 256), k // 64
    if nst < 32:
        return 0
    s = 1
    if tiles <= 128:
        s = min(256 // tiles, nst // 8, 16)
    elif tiles < 2048 and nst >= 128:  # a few rounds with a mostly empty last one: see gemm_choose_splits
        best = 0
        for c in range(1, 5):
            cost = -(-tiles * c // 256) * nst * 1400 // c + (c * m * n // 500 if c > 1 else 0)
            if c == 1 or cost * 100 < best * 97:
                best, s = cost, c
    if s < 2:
        return 0
    sps = -(-nst // s)
    return -(-nst // sps) * m * n * 4


GEMM_SCHED = {None: 0, "pp": 1 << 8, "fl": 3 << 8, "fl_persist": 4 << 8, "fl_persist_sync": 5 << 8}  # include/tamd.h


@_device_guard
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
    # a residual epilogue on a tile grid that cannot fill the GPU (o_proj / down_proj of a short prompt: 80 tiles on 256
    # CUs): the residual goes into C first and the product is accumulated onto it, which split-K can do -- the same
    # roundings, round(round(acc) + R), for the price of copying a small C
    ws_bytes = 0
    if residual is not None:
        # the kernel reads R as the output's element type through 16-byte accesses: checked here because the rewrite
        # below takes R out of the C call (whose own checks would otherwise catch a stray dtype / stride / alignment)
        if residual.dtype != out.dtype:
            raise TamdError(f"gemm residual dtype {residual.dtype} != output dtype {out.dtype}")
        if residual.dim() != 2 or residual.stride(1) != 1 or residual.stride(0) % 8 or residual.data_ptr() % 16:
            raise TamdError("gemm residual must be a 2-D row-major view with a 16-byte aligned base and a row stride "
                            "that is a multiple of 8 elements")
    if sched is None:  # (one evaluation of the split-K policy per GEMM: this path is host-bound for small models)
        if epilogue == EPI_RESIDUAL and bias is None and residual is not None and residual.shape == out.shape:
            ws_bytes = gemm_workspace_bytes(m, n, k_a, EPI_ACCUM)
            if ws_bytes:
                out.copy_(residual)
                residual, epilogue = None, EPI_ACCUM
        else:
            ws_bytes = gemm_workspace_bytes(m, n, k_a, epilogue)
    ldr = residual.stride(0) if residual is not None else 0
    # split-K for tile grids that cannot fill the GPU (weight gradients of narrow layers): needs an fp32 workspace
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


def attn_bwd_rope_supported(q, k, cos, head_dim) -> bool:
    """The attention backward can apply the transposed rotary embedding to dq / dk on their way out."""
    return head_dim == 128 and q.shape[1] == k.shape[1] and cos.shape[-1] == 128 and cos.dim() in (2, 3)


def gemm_rope_supported(x2, wqkv, cos, head_dim) -> bool:
    """Shapes the q|k|v GEMM with the rotary epilogue takes (csrc/gemm.hip tamd_gemm_rope): heads of 128; a cos / sin
    table shared by the batch ([seq, 128]) needs seq >= 128."""
    n, k = wqkv.shape
    return (head_dim == 128 and (cos.dim() == 3 and cos.shape[0] > 1 or cos.shape[-2] >= 128) and x2.dtype in (torch.bfloat16, torch.float16) and wqkv.dtype == x2.dtype and k % 64 == 0
            and n % 128 == 0 and x2.stride(1) == 1 and wqkv.stride(1) == 1 and x2.stride(0) % 8 == 0
            and wqkv.stride(0) % 8 == 0 and cos.shape[-1] == 128
            # a tile grid that cannot fill the GPU (a short prompt) is better off with split-K and the rotary kernel
            and gemm_workspace_bytes(x2.shape[0], n, k, EPI_NONE) == 0)


@_device_guard
def raw_gemm_rope(x2, wqkv, cos, sin, seq, rope_heads, head_dim):
    """qkv [T, N] = x2 [T, K] . wqkv [N, K]^T with apply_rotary_pos_emb on the first rope_heads heads (query + key) in
    the GEMM epilogue; bit-identical to raw_gemm followed by raw_rope_."""
    be = _prep(x2, wqkv, cos, sin)
    cos, sin = _c(cos), _c(sin)
    if cos.dtype != x2.dtype:
        cos, sin = cos.to(x2.dtype), sin.to(x2.dtype)
    cos_batch = cos.shape[0] if cos.dim() == 3 else 1
    t, k = x2.shape
    n = wqkv.shape[0]
    out = torch.empty(t, n, dtype=x2.dtype, device=x2.device)
    be.lib.check(be.lib.tamd_gemm_rope(_p(x2), _p(wqkv), _p(out), _p(cos), _p(sin), t, n, k, x2.stride(0), wqkv.stride(0),
                                       out.stride(0), seq, cos_batch, rope_heads * head_dim, _code(x2), be.stream(x2)),
                 "tamd_gemm_rope")
    return out


def gemm_swiglu_supported(x2, wgu) -> bool:
    """Shapes the fused gate|up GEMM + SiLU*up epilogue takes (csrc/gemm.hip tamd_gemm_swiglu)."""
    two_i, k = wgu.shape
    return (x2.dtype in (torch.bfloat16, torch.float16) and wgu.dtype == x2.dtype and k % 64 == 0 and two_i % 16 == 0
            and x2.stride(1) == 1 and wgu.stride(1) == 1 and x2.stride(0) % 8 == 0 and wgu.stride(0) % 8 == 0
            and two_i * wgu.stride(0) * 2 < 2 ** 31
            and gemm_workspace_bytes(x2.shape[0], two_i, k, EPI_NONE) == 0)  # (small grids: split-K + swiglu kernel)


@_device_guard
def raw_gemm_swiglu(x2, wgu, need_gu=True):
    """x2 [T, K], wgu [2I, K] = [gate_proj.weight ; up_proj.weight]  ->  (gu [T, 2I] or None, act [T, I])."""
    be = _prep(x2, wgu)
    t, k = x2.shape
    inter = wgu.shape[0] // 2
    gu = torch.empty(t, 2 * inter, dtype=x2.dtype, device=x2.device) if need_gu else None
    act = torch.empty(t, inter, dtype=x2.dtype, device=x2.device)
    be.lib.check(be.lib.tamd_gemm_swiglu(_p(x2), _p(wgu), _p(gu), _p(act), t, inter, k, x2.stride(0), wgu.stride(0),
                                         2 * inter, inter, _code(x2), be.stream(x2)), "tamd_gemm_swiglu")
    return gu, act


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


def _check_key_valid(key_valid, q, k):
    """[batch, seq_k] key-validity plane (1 = attend): the kernels index it as key_valid[b * seq_k + key]."""
    if key_valid is None:
        return None
    if tuple(key_valid.shape) != (q.shape[0], k.shape[1]):
        raise TamdError(f"key_valid must be [batch, seq_k] = {(q.shape[0], k.shape[1])}, got {tuple(key_valid.shape)}")
    return _c(key_valid.to(torch.uint8))


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


@_device_guard
def raw_attn_fwd(q, k, v, scale, causal, key_valid=None, need_lse=True, out=None, dropout_p=0.0, seed=0,
                 q_start=None):
    """q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (strided views fine) -> o [B,Sq,Hq,D] contiguous, lse [B,Hq,Sq] fp32."""
    be = _prep(q, k, v, key_valid, out)
    b, sq, hq, d = q.shape
    o = out if out is not None else torch.empty(b, sq, hq, d, dtype=q.dtype, device=q.device)
    lse = torch.empty(b, hq, sq, dtype=torch.float32, device=q.device) if need_lse else None
    key_valid = _check_key_valid(key_valid, q, k)
    q_start = _check_q_start(q_start, q, causal)
    p = _attn_params(q, k, v, o, lse, key_valid, scale, causal, dropout_p, seed, q_start)
    be.lib.check(be.lib.tamd_attn_fwd(ctypes.byref(p), be.stream(q)), "tamd_attn_fwd")
    return o, lse


@_device_guard
def raw_attn_bwd(q, k, v, o, lse, dout, scale, causal, key_valid=None, dq=None, dk=None, dv=None,
                 dropout_p=0.0, seed=0, q_start=None, rope=None):
    """Gradients written into dq/dk/dv (views with the strides of q/k/v) or freshly allocated.  rope = (cos, sin)
    ([seq, 128] or [batch, seq, 128], the storage dtype): q and k had been rotated before the attention, dq and dk leave
    through the transposed rotation (the same bits as raw_rope_(conj=True) on the stored gradients)."""
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
    key_valid = _check_key_valid(key_valid, q, k)
    delta = torch.empty((2,) + tuple(lse.shape), dtype=torch.float32, device=lse.device)  # delta | lse*log2(e)
    bp = _cabi.AttnBwdParams()
    q_start = _check_q_start(q_start, q, causal)
    bp.fwd = _attn_params(q, k, v, o, lse, key_valid, scale, causal, dropout_p, seed, q_start)
    bp.dout, bp.dq, bp.dk, bp.dv, bp.delta = (dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                              delta.data_ptr())
    if rope is not None:
        cos, sin = (_c(t) if t.dtype == q.dtype else t.to(q.dtype).contiguous() for t in rope)
        be.check_tensor(cos), be.check_tensor(sin)
        bp.rope_cos, bp.rope_sin = cos.data_ptr(), sin.data_ptr()
        bp.rope_cos_batch = cos.shape[0] if cos.dim() == 3 else 1
    be.lib.check(be.lib.tamd_attn_bwd(ctypes.byref(bp), be.stream(q)), "tamd_attn_bwd")
    return dq, dk, dv


# --------------------------------------------------------------------------- torch.ops.tamd.*
# Every kernel entry point is a dispatcher op (`torch.ops.tamd.<name>`, CUDA key = HIP on ROCm) with a fake (Meta)
# implementation; the differentiable ops additionally carry their backward through `torch.library.register_autograd`
# -- the reference's own precedent for custom ops: src/transformers/integrations/moe.py:245-257.  The model code in
# `transformers_amd/models/` only ever calls `torch.ops.tamd.*` (through the thin wrappers at the end of this file).
# The "CPU" registrations are the same functions: with the product backend a CPU tensor raises TamdError in `_prep`
# (there is no CPU implementation); tests/hipemu swaps in the CPU execution model of the kernels.
_LIB = torch.library.Library("tamd", "DEF")
T = torch.ops.tamd  # op namespace (attributes resolve lazily, after the definitions below)


def define_op(schema: str, impl, fake, backward=None, setup_context=None):
    """Define `tamd::<name>`: schema + CUDA/CPU implementation + fake (Meta) + optional autograd formula."""
    name = schema.split("(", 1)[0].strip()
    _LIB.define(schema)
    _LIB.impl(name, impl, "CUDA")
    _LIB.impl(name, impl, "CPU")
    torch.library.register_fake(f"tamd::{name}", fake, lib=_LIB)
    if backward is not None:
        torch.library.register_autograd(f"tamd::{name}", backward, setup_context=setup_context, lib=_LIB)
    return name


def _rows(x):
    return x.numel() // x.shape[-1]


def _f32(x, *shape):
    return x.new_empty(shape, dtype=torch.float32)


def _nothing(x):
    """Placeholder for an absent tensor in an op's return tuple (op schemas cannot return `Tensor?`)."""
    return x.new_empty(0)


# ---- kernel-level ops (one per C-ABI entry point, no autograd) ---------------------------------------------
def _rmsnorm_fwd_impl(x, w, eps, residual=None):
    y, h, rstd = raw_rmsnorm_fwd(x, w, eps, residual)
    return y, (h if residual is not None else _nothing(x)), rstd  # (an op output must not alias an input)


define_op("rmsnorm_fwd(Tensor x, Tensor w, float eps, Tensor? residual=None) -> (Tensor, Tensor, Tensor)",
          _rmsnorm_fwd_impl,
          lambda x, w, eps, residual=None: (torch.empty_like(x),
                                            torch.empty_like(x) if residual is not None else _nothing(x),
                                            _f32(x, _rows(x))))
define_op("rmsnorm_bwd(Tensor dy, Tensor h, Tensor w, Tensor rstd, Tensor? dres=None) -> (Tensor, Tensor)",
          raw_rmsnorm_bwd, lambda dy, h, w, rstd, dres=None: (torch.empty_like(h), torch.empty_like(w)))


def _layernorm_fwd_impl(x, w, b, eps, residual=None):
    y, h, mean, rstd = raw_layernorm_fwd(x, w, b, eps, residual)
    return y, (h if residual is not None else _nothing(x)), mean, rstd


define_op("layernorm_fwd(Tensor x, Tensor w, Tensor? b, float eps, Tensor? residual=None) -> "
          "(Tensor, Tensor, Tensor, Tensor)", _layernorm_fwd_impl,
          lambda x, w, b, eps, residual=None: (torch.empty_like(x),
                                               torch.empty_like(x) if residual is not None else _nothing(x),
                                               _f32(x, _rows(x)), _f32(x, _rows(x))))


def _layernorm_bwd_impl(dy, h, w, mean, rstd, dres=None, need_db=True):
    dx, dw, db = raw_layernorm_bwd(dy, h, w, mean, rstd, dres, need_db)
    return dx, dw, db if db is not None else _nothing(w)


define_op("layernorm_bwd(Tensor dy, Tensor h, Tensor w, Tensor mean, Tensor rstd, Tensor? dres=None, "
          "bool need_db=True) -> (Tensor, Tensor, Tensor)", _layernorm_bwd_impl,
          lambda dy, h, w, mean, rstd, dres=None, need_db=True: (torch.empty_like(h), torch.empty_like(w),
                                                                 torch.empty_like(w) if need_db else _nothing(w)))


define_op("layernorm_dropout_fwd(Tensor x, Tensor w, Tensor? b, float eps, Tensor residual, float dropout_p, int seed) "
          "-> (Tensor, Tensor, Tensor, Tensor)", raw_layernorm_dropout_fwd,
          lambda x, w, b, eps, residual, dropout_p, seed: (torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x)),
                                                           _f32(x, _rows(x))))


def _layernorm_dropout_bwd_impl(dy, h, w, mean, rstd, dropout_p, seed, dres=None, need_db=True):
    dx, dxd, dw, db = raw_layernorm_dropout_bwd(dy, h, w, mean, rstd, dropout_p, seed, dres, need_db)
    return dx, dxd, dw, db if db is not None else _nothing(w)


define_op("layernorm_dropout_bwd(Tensor dy, Tensor h, Tensor w, Tensor mean, Tensor rstd, float dropout_p, int seed, "
          "Tensor? dres=None, bool need_db=True) -> (Tensor, Tensor, Tensor, Tensor)", _layernorm_dropout_bwd_impl,
          lambda dy, h, w, mean, rstd, dropout_p, seed, dres=None, need_db=True: (
              torch.empty_like(h), torch.empty_like(h), torch.empty_like(w),
              torch.empty_like(w) if need_db else _nothing(w)))


def _rope_impl(x2d, cos, sin, seq, nheads, head_dim, conj=False):
    raw_rope_(x2d, cos, sin, seq, nheads, head_dim, conj)


define_op("rope_(Tensor(a!) x2d, Tensor cos, Tensor sin, int seq, int nheads, int head_dim, bool conj=False) -> ()",
          _rope_impl, lambda x2d, cos, sin, seq, nheads, head_dim, conj=False: None)
define_op("embedding_fwd(Tensor ids, Tensor table) -> Tensor", raw_embedding_fwd,
          lambda ids, table: table.new_empty(*ids.shape, table.shape[1]))
define_op("embedding_bwd(Tensor ids, Tensor dout, int vocab, int padding_idx=-1) -> Tensor", raw_embedding_bwd,
          lambda ids, dout, vocab, padding_idx=-1: dout.new_empty(vocab, dout.shape[-1]))


def _bert_embeddings_fwd_impl(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, keep_pre_ln):
    out, pre, mean, rstd = raw_bert_embeddings_fwd(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w,
                                                   ln_b, eps, keep_pre_ln)
    return out, pre if pre is not None else _nothing(out), mean, rstd


define_op("bert_embeddings_fwd(Tensor input_ids, Tensor token_type_ids, Tensor position_ids, Tensor word, Tensor typ, "
          "Tensor pos, Tensor ln_w, Tensor ln_b, float eps, bool keep_pre_ln) -> (Tensor, Tensor, Tensor, Tensor)",
          _bert_embeddings_fwd_impl,
          lambda input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, keep_pre_ln: (
              word.new_empty(*input_ids.shape, word.shape[1]),
              word.new_empty(*input_ids.shape, word.shape[1]) if keep_pre_ln else _nothing(word),
              _f32(word, input_ids.numel()), _f32(word, input_ids.numel())))
define_op("swiglu_fwd(Tensor gu) -> Tensor", raw_swiglu_fwd, lambda gu: gu.new_empty(gu.shape[0], gu.shape[1] // 2))


def _swiglu_bwd_impl(gu, dact, want_act=False):
    dgu, act = raw_swiglu_bwd(gu, dact, want_act=want_act)
    return dgu, act if act is not None else _nothing(gu)


define_op("swiglu_bwd(Tensor gu, Tensor dact, bool want_act=False) -> (Tensor, Tensor)", _swiglu_bwd_impl,
          lambda gu, dact, want_act=False: (torch.empty_like(gu), torch.empty_like(dact) if want_act else _nothing(gu)))
define_op("bias_act_fwd(Tensor x, Tensor? bias, int act) -> Tensor", raw_bias_act_fwd,
          lambda x, bias, act: torch.empty_like(x))
define_op("bias_act_bwd(Tensor x, Tensor? bias, Tensor dy, int act) -> Tensor", raw_bias_act_bwd,
          lambda x, bias, dy, act: torch.empty_like(x))
define_op("add(Tensor a, Tensor b) -> Tensor", raw_add, lambda a, b: torch.empty_like(a))
define_op("colsum(Tensor x2d) -> Tensor", raw_colsum, lambda x2d: x2d.new_empty(x2d.shape[1]))
define_op("transpose(Tensor x2d) -> Tensor", raw_transpose, lambda x2d: x2d.new_empty(x2d.shape[1], x2d.shape[0]))
define_op("cross_entropy_fwd(Tensor logits2d, Tensor labels, int ignore_index=-100) -> (Tensor, Tensor)",
          raw_cross_entropy_fwd,
          lambda logits2d, labels, ignore_index=-100: (_f32(logits2d, logits2d.shape[0]),
                                                       _f32(logits2d, logits2d.shape[0])))
define_op("cross_entropy_bwd(Tensor logits2d, Tensor labels, Tensor lse, Tensor gscale, int ignore_index=-100) -> Tensor",
          lambda logits2d, labels, lse, gscale, ignore_index=-100: raw_cross_entropy_bwd(logits2d, labels, lse, gscale,
                                                                                         ignore_index),
          lambda logits2d, labels, lse, gscale, ignore_index=-100: torch.empty_strided(
              logits2d.shape, (logits2d.stride(0), 1), dtype=logits2d.dtype, device=logits2d.device))


def _adamw_impl(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    raw_adamw_step_(p, g, m, v, lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, step=step,
                    grad_scale=grad_scale)


define_op("adamw_step_(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, "
          "float eps, float weight_decay, int step, float grad_scale=1.0) -> ()", _adamw_impl,
          lambda p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0: None)


def _gemm_shape(a, b, a_km, b_kn):
    return (a.shape[1] if a_km else a.shape[0]), (b.shape[1] if b_kn else b.shape[0])


def _gemm_impl(a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0):
    return raw_gemm(a, b, a_km=a_km, b_kn=b_kn, bias=bias, residual=residual, epilogue=epilogue, act=act)


def _gemm_out_impl(out, a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0):
    raw_gemm(a, b, a_km=a_km, b_kn=b_kn, bias=bias, residual=residual, epilogue=epilogue, act=act, out=out)


define_op("gemm(Tensor a, Tensor b, bool a_km=False, bool b_kn=False, Tensor? bias=None, Tensor? residual=None, "
          "int epilogue=0, int act=0) -> Tensor", _gemm_impl,
          lambda a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0: a.new_empty(
              *_gemm_shape(a, b, a_km, b_kn)))
define_op("gemm_out(Tensor(a!) out, Tensor a, Tensor b, bool a_km=False, bool b_kn=False, Tensor? bias=None, "
          "Tensor? residual=None, int epilogue=0, int act=0) -> ()", _gemm_out_impl,
          lambda out, a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0: None)


def _gemm_swiglu_impl(x2, wgu, need_gu=True):
    gu, act = raw_gemm_swiglu(x2, wgu, need_gu)
    return gu if gu is not None else _nothing(x2), act


define_op("gemm_swiglu(Tensor x2, Tensor wgu, bool need_gu=True) -> (Tensor, Tensor)", _gemm_swiglu_impl,
          lambda x2, wgu, need_gu=True: (x2.new_empty(x2.shape[0], wgu.shape[0]) if need_gu else _nothing(x2),
                                         x2.new_empty(x2.shape[0], wgu.shape[0] // 2)))


def _attn_fwd_impl(q, k, v, scale, causal, key_valid=None, need_lse=True, dropout_p=0.0, seed=0, q_start=None):
    o, lse = raw_attn_fwd(q, k, v, scale, causal, key_valid, need_lse=need_lse, dropout_p=dropout_p, seed=seed,
                          q_start=q_start)
    return o, lse if lse is not None else _nothing(q)


define_op("attn_fwd(Tensor q, Tensor k, Tensor v, float scale, bool causal, Tensor? key_valid=None, "
          "bool need_lse=True, float dropout_p=0.0, int seed=0, Tensor? q_start=None) -> (Tensor, Tensor)",
          _attn_fwd_impl,
          lambda q, k, v, scale, causal, key_valid=None, need_lse=True, dropout_p=0.0, seed=0, q_start=None: (
              q.new_empty(q.shape), _f32(q, q.shape[0], q.shape[2], q.shape[1]) if need_lse else _nothing(q)))


def _attn_bwd_impl(q, k, v, o, lse, dout, scale, causal, key_valid=None, dropout_p=0.0, seed=0, q_start=None):
    return raw_attn_bwd(q, k, v, o, lse, dout, scale, causal, key_valid, dropout_p=dropout_p, seed=seed,
                        q_start=q_start)


define_op("attn_bwd(Tensor q, Tensor k, Tensor v, Tensor o, Tensor lse, Tensor dout, float scale, bool causal, "
          "Tensor? key_valid=None, float dropout_p=0.0, int seed=0, Tensor? q_start=None) -> (Tensor, Tensor, Tensor)",
          _attn_bwd_impl,
          lambda q, k, v, o, lse, dout, scale, causal, key_valid=None, dropout_p=0.0, seed=0, q_start=None: (
              torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device),
              torch.empty_strided(k.shape, k.stride(), dtype=k.dtype, device=k.device),
              torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)))


# ---- differentiable ops --------------------------------------------------------------------------------------
# Convention: forward ops return what the backward needs as extra outputs (saved in `setup_context`); auxiliary
# outputs never receive gradients (`set_materialize_grads(False)` -> None).  A `train` flag tells a forward op whether
# anything will be differentiated (it runs below autograd and cannot see `requires_grad`).
def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# RMSNorm -- LlamaRMSNorm.forward, models/llama/modeling_llama.py:62-67
def _rmsnorm_impl(x, w, eps):
    y, _, rstd = raw_rmsnorm_fwd(x, w, eps, None)
    return y, rstd


def _rmsnorm_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])
    ctx.set_materialize_grads(False)


def _rmsnorm_backward(ctx, dy, _drstd):
    if dy is None:
        return None, None, None
    x, w, rstd = ctx.saved_tensors
    dx, dw = T.rmsnorm_bwd(dy, x, w, rstd)
    return dx, dw, None


define_op("rmsnorm(Tensor x, Tensor w, float eps) -> (Tensor, Tensor)", _rmsnorm_impl,
          lambda x, w, eps: (torch.empty_like(x), _f32(x, _rows(x))), _rmsnorm_backward, _rmsnorm_setup)


# h = x + residual; y = RMSNorm(h): the residual add of LlamaDecoderLayer.forward (modeling_llama.py:317,323)
def _add_rmsnorm_impl(x, residual, w, eps):
    return raw_rmsnorm_fwd(x, w, eps, residual)


def _add_rmsnorm_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[2], output[2])
    ctx.set_materialize_grads(False)


def _add_rmsnorm_backward(ctx, dy, dh, _drstd):
    if dy is None:
        return dh, dh, None, None
    h, w, rstd = ctx.saved_tensors
    dx, dw = T.rmsnorm_bwd(dy, h, w, rstd, dh)
    return dx, dx, dw, None


define_op("add_rmsnorm(Tensor x, Tensor residual, Tensor w, float eps) -> (Tensor, Tensor, Tensor)",
          _add_rmsnorm_impl, lambda x, residual, w, eps: (torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x))),
          _add_rmsnorm_backward, _add_rmsnorm_setup)


# LayerNorm -- call sites models/bert/modeling_bert.py:62,106; gpt2 :252-254; clip :358-360
def _layernorm_impl(x, w, b, eps):
    y, _, mean, rstd = raw_layernorm_fwd(x, w, b, eps, None)
    return y, mean, rstd


def _layernorm_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1], output[2])
    ctx.has_b = inputs[2] is not None
    ctx.set_materialize_grads(False)


def _layernorm_backward(ctx, dy, _dm, _dr):
    if dy is None:
        return None, None, None, None
    x, w, mean, rstd = ctx.saved_tensors
    dx, dw, db = T.layernorm_bwd(dy, x, w, mean, rstd, None, ctx.has_b)
    return dx, dw, (db if ctx.has_b else None), None


define_op("layernorm(Tensor x, Tensor w, Tensor? b, float eps) -> (Tensor, Tensor, Tensor)", _layernorm_impl,
          lambda x, w, b, eps: (torch.empty_like(x), _f32(x, _rows(x)), _f32(x, _rows(x))),
          _layernorm_backward, _layernorm_setup)


# y = LayerNorm(x + residual) -> (y, h): BertSelfOutput / BertOutput (modeling_bert.py:289-293, :347-351)
def _add_layernorm_impl(x, residual, w, b, eps):
    return raw_layernorm_fwd(x, w, b, eps, residual)


def _add_layernorm_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[2], output[2], output[3])
    ctx.has_b = inputs[3] is not None
    ctx.set_materialize_grads(False)


def _add_layernorm_backward(ctx, dy, dh, _dm, _dr):
    if dy is None:
        return dh, dh, None, None, None
    h, w, mean, rstd = ctx.saved_tensors
    dx, dw, db = T.layernorm_bwd(dy, h, w, mean, rstd, dh, ctx.has_b)
    return dx, dx, dw, (db if ctx.has_b else None), None


define_op("add_layernorm(Tensor x, Tensor residual, Tensor w, Tensor? b, float eps) -> "
          "(Tensor, Tensor, Tensor, Tensor)", _add_layernorm_impl,
          lambda x, residual, w, b, eps: (torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x)),
                                          _f32(x, _rows(x))), _add_layernorm_backward, _add_layernorm_setup)


# y = LayerNorm(dropout(x, p) + residual) -> (y, h): BertSelfOutput / BertOutput in train mode
# (modeling_bert.py:289-293, :347-351); the keep mask is regenerated from (seed, element index) in the backward
def _dropout_add_layernorm_impl(x, residual, w, b, eps, dropout_p, seed):
    return raw_layernorm_dropout_fwd(x, w, b, eps, residual, dropout_p, seed)


def _dropout_add_layernorm_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[2], output[2], output[3])
    ctx.has_b = inputs[3] is not None
    ctx.drop = (inputs[5], inputs[6])
    ctx.set_materialize_grads(False)


def _dropout_add_layernorm_backward(ctx, dy, dh, _dm, _dr):
    none = (None,) * 7
    if dy is None:
        if dh is None:
            return none
        raise TamdError("dropout_add_layernorm: only the pre-norm sum is differentiated; use ops.layernorm pieces")
    h, w, mean, rstd = ctx.saved_tensors
    dx, dxd, dw, db = T.layernorm_dropout_bwd(dy, h, w, mean, rstd, ctx.drop[0], ctx.drop[1], dh, ctx.has_b)
    return (dxd, dx, dw, (db if ctx.has_b else None)) + none[4:]


define_op("dropout_add_layernorm(Tensor x, Tensor residual, Tensor w, Tensor? b, float eps, float dropout_p, int seed) "
          "-> (Tensor, Tensor, Tensor, Tensor)", _dropout_add_layernorm_impl,
          lambda x, residual, w, b, eps, dropout_p, seed: (torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x)),
                                                           _f32(x, _rows(x))),
          _dropout_add_layernorm_backward, _dropout_add_layernorm_setup)


# y = act(x W^T + b) [+ residual] on the MFMA GEMM; dX and dW use the k-major operand modes (no HBM transposes).
# nn.Linear call sites: see csrc/gemm.hip header.
def _linear_impl(x, w, bias, residual, act, train):
    k = x.shape[-1]
    x2 = _c(x).view(-1, k)
    epi, r2 = EPI_NONE, None
    if act != ACT_NONE and (bias is None or residual is not None):
        raise TamdError("activation epilogue needs a bias and no residual")
    if residual is not None:
        epi, r2 = EPI_RESIDUAL, _c(residual).view(-1, w.shape[0])
    elif bias is not None and act != ACT_NONE:
        epi = EPI_BIAS_ACT
    elif bias is not None:
        epi = EPI_BIAS
    if epi == EPI_BIAS_ACT and train:
        # keep the pre-activation for the backward: GEMM+bias, then the activation kernel
        pre = raw_gemm(x2, w, bias=bias, epilogue=EPI_BIAS)
        y = raw_bias_act_fwd(pre, None, act)
    else:
        pre = _nothing(x)
        y = raw_gemm(x2, w, bias=bias, residual=r2, epilogue=epi, act=act)
    return y.view(*x.shape[:-1], w.shape[0]), pre


def _linear_setup(ctx, inputs, output):
    x, w, bias, residual, act, _train = inputs
    ctx.save_for_backward(x, w, output[1])
    ctx.has_bias, ctx.has_res, ctx.act = bias is not None, residual is not None, act
    ctx.set_materialize_grads(False)


def _linear_backward(ctx, dy, _dpre):
    if dy is None:
        return None, None, None, None, None, None
    x, w, pre = ctx.saved_tensors
    n, k = w.shape
    dy2 = _c(dy).view(-1, n)
    dres = dy if ctx.has_res else None
    if ctx.act != ACT_NONE:
        if pre.numel() == 0:
            raise TamdError("linear(act=...) was run with train=False but is being differentiated")
        dy2 = T.bias_act_bwd(pre, None, dy2, ctx.act)
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = T.gemm(dy2, w, False, True).view(x.shape)                      # dX = dY . W
    if ctx.needs_input_grad[1]:
        dw = T.gemm(dy2, _c(x).view(-1, k), True, True)                     # dW = dY^T . X
    if ctx.has_bias and ctx.needs_input_grad[2]:
        db = T.colsum(dy2)
    return dx, dw, db, dres, None, None


define_op("linear(Tensor x, Tensor w, Tensor? bias, Tensor? residual, int act, bool train) -> (Tensor, Tensor)",
          _linear_impl,
          lambda x, w, bias, residual, act, train: (
              x.new_empty(*x.shape[:-1], w.shape[0]),
              x.new_empty(_rows(x), w.shape[0]) if (train and act != ACT_NONE) else _nothing(x)),
          _linear_backward, _linear_setup)


# y = x . Wf^T (+ bf), Wf = row-concatenation of the member weights (fused QKV / gate|up, fused_params.py).
# Gradients go to the member parameters: one fused dW GEMM, each member receives its row slice.
def _fused_linear_impl(x, wf, bf, members):
    x2 = _c(x).view(-1, x.shape[-1])
    y = raw_gemm(x2, wf, bias=bf, epilogue=EPI_BIAS if bf is not None else EPI_NONE)
    return y.view(*x.shape[:-1], wf.shape[0])


def _fused_linear_setup(ctx, inputs, output):
    x, wf, bf, members = inputs
    ctx.save_for_backward(x, wf)
    ctx.has_bias = bf is not None
    n_w = len(members) // 2 if bf is not None else len(members)
    ctx.splits = [m.shape[0] for m in members[:n_w]]


def _fused_linear_backward(ctx, dy):
    x, wf = ctx.saved_tensors
    dy2 = _c(dy).view(-1, wf.shape[0])
    dx = T.gemm(dy2, wf, False, True).view(x.shape) if ctx.needs_input_grad[0] else None
    grads = list(torch.split(T.gemm(dy2, _c(x).view(-1, x.shape[-1]), True, True), ctx.splits, dim=0))
    if ctx.has_bias:
        grads += list(torch.split(T.colsum(dy2), ctx.splits, dim=0))
    return dx, None, None, grads


define_op("fused_linear(Tensor x, Tensor wf, Tensor? bf, Tensor[] members) -> Tensor", _fused_linear_impl,
          lambda x, wf, bf, members: x.new_empty(*x.shape[:-1], wf.shape[0]), _fused_linear_backward,
          _fused_linear_setup)


# y = x @ W + b with W stored [in, out] (GPT-2 Conv1D, pytorch_utils.py:117-121): the k-major B operand
def _conv1d_impl(x, w, b):
    x2 = _c(x).view(-1, x.shape[-1])
    y = raw_gemm(x2, w, b_kn=True, bias=b, epilogue=EPI_BIAS if b is not None else EPI_NONE)
    return y.view(*x.shape[:-1], w.shape[1])


def _conv1d_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.has_bias = inputs[2] is not None


def _conv1d_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy2 = _c(dy).view(-1, w.shape[1])
    dx = T.gemm(dy2, w).view(x.shape)                                       # dX = dY . W^T  (W is [N=in, K=out])
    dw = T.gemm(_c(x).view(-1, x.shape[-1]), dy2, True, True)               # dW[in,out] = X^T . dY
    return dx, dw, (T.colsum(dy2) if ctx.has_bias else None)


define_op("conv1d(Tensor x, Tensor w, Tensor? b) -> Tensor", _conv1d_impl,
          lambda x, w, b: x.new_empty(*x.shape[:-1], w.shape[1]), _conv1d_backward, _conv1d_setup)


# Rotary embedding on the first `nheads` heads of a [B, S, row] projection output
# (models/llama/modeling_llama.py:130-160).  Out of place at this level (autograd needs the input intact).
def _rope_fn_impl(x, cos, sin, nheads, head_dim, conj=False):
    b, s, row = x.shape
    y = x.clone(memory_format=torch.contiguous_format)
    raw_rope_(y.view(b * s, row), cos, sin, s, nheads, head_dim, conj=conj)
    return y


def _rope_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1], inputs[2])
    ctx.meta = inputs[3:]


def _rope_backward(ctx, dy):
    cos, sin = ctx.saved_tensors
    nheads, head_dim, conj = ctx.meta
    return T.rope(dy, cos, sin, nheads, head_dim, not conj), None, None, None, None, None


define_op("rope(Tensor x, Tensor cos, Tensor sin, int nheads, int head_dim, bool conj=False) -> Tensor", _rope_fn_impl,
          lambda x, cos, sin, nheads, head_dim, conj=False: torch.empty_like(x, memory_format=torch.contiguous_format),
          _rope_backward, _rope_setup)


# softmax(scale QK^T + mask) V on [B,S,H,D] views.  Reference: eager_attention_forward,
# models/llama/modeling_llama.py:191-213 and siblings.
def _attention_impl(q, k, v, key_valid, scale, causal, dropout_p, seed, q_start, train):
    o, lse = raw_attn_fwd(q, k, v, scale, causal, key_valid, need_lse=train, dropout_p=dropout_p, seed=seed,
                          q_start=q_start)
    return o, lse if lse is not None else _nothing(q)


def _attention_setup(ctx, inputs, output):
    q, k, v, key_valid, scale, causal, dropout_p, seed, q_start, _train = inputs
    ctx.save_for_backward(q, k, v, output[0], output[1], key_valid, q_start)
    ctx.meta = (scale, causal, dropout_p, seed)
    ctx.set_materialize_grads(False)


def _attention_backward(ctx, do, _dlse):
    none = (None,) * 10
    if do is None:
        return none
    q, k, v, o, lse, key_valid, q_start = ctx.saved_tensors
    if lse.numel() == 0:
        raise TamdError("attention was run with train=False but is being differentiated")
    scale, causal, dropout_p, seed = ctx.meta
    dq, dk, dv = T.attn_bwd(q, k, v, o, lse, do, scale, causal, key_valid, dropout_p, seed, q_start)
    return (dq, dk, dv) + none[3:]


define_op("attention(Tensor q, Tensor k, Tensor v, Tensor? key_valid, float scale, bool causal, float dropout_p, "
          "int seed, Tensor? q_start, bool train) -> (Tensor, Tensor)", _attention_impl,
          lambda q, k, v, key_valid, scale, causal, dropout_p, seed, q_start, train: (
              q.new_empty(q.shape), _f32(q, q.shape[0], q.shape[2], q.shape[1]) if train else _nothing(q)),
          _attention_backward, _attention_setup)


# act = silu(gate) * up on a fused [T, 2I] projection output (modeling_llama.py:174-176)
def _swiglu_impl(gu):
    shape = gu.shape
    return raw_swiglu_fwd(_c(gu).view(-1, shape[-1])).view(*shape[:-1], shape[-1] // 2)


def _swiglu_backward(ctx, dact):
    (gu,) = ctx.saved_tensors
    gu2 = _c(gu).view(-1, gu.shape[-1])
    dgu, _ = T.swiglu_bwd(gu2, _c(dact).view(gu2.shape[0], -1), False)
    return dgu.view(gu.shape)


define_op("swiglu(Tensor gu) -> Tensor", _swiglu_impl, lambda gu: gu.new_empty(*gu.shape[:-1], gu.shape[-1] // 2),
          _swiglu_backward, lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


def _bias_act_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.act = inputs[2]


def _bias_act_backward(ctx, dy):
    x, bias = ctx.saved_tensors
    dx = T.bias_act_bwd(x, bias, dy, ctx.act)
    db = T.colsum(dx.view(-1, dx.shape[-1])) if bias is not None else None
    return dx, db, None


define_op("bias_act(Tensor x, Tensor? bias, int act) -> Tensor", raw_bias_act_fwd,
          lambda x, bias, act: torch.empty_like(x), _bias_act_backward, _bias_act_setup)


# nn.Embedding (models/llama/modeling_llama.py:381): bit-exact gather, sorted scatter-add backward
def _embedding_impl(ids, table, padding_idx=-1):
    return raw_embedding_fwd(ids, table)


def _embedding_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.meta = (inputs[1].shape[0], inputs[2])


def _embedding_backward(ctx, dout):
    (ids,) = ctx.saved_tensors
    vocab, padding_idx = ctx.meta
    return None, T.embedding_bwd(ids, dout, vocab, padding_idx), None


define_op("embedding(Tensor ids, Tensor table, int padding_idx=-1) -> Tensor", _embedding_impl,
          lambda ids, table, padding_idx=-1: table.new_empty(*ids.shape, table.shape[1]), _embedding_backward,
          _embedding_setup)


# BertEmbeddings.forward (modeling_bert.py:68-108) as one kernel: 3 gathers + 2 adds + LayerNorm
def _bert_embeddings_setup(ctx, inputs, output):
    input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, _ln_b, _eps, padding_idx, _train = inputs
    ctx.save_for_backward(input_ids, token_type_ids, position_ids, output[1], ln_w, output[2], output[3])
    ctx.meta = (word.shape[0], typ.shape[0], pos.shape[0], padding_idx)
    ctx.set_materialize_grads(False)


def _bert_embeddings_backward(ctx, dy, _dpre, _dm, _dr):
    none = (None,) * 11
    if dy is None:
        return none
    input_ids, token_type_ids, position_ids, pre, ln_w, mean, rstd = ctx.saved_tensors
    if pre.numel() == 0:
        raise TamdError("bert_embeddings was run with train=False but is being differentiated")
    vocab, tvocab, npos, padding_idx = ctx.meta
    d_pre, dw, db = T.layernorm_bwd(dy, pre, ln_w, mean, rstd)
    d_word = T.embedding_bwd(input_ids, d_pre, vocab, padding_idx)
    d_typ = T.embedding_bwd(token_type_ids, d_pre, tvocab, -1)
    d_pos = T.embedding_bwd(position_ids, d_pre, npos, -1)
    return (None, None, None, d_word, d_typ, d_pos, dw, db, None, None, None)


define_op("bert_embeddings(Tensor input_ids, Tensor token_type_ids, Tensor position_ids, Tensor word, Tensor typ, "
          "Tensor pos, Tensor ln_w, Tensor ln_b, float eps, int padding_idx, bool train) -> "
          "(Tensor, Tensor, Tensor, Tensor)",
          lambda input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, padding_idx, train:
          _bert_embeddings_fwd_impl(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, train),
          lambda input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, padding_idx, train: (
              word.new_empty(*input_ids.shape, word.shape[1]),
              word.new_empty(*input_ids.shape, word.shape[1]) if train else _nothing(word),
              _f32(word, input_ids.numel()), _f32(word, input_ids.numel())),
          _bert_embeddings_backward, _bert_embeddings_setup)


# fixed_cross_entropy on `logits.float()` (loss/loss_utils.py:32-46) without materialising fp32 logits.
# Returns the SUM of per-token losses; the caller divides (mean over valid labels or num_items_in_batch).
def _cross_entropy_sum_impl(logits2d, labels, ignore_index=-100):
    lse, row_loss = raw_cross_entropy_fwd(logits2d, labels, ignore_index)
    return row_loss.sum(), lse


def _cross_entropy_sum_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])
    ctx.ignore_index = inputs[2]
    ctx.set_materialize_grads(False)


def _cross_entropy_sum_backward(ctx, g, _dlse):
    if g is None:
        return None, None, None
    logits2d, labels, lse = ctx.saved_tensors
    gs = g.detach().to(torch.float32).reshape(1).contiguous()
    return T.cross_entropy_bwd(logits2d, labels, lse, gs, ctx.ignore_index), None, None


define_op("cross_entropy_sum(Tensor logits2d, Tensor labels, int ignore_index=-100) -> (Tensor, Tensor)",
          _cross_entropy_sum_impl,
          lambda logits2d, labels, ignore_index=-100: (_f32(logits2d), _f32(logits2d, logits2d.shape[0])),
          _cross_entropy_sum_backward, _cross_entropy_sum_setup)


# lm_head + causal-LM loss without ever holding the [tokens, vocab] logits (reference:
# `logits = self.lm_head(hidden)` then ForCausalLMLoss, modeling_llama.py:477-484 / loss/loss_utils.py:49-71).
# Tokens are processed in chunks: logits_c = h_c W^T (MFMA GEMM) -> cross-entropy forward (lse, per-token loss) ->
# dlogits_c, already scaled by 1/normaliser -> dh_c = dlogits_c W and dW += dlogits_c^T h_c (accumulate epilogue).
# The same three GEMMs as the unfused path, no recomputation; the gradients are produced in the forward and only
# multiplied by the upstream scalar in the backward.  Peak extra memory: one chunk of logits instead of 2 x [T, V].
# dW accumulates in the storage dtype across chunks (<= 8 roundings at the default chunking).
def _linear_cross_entropy_impl(h2d, w, labels, normaliser, ignore_index, chunk, need_dh, need_dw):
    t = h2d.shape[0]
    gs = (1.0 / normaliser.to(torch.float32)).reshape(1).contiguous()
    loss = torch.zeros((), dtype=torch.float32, device=h2d.device)
    dh = torch.empty_like(h2d) if need_dh else _nothing(h2d)
    dw = torch.empty_like(w) if need_dw else _nothing(w)
    first = True
    for c0 in range(0, t, chunk):
        c1 = min(c0 + chunk, t)
        hc, lc = h2d[c0:c1], labels[c0:c1]
        logits = raw_gemm(hc, w)
        lse, row_loss = raw_cross_entropy_fwd(logits, lc, ignore_index)
        loss = loss + row_loss.sum()
        if need_dh or need_dw:
            dlog = raw_cross_entropy_bwd(logits, lc, lse, gs, ignore_index)
            del logits
            if need_dh:
                raw_gemm(dlog, w, b_kn=True, out=dh[c0:c1])
            if need_dw:
                raw_gemm(dlog, hc, a_km=True, b_kn=True, epilogue=EPI_NONE if first else EPI_ACCUM, out=dw)
            del dlog
        first = False
    return loss * gs[0], dh, dw


def _linear_cross_entropy_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], output[2])
    ctx.set_materialize_grads(False)


def _linear_cross_entropy_backward(ctx, g, _ddh, _ddw):
    none = (None,) * 8
    if g is None:
        return none
    dh, dw = ctx.saved_tensors
    g = g.detach()
    if (ctx.needs_input_grad[0] and dh.numel() == 0) or (ctx.needs_input_grad[1] and dw.numel() == 0):
        raise TamdError("linear_cross_entropy: a gradient is requested that the forward was told not to produce")
    return ((dh * g.to(dh.dtype)) if ctx.needs_input_grad[0] else None,
            (dw * g.to(dw.dtype)) if ctx.needs_input_grad[1] else None) + none[2:]


define_op("linear_cross_entropy(Tensor h2d, Tensor w, Tensor labels, Tensor normaliser, int ignore_index, int chunk, "
          "bool need_dh, bool need_dw) -> (Tensor, Tensor, Tensor)", _linear_cross_entropy_impl,
          lambda h2d, w, labels, normaliser, ignore_index, chunk, need_dh, need_dw: (
              _f32(h2d), torch.empty_like(h2d) if need_dh else _nothing(h2d),
              torch.empty_like(w) if need_dw else _nothing(w)),
          _linear_cross_entropy_backward, _linear_cross_entropy_setup)


# Vocabulary projection whose width is not a multiple of 8 (+ optional token-level cross-entropy): BERT's MLM head,
# `self.decoder(hidden)` then `CrossEntropyLoss()(scores.view(-1, V), labels.view(-1))` (models/bert/modeling_bert.py:
# 483-496, 970-975; bert-base: V = 30522).  The GEMM stores 16-byte row segments, so the weight / bias live in buffers
# padded with ZERO rows to Vp = a multiple of 64 (fused_params.PaddedRows: the tied parameters are row-slice views of them),
# the logits are computed as [M, Vp] and handed out as the [M, V] view of that buffer (row stride Vp), and the backward runs
# the dX / dW GEMMs on the K-padded gradient buffer the loss kernel writes (padding columns zero).
#   -> (loss_sum fp32 scalar: SUM of the per-token losses, 0 when labels is None; logits [M, V] (strided); lse [M] fp32)
def _padded_vocab_head_impl(h2d, w_pad, b_pad, w, b, labels, ignore_index, train):
    n = w.shape[0]
    logits_pad = raw_gemm(h2d, w_pad, bias=b_pad, epilogue=EPI_BIAS if b_pad is not None else EPI_NONE)
    logits = logits_pad[:, :n] if n != w_pad.shape[0] else logits_pad
    if labels is None:
        return torch.zeros((), dtype=torch.float32, device=h2d.device), logits, _f32(h2d, 0)
    lse, row_loss = raw_cross_entropy_fwd(logits, labels, ignore_index)
    return row_loss.sum(), logits, lse


def _padded_vocab_head_fake(h2d, w_pad, b_pad, w, b, labels, ignore_index, train):
    m, vp = h2d.shape[0], w_pad.shape[0]
    logits = torch.empty_strided((m, w.shape[0]), (vp, 1), dtype=h2d.dtype, device=h2d.device)
    return _f32(h2d), logits, _f32(h2d, m if labels is not None else 0)


def _padded_vocab_head_setup(ctx, inputs, output):
    h2d, w_pad, _b_pad, _w, b, labels, ignore_index, _train = inputs
    ctx.save_for_backward(h2d, w_pad, labels, output[1], output[2])
    ctx.has_bias, ctx.ignore_index = b is not None, ignore_index
    ctx.set_materialize_grads(False)


def _padded_vocab_head_backward(ctx, g_loss, g_logits, _dlse):
    none = (None,) * 8
    if g_loss is None and g_logits is None:
        return none
    h2d, w_pad, labels, logits, lse = ctx.saved_tensors
    m, n = logits.shape
    vp = w_pad.shape[0]
    dlog = None
    if g_loss is not None and labels is not None:
        gs = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        dlog = raw_cross_entropy_bwd(logits, labels, lse, gs, ctx.ignore_index, padded=True)  # [M, Vp], padding zero
    if g_logits is not None:  # the scores themselves were differentiated (a custom loss on `logits`): generic path
        if dlog is None:
            dlog = torch.zeros(m, vp, dtype=logits.dtype, device=logits.device)
        dlog[:, :n] += g_logits
    dh = dw = db = None
    if ctx.needs_input_grad[0]:
        dh = raw_gemm(dlog, w_pad, b_kn=True)                            # dX = dY . W        (K = Vp, padded with zeros)
    if ctx.needs_input_grad[3]:
        dw = raw_gemm(dlog, h2d, a_km=True, b_kn=True)[:n]               # dW = dY^T . X      (rows V .. Vp-1 dropped)
    if ctx.has_bias and ctx.needs_input_grad[4]:
        db = raw_colsum(dlog)[:n]
    return (dh, None, None, dw, db) + none[5:]


define_op("padded_vocab_head(Tensor h2d, Tensor w_pad, Tensor? b_pad, Tensor w, Tensor? b, Tensor? labels, "
          "int ignore_index, bool train) -> (Tensor, Tensor, Tensor)", _padded_vocab_head_impl, _padded_vocab_head_fake,
          _padded_vocab_head_backward, _padded_vocab_head_setup)


# --------------------------------------------------------------------------- Python-level wrappers
def rmsnorm(x, w, eps, residual=None):
    """-> y  (or (y, h) with h = x + residual when a residual is given)."""
    if residual is None:
        return T.rmsnorm(x, w, float(eps))[0]
    y, h, _ = T.add_rmsnorm(x, residual, w, float(eps))
    return y, h


def layernorm(x, w, b, eps, residual=None):
    if residual is None:
        return T.layernorm(x, w, b, float(eps))[0]
    y, h, _, _ = T.add_layernorm(x, residual, w, b, float(eps))
    return y, h


def dropout_add_layernorm(x, residual, w, b, eps, dropout_p, seed=None):
    """LayerNorm(dropout(x, p) + residual) -> y, the dropout inside the norm kernel (seed from torch's CPU generator
    unless given: `torch.manual_seed` repeats it, activation checkpointing regenerates it)."""
    if seed is None:
        seed = dropout_seed()
    return T.dropout_add_layernorm(x, residual, w, b, float(eps), float(dropout_p), int(seed))[0]


def hidden_dropout_keep_mask(seed: int, rows: int, cols: int, p: float) -> torch.Tensor:
    """The keep mask [rows, cols] of `dropout_add_layernorm`, rebuilt on the host (tests / debugging)."""
    return dropout_keep_mask(seed, 1, 1, rows, cols, p)[0, 0]


def linear(x, w, bias=None, residual=None, act=ACT_NONE):
    return T.linear(x, w, bias, residual, int(act), _wants_grad(x, w, bias, residual))[0]


def fused_linear(x, wf, bf, members):
    return T.fused_linear(x, wf, bf, list(members))


def conv1d(x, w, b=None):
    return T.conv1d(x, w, b)


def rope(x, cos, sin, nheads, head_dim):
    return T.rope(x, cos, sin, int(nheads), int(head_dim))


def swiglu(gu):
    return T.swiglu(gu)


def bias_act(x, bias, act):
    return T.bias_act(x, bias, int(act))


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
    return T.attention(q, k, v, key_valid, float(scale), bool(causal), float(dropout_p), int(seed or 0), q_start,
                       _wants_grad(q, k, v))[0]


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


def embedding(ids, table, padding_idx=None):
    return T.embedding(ids, table, -1 if padding_idx is None else int(padding_idx))


def cross_entropy_sum(logits2d, labels, ignore_index=-100):
    return T.cross_entropy_sum(logits2d, labels, int(ignore_index))[0]


def causal_lm_loss(logits, labels, vocab_size, num_items_in_batch=None, ignore_index=-100, shift_labels=None,
                   **_unused):
    """Drop-in for ForCausalLMLoss (loss/loss_utils.py:49-71): same shifting, same reductions."""
    if shift_labels is None:
        labels = torch.nn.functional.pad(labels, (0, 1), value=ignore_index)
        shift_labels = labels[..., 1:].contiguous()
    logits2d = logits.reshape(-1, vocab_size)
    shift_labels = shift_labels.reshape(-1).to(logits.device)
    total = cross_entropy_sum(logits2d, shift_labels, ignore_index)
    if num_items_in_batch is not None:
        if torch.is_tensor(num_items_in_batch):
            num_items_in_batch = num_items_in_batch.to(total.device)
        return total / num_items_in_batch
    n_valid = (shift_labels != ignore_index).sum()
    return total / n_valid


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
    grad = torch.is_grad_enabled()
    return T.linear_cross_entropy(h2d, weight, labels, norm, int(ignore_index), int(chunk_tokens),
                                  grad and hidden.requires_grad, grad and weight.requires_grad)[0]


def padded_vocab_head(hidden, w_pad, b_pad, w, b, labels=None, ignore_index=-100):
    """`hidden @ w.T + b` for a vocabulary width that is not a multiple of 8, through zero-padded weight / bias buffers
    (fused_params.PaddedRows), with the token-level cross-entropy of `labels` (mean over the labels != ignore_index, what
    `CrossEntropyLoss()` computes) when labels are given.  -> (loss or None, logits [..., V])"""
    h2d = _c(hidden).view(-1, hidden.shape[-1])
    lab = None if labels is None else labels.reshape(-1).to(hidden.device).contiguous()
    loss_sum, logits, _ = T.padded_vocab_head(h2d, w_pad, b_pad, w, b, lab, int(ignore_index),
                                              _wants_grad(hidden, w, b))
    out = logits.view(*hidden.shape[:-1], w.shape[0])  # (splitting the leading dimension of a row-strided matrix is a view)
    if lab is None:
        return None, out
    return loss_sum / (lab != ignore_index).sum(), out


def bert_embeddings(input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, padding_idx=None):
    """BertEmbeddings.forward (modeling_bert.py:68-108) minus the dropout: 3 gathers + 2 adds + LayerNorm, one kernel."""
    train = _wants_grad(word, typ, pos, ln_w, ln_b)
    return T.bert_embeddings(_c(input_ids), _c(token_type_ids), _c(position_ids), word, typ, pos, ln_w, ln_b,
                             float(eps), -1 if padding_idx is None else int(padding_idx), train)[0]
