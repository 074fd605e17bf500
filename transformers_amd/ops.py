"""torch-level entry points of the MI355X kernels.

Three layers, all thin:

1. ``torch.ops.tamd.*`` -- the dispatcher ops.  Schemas and implementations are COMPILED: ``csrc/torch_binding.cpp``
                 (``libtamd_torch.so``, ``TORCH_LIBRARY(tamd, ...)``) checks operands, allocates outputs with torch's
                 caching allocator and calls the C-ABI entry points of ``include/tamd.h`` on the calling thread's current
                 HIP stream -- kernel-level ops (``gemm``, ``attn_fwd``, ``rmsnorm_bwd`` ...) and the forward / backward
                 bodies of the differentiable ones (``linear``, ``attention``, ``llama_layer``, ``bert_layer`` ...: a whole
                 decoder layer is one dispatcher call that issues all its launches from C++).
2. this module   registers what belongs to Python with ``torch.library`` -- the fake (Meta) implementations and the
                 autograd formulas (the contract of SURVEY.md section 8a; reference precedent for op + fake + autograd:
                 src/transformers/integrations/moe.py:245-257) -- and binds the C-ABI library the compiled ops call into:
                 ``libtamd.so`` in the product, ``libtamd_diag.so`` for the measurement tools, the CPU execution model of
                 the same kernels in the CPU test-suite.
3. wrappers      ``ops.linear(...)``, ``ops.attention(...)`` ...: Python conveniences (default arguments, the "does
                 anything need a gradient" flag) around ``torch.ops.tamd.*``; ``ops.raw_*`` are keyword-friendly aliases of
                 the kernel-level ops for tests and tools.  The model code under ``transformers_amd/models/`` uses only the
                 wrappers and ``torch.ops.tamd.*``.

The HIP library is mandatory: there is no CPU or eager fallback in this module.  If libtamd.so / libtamd_torch.so are
missing, or a tensor is not on a GPU, the call raises.
"""
from __future__ import annotations

import threading
from typing import Optional

import torch

from . import _cabi, _native
from ._cabi import (ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, ACT_QUICK_GELU, ACT_SILU, EPI_ACCUM, EPI_BIAS,
                    EPI_BIAS_ACT, EPI_NONE, EPI_RESIDUAL, GEMM_A_KM, GEMM_B_KN, TamdError)

_DTYPE_CODE = {torch.bfloat16: _cabi.TAMD_BF16, torch.float16: _cabi.TAMD_F16, torch.float32: _cabi.TAMD_F32}
ACT_CODES = {"none": ACT_NONE, "gelu": ACT_GELU_ERF, "gelu_new": ACT_GELU_TANH, "gelu_pytorch_tanh": ACT_GELU_TANH,
             "quick_gelu": ACT_QUICK_GELU, "silu": ACT_SILU, "swish": ACT_SILU}


# --------------------------------------------------------------------------- backend
class HipBackend:
    """libtamd.so: what the compiled ops are bound to in the product.  `.lib` is the ctypes view of the same library
    (ABI version check; the diagnostic entry points of libtamd_diag.so for the tools)."""

    name = "hip"

    def __init__(self):
        path = _cabi.default_library_path()
        if not path.exists():
            raise TamdError(
                f"{path} not found: build it with `python -m transformers_amd.build` "
                "(the MI355X path has no CPU/eager fallback)")
        self.lib = _cabi.TamdLib(path)

    def stream(self, t: torch.Tensor):
        """Raw handle of torch's current HIP stream on the tensor's device (for direct C-ABI calls in tests / tools)."""
        return torch.cuda.current_stream(t.device).cuda_stream


_backend = None
_backend_lock = threading.Lock()


def backend():
    global _backend
    if _backend is None:
        with _backend_lock:
            if _backend is None:
                b = HipBackend()
                _native.bind(b.lib.path, emulated=False)
                _backend = b
    return _backend


def _set_backend(b):
    """Hook for the tools (libtamd_diag.so) and the CPU test-suite (tests/hipemu installs the CPU execution model of the
    same kernels here): the compiled ops are re-bound to `b`'s library; None = back to the product library."""
    global _backend
    old, _backend = _backend, b
    if b is None:
        backend()
    else:
        _native.bind(b.lib.path, emulated=b.name != "hip")
    return old


try:  # bind the product library now: the compiled ops do not pass through `backend()`
    backend()
except TamdError:  # libtamd.so not built yet: the first op call says so
    pass


def backend_is_emulated() -> bool:
    """True only while tests/hipemu has installed the CPU execution model of the kernels."""
    return _backend is not None and _backend.name != "hip"


def _code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise TamdError(f"unsupported dtype {t.dtype}") from None


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_native.load()  # registers the schemas and implementations of torch.ops.tamd.* (raises if the library is not built)
T = torch.ops.tamd  # the compiled op namespace (csrc/torch_binding.cpp)


# --------------------------------------------------------------------------- kernel-level ops by keyword (tests, tools)
# One alias per C-ABI entry point: `torch.ops.tamd.<kernel op>` with keyword arguments and None for absent tensors.
def _opt(t):
    return t if (t is not None and t.numel() > 0) else None


def raw_rmsnorm_fwd(x, w, eps, residual=None):
    """-> (y, h, rstd); h is x+residual (or x itself when residual is None)."""
    y, h, rstd = T.rmsnorm_fwd(x, w, float(eps), residual)
    return y, (h if residual is not None else x), rstd


def raw_rmsnorm_bwd(dy, h, w, rstd, dres=None):
    return T.rmsnorm_bwd(dy, h, w, rstd, dres)


def raw_layernorm_fwd(x, w, b, eps, residual=None):
    y, h, mean, rstd = T.layernorm_fwd(x, w, b, float(eps), residual)
    return y, (h if residual is not None else x), mean, rstd


def raw_layernorm_bwd(dy, h, w, mean, rstd, dres=None, need_db=True):
    dx, dw, db, _ = T.layernorm_bwd(dy, h, w, mean, rstd, dres, need_db, False)
    return dx, dw, (db if need_db else None)


def raw_layernorm_dropout_fwd(x, w, b, eps, residual, dropout_p, seed, seed_dev=None):
    """h = dropout(x, p) + residual; y = LayerNorm(h)  ->  (y, h, mean, rstd).  Keep mask: the counter-based hash of
    (seed, flat element index), `hidden_dropout_keep_mask` on the host.  seed_dev: the seed as an int64[1] device tensor
    (replaces `seed`: what a captured training step uses)."""
    return T.layernorm_dropout_fwd(x, w, b, float(eps), residual, float(dropout_p), int(seed) & 0x7FFFFFFFFFFFFFFF, seed_dev)


def raw_layernorm_dropout_bwd(dy, h, w, mean, rstd, dropout_p, seed, dres=None, need_db=True, seed_dev=None):
    """-> (dx = gradient of the residual input, dx_drop = gradient of the dropped-out input, dw, db)."""
    dx, dxd, dw, db, _ = T.layernorm_dropout_bwd(dy, h, w, mean, rstd, float(dropout_p), int(seed) & 0x7FFFFFFFFFFFFFFF, dres,
                                              need_db, False, seed_dev)
    return dx, dxd, dw, (db if need_db else None)


def raw_rope_(x2d, cos, sin, seq, nheads, head_dim, conj=False):
    """In-place rotary on the first `nheads` heads of every row of x2d [tokens, row_stride]."""
    T.rope_(x2d, cos, sin, int(seq), int(nheads), int(head_dim), bool(conj))
    return x2d


def raw_embedding_fwd(ids, table):
    return T.embedding_fwd(ids, table)


def raw_embedding_bwd(ids, dout, vocab, padding_idx=-1):
    return T.embedding_bwd(ids, dout, int(vocab), -1 if padding_idx is None else int(padding_idx))


def raw_swiglu_fwd(gu):
    """gu [T, 2I] = [gate | up]  ->  act [T, I]"""
    return T.swiglu_fwd(gu)


def raw_swiglu_bwd(gu, dact, want_act=False):
    dgu, act = T.swiglu_bwd(gu, dact, want_act)
    return dgu, (act if want_act else None)


def raw_bias_act_fwd(x, bias, act):
    return T.bias_act_fwd(x, bias, int(act))


def raw_bias_act_bwd(x, bias, dy, act, need_colsum=False):
    """-> dx, or (dx, column sums of dx = the bias gradient) with need_colsum."""
    dx, dc = T.bias_act_bwd(x, bias, dy, int(act), bool(need_colsum))
    return (dx, dc) if need_colsum else dx


def raw_add(a, b):
    return T.add(a, b)


def raw_adamw_step_(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """In-place fused AdamW step on one tensor (p, m, v updated); torch.optim.AdamW semantics, include/tamd.h."""
    T.adamw_step_(p, g, m, v, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                  float(grad_scale))


def raw_colsum(x2d):
    return T.colsum(x2d)


def raw_transpose(x2d):
    return T.transpose(x2d)


def raw_cross_entropy_fwd(logits2d, labels, ignore_index=-100):
    return T.cross_entropy_fwd(logits2d, labels, int(ignore_index))


def raw_cross_entropy_bwd(logits2d, labels, lse, gscale, ignore_index=-100):
    """-> dlogits [t, v], a view of a fresh [t, ld] buffer (ld = the row stride of logits2d) whose padding columns
    v .. ld-1 the kernel zeroes."""
    return T.cross_entropy_bwd(logits2d, labels, lse, gscale, int(ignore_index))


def gemm_supported(m, n, k, dtype) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and k % 8 == 0 and n % 8 == 0 and m > 0


def gemm_workspace_bytes(m: int, n: int, k: int, epilogue: int, flags: int = 0) -> int:
    """Host-side mirror of tamd_gemm_workspace_bytes (csrc/gemm.hip gemm_choose_splits): one ctypes round trip per
    GEMM is ~10 us, which small-model steps (hundreds of 30-us kernels) cannot hide.  tests/test_kernels.py keeps the
    two in step."""
    if k % 64 or n % 4 or epilogue not in (EPI_NONE, EPI_BIAS, EPI_RESIDUAL, EPI_ACCUM):
        return 0
    if (flags & 3) == 0 and m <= 16 and epilogue <= EPI_RESIDUAL and not (m > 4 and n >= 65536):  # csrc/gemv.hip
        return 0
    tiles, nst = -(-m // 256) * -(-n // 256), k // 64
    if nst < 32:
        return 0
    if (flags & 3) == 0 and tiles <= 128 and nst < 64:  # forward product, short K: the 128 x 128 kernel, no split
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


GEMM_SCHED = {None: 0, "pp": 1, "sm": 2, "fl": 3}  # TAMD_GEMM_SCHED_* >> 8 (include/tamd.h)


def raw_gemm(a, b, *, a_km=False, b_kn=False, bias=None, residual=None, epilogue=EPI_NONE, act=ACT_NONE, out=None,
             sched=None):
    """C[M,N] = epi(A . B^T).  a: [M,K] (or [K,M] if a_km); b: [N,K] (or [K,N] if b_kn).  A schedule hint turns the
    split-K policy off."""
    if out is None:
        return T.gemm(a, b, a_km, b_kn, bias, residual, int(epilogue), int(act), GEMM_SCHED[sched])
    T.gemm_out(out, a, b, a_km, b_kn, bias, residual, int(epilogue), int(act), GEMM_SCHED[sched])
    return out


def attn_bwd_rope_supported(q, k, cos, head_dim) -> bool:
    """The attention backward can apply the transposed rotary embedding to dq / dk on their way out."""
    return head_dim == 128 and q.shape[1] == k.shape[1] and cos.shape[-1] == 128 and cos.dim() in (2, 3)


def gemm_swiglu_supported(x2, wgu) -> bool:
    """Shapes the fused gate|up GEMM + SiLU*up epilogue takes (csrc/gemm.hip tamd_gemm_swiglu; mirrored in torch_binding.cpp)."""
    two_i, k = wgu.shape
    return (x2.dtype in (torch.bfloat16, torch.float16) and wgu.dtype == x2.dtype and k % 64 == 0 and two_i % 16 == 0
            and x2.stride(1) == 1 and wgu.stride(1) == 1 and x2.stride(0) % 8 == 0 and wgu.stride(0) % 8 == 0
            and two_i * wgu.stride(0) * 2 < 2 ** 31
            and (x2.shape[0] <= 16  # (a decode step: csrc/gemv.hip behind the same entry point)
                 or gemm_workspace_bytes(x2.shape[0], two_i, k, EPI_NONE, 3) == 0))  # (small grids: plain GEMM + swiglu kernel)


def raw_gemm_swiglu(x2, wgu, need_gu=True):
    """x2 [T, K], wgu [2I, K] = [gate_proj.weight ; up_proj.weight]  ->  (gu [T, 2I] or None, act [T, I])."""
    gu, act = T.gemm_swiglu(x2, wgu, need_gu)
    return (gu if need_gu else None), act


def gemm_swiglu_bwd_supported(dy, wd, gu) -> bool:
    """Shapes the down projection's dX GEMM takes with the SiLU*up backward as its way out (csrc/gemm.hip tamd_gemm_swiglu_bwd):
    full 256 x 256 grids; elsewhere the plain product + `raw_swiglu_bwd`.  Mirrored in torch_binding.cpp."""
    k, inter = wd.shape
    t = dy.shape[0]
    return (dy.dtype in (torch.bfloat16, torch.float16) and wd.dtype == dy.dtype and gu.dtype == dy.dtype and k % 64 == 0
            and inter % 8 == 0 and dy.stride(1) == 1 and wd.stride(1) == 1 and dy.stride(0) % 8 == 0 and wd.stride(0) % 8 == 0
            and gu.is_contiguous() and tuple(gu.shape) == (t, 2 * inter) and 128 * 2 * inter * 2 < 2 ** 31 and t > 16
            and gemm_workspace_bytes(t, inter, k, EPI_NONE, 2) == 0)  # (flags 2 = TAMD_GEMM_B_KN)


def raw_gemm_swiglu_bwd(dy, wd, gu):
    """dy [T, K], wd [K, I] = down_proj.weight, gu [T, 2I] = the forward's gate | up  ->  d_gu [T, 2I]; d_act = dy . wd is never
    written.  Bit-identical to `raw_gemm(dy, wd, b_kn=True)` followed by `raw_swiglu_bwd`."""
    return T.gemm_swiglu_bwd(dy, wd, gu)


def raw_attn_fwd(q, k, v, scale, causal, key_valid=None, need_lse=True, dropout_p=0.0, seed=0, q_start=None, seed_dev=None):
    """q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (strided views fine) -> o [B,Sq,Hq,D] contiguous, lse [B,Hq,Sq] fp32."""
    o, lse = T.attn_fwd(q, k, v, float(scale), bool(causal), key_valid, need_lse, float(dropout_p),
                        int(seed) & 0x7FFFFFFFFFFFFFFF, q_start, seed_dev)
    return o, (lse if need_lse else None)


def raw_attn_bwd(q, k, v, o, lse, dout, scale, causal, key_valid=None, dq=None, dk=None, dv=None,
                 dropout_p=0.0, seed=0, q_start=None, rope=None, seed_dev=None):
    """Gradients written into dq/dk/dv (views with the strides of q/k/v) or freshly allocated.  rope = (cos, sin)
    ([seq, 128] or [batch, seq, 128], the storage dtype): q and k had been rotated before the attention, dq and dk leave
    through the transposed rotation (the same bits as raw_rope_(conj=True) on the stored gradients)."""
    cos, sin = rope if rope is not None else (None, None)
    seed = int(seed) & 0x7FFFFFFFFFFFFFFF
    if dq is None:
        return T.attn_bwd(q, k, v, o, lse, dout, float(scale), bool(causal), key_valid, float(dropout_p), seed, q_start,
                          cos, sin, seed_dev)
    T.attn_bwd_out(dq, dk, dv, q, k, v, o, lse, dout, float(scale), bool(causal), key_valid, float(dropout_p), seed,
                   q_start, cos, sin, seed_dev)
    return dq, dk, dv


# --------------------------------------------------------------------------- fake (Meta) implementations + autograd
# The ops are defined and implemented in csrc/torch_binding.cpp; what Python adds per op is its fake implementation
# (shapes without data: tracing / shape tooling / opcheck) and, for the differentiable ones, the autograd formula --
# the reference's own precedent for custom ops: src/transformers/integrations/moe.py:245-257.
_LIB = torch.library.Library("tamd", "FRAGMENT")


def register(name: str, fake, backward=None, setup_context=None):
    """Attach a fake (Meta) implementation and -- optionally -- an autograd formula to the compiled op `tamd::<name>`."""
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
register("rmsnorm_fwd", lambda x, w, eps, residual=None: (torch.empty_like(x),
                                                          torch.empty_like(x) if residual is not None else _nothing(x),
                                                          _f32(x, _rows(x))))
register("rmsnorm_bwd", lambda dy, h, w, rstd, dres=None: (torch.empty_like(h), torch.empty_like(w)))
register("layernorm_fwd", lambda x, w, b, eps, residual=None: (torch.empty_like(x),
                                                               torch.empty_like(x) if residual is not None else _nothing(x),
                                                               _f32(x, _rows(x)), _f32(x, _rows(x))))
register("layernorm_bwd", lambda dy, h, w, mean, rstd, dres=None, need_db=True, need_colsum=False: (
    torch.empty_like(h), torch.empty_like(w), torch.empty_like(w) if need_db else _nothing(w),
    torch.empty_like(w) if need_colsum else _nothing(w)))
register("layernorm_dropout_fwd", lambda x, w, b, eps, residual, dropout_p, seed, seed_dev=None: (
    torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x)), _f32(x, _rows(x))))
register("layernorm_dropout_bwd", lambda dy, h, w, mean, rstd, dropout_p, seed, dres=None, need_db=True, need_colsum=False, seed_dev=None: (
    torch.empty_like(h), torch.empty_like(h), torch.empty_like(w), torch.empty_like(w) if need_db else _nothing(w),
    torch.empty_like(w) if need_colsum else _nothing(w)))
register("rope_", lambda x2d, cos, sin, seq, nheads, head_dim, conj=False: None)
register("embedding_fwd", lambda ids, table: table.new_empty(*ids.shape, table.shape[1]))
register("embedding_bwd", lambda ids, dout, vocab, padding_idx=-1: dout.new_empty(vocab, dout.shape[-1]))
register("bert_embeddings_fwd",
         lambda input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, keep_pre_ln: (
             word.new_empty(*input_ids.shape, word.shape[1]),
             word.new_empty(*input_ids.shape, word.shape[1]) if keep_pre_ln else _nothing(word),
             _f32(word, input_ids.numel()), _f32(word, input_ids.numel())))
register("swiglu_fwd", lambda gu: gu.new_empty(gu.shape[0], gu.shape[1] // 2))
register("swiglu_bwd", lambda gu, dact, want_act=False: (torch.empty_like(gu),
                                                         torch.empty_like(dact) if want_act else _nothing(gu)))
register("bias_act_fwd", lambda x, bias, act: torch.empty_like(x))
register("bias_act_bwd", lambda x, bias, dy, act, need_colsum=False: (
    torch.empty_like(x), x.new_empty(x.shape[-1]) if need_colsum else _nothing(x)))
register("gemm_colscale", lambda x2, w, bias, scale_cols, col_scale: x2.new_empty(x2.shape[0], w.shape[0]))
register("gemm_dw_segments", lambda dy, x, segs: None)
register("gemm_dw_group", lambda dy, x: [a.new_empty(a.shape[1], b.shape[1]) for a, b in zip(dy, x)])
register("add", lambda a, b: torch.empty_like(a))
register("colsum", lambda x2d: x2d.new_empty(x2d.shape[1]))
register("transpose", lambda x2d: x2d.new_empty(x2d.shape[1], x2d.shape[0]))
register("cross_entropy_fwd", lambda logits2d, labels, ignore_index=-100: (_f32(logits2d, logits2d.shape[0]),
                                                                            _f32(logits2d, logits2d.shape[0])))
register("cross_entropy_bwd", lambda logits2d, labels, lse, gscale, ignore_index=-100: torch.empty_strided(
    logits2d.shape, (logits2d.stride(0), 1), dtype=logits2d.dtype, device=logits2d.device))
register("adamw_step_", lambda p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0: None)
register("mt_sumsq", lambda table, n_tensors, total_chunks, partials, dtype: None)
register("mt_norm_finish", lambda partials, out, max_norm: None)
register("mt_scale_", lambda table, n_tensors, total_chunks, coef, dtype: None)
register("mt_adamw_step_", lambda table, n_tensors, total_chunks, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                                  grad_scale_dev, dtype, state_dtype: None)


def _gemm_shape(a, b, a_km, b_kn):
    return (a.shape[1] if a_km else a.shape[0]), (b.shape[1] if b_kn else b.shape[0])


register("gemm", lambda a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0, sched=0: a.new_empty(
    *_gemm_shape(a, b, a_km, b_kn)))
register("gemm_out", lambda out, a, b, a_km=False, b_kn=False, bias=None, residual=None, epilogue=0, act=0, sched=0: None)
register("gemm_swiglu", lambda x2, wgu, need_gu=True: (x2.new_empty(x2.shape[0], wgu.shape[0]) if need_gu else _nothing(x2),
                                                       x2.new_empty(x2.shape[0], wgu.shape[0] // 2)))
register("gemm_swiglu_bwd", lambda dy, wd, gu: torch.empty_like(gu))
register("attn_fwd", lambda q, k, v, scale, causal, key_valid=None, need_lse=True, dropout_p=0.0, seed=0, q_start=None, seed_dev=None: (
    q.new_empty(q.shape), _f32(q, q.shape[0], q.shape[2], q.shape[1]) if need_lse else _nothing(q)))
register("attn_bwd", lambda q, k, v, o, lse, dout, scale, causal, key_valid=None, dropout_p=0.0, seed=0, q_start=None,
         rope_cos=None, rope_sin=None, seed_dev=None: (
             torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device),
             torch.empty_strided(k.shape, k.stride(), dtype=k.dtype, device=k.device),
             torch.empty_strided(v.shape, v.stride(), dtype=v.dtype, device=v.device)))
register("attn_bwd_out", lambda dq, dk, dv, q, k, v, o, lse, dout, scale, causal, key_valid=None, dropout_p=0.0, seed=0,
         q_start=None, rope_cos=None, rope_sin=None, seed_dev=None: None)


# ---- differentiable ops --------------------------------------------------------------------------------------
# Convention: forward ops return what the backward needs as extra outputs (saved in `setup_context`); auxiliary
# outputs never receive gradients (`set_materialize_grads(False)` -> None).  A `train` flag tells a forward op whether
# anything will be differentiated (it runs below autograd and cannot see `requires_grad`).
def _wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# RMSNorm -- LlamaRMSNorm.forward, models/llama/modeling_llama.py:62-67
def _rmsnorm_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])
    ctx.set_materialize_grads(False)


def _rmsnorm_backward(ctx, dy, _drstd):
    if dy is None:
        return None, None, None
    x, w, rstd = ctx.saved_tensors
    dx, dw = T.rmsnorm_bwd(dy, x, w, rstd)
    return dx, dw, None


register("rmsnorm", lambda x, w, eps: (torch.empty_like(x), _f32(x, _rows(x))), _rmsnorm_backward, _rmsnorm_setup)


# h = x + residual; y = RMSNorm(h): the residual add of LlamaDecoderLayer.forward (modeling_llama.py:317,323)
def _add_rmsnorm_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[2], output[2])
    ctx.set_materialize_grads(False)


def _add_rmsnorm_backward(ctx, dy, dh, _drstd):
    if dy is None:
        return dh, dh, None, None
    h, w, rstd = ctx.saved_tensors
    dx, dw = T.rmsnorm_bwd(dy, h, w, rstd, dh)
    return dx, dx, dw, None


register("add_rmsnorm", lambda x, residual, w, eps: (torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x))),
         _add_rmsnorm_backward, _add_rmsnorm_setup)


# LayerNorm -- call sites models/bert/modeling_bert.py:62,106; gpt2 :252-254; clip :358-360
def _layernorm_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1], output[2])
    ctx.has_b = inputs[2] is not None
    ctx.set_materialize_grads(False)


def _layernorm_backward(ctx, dy, _dm, _dr):
    if dy is None:
        return None, None, None, None
    x, w, mean, rstd = ctx.saved_tensors
    dx, dw, db, _ = T.layernorm_bwd(dy, x, w, mean, rstd, None, ctx.has_b, False)
    return dx, dw, (db if ctx.has_b else None), None


register("layernorm", lambda x, w, b, eps: (torch.empty_like(x), _f32(x, _rows(x)), _f32(x, _rows(x))),
         _layernorm_backward, _layernorm_setup)


# y = LayerNorm(x + residual) -> (y, h): BertSelfOutput / BertOutput (modeling_bert.py:289-293, :347-351)
def _add_layernorm_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[2], output[2], output[3])
    ctx.has_b = inputs[3] is not None
    ctx.set_materialize_grads(False)


def _add_layernorm_backward(ctx, dy, dh, _dm, _dr):
    if dy is None:
        return dh, dh, None, None, None
    h, w, mean, rstd = ctx.saved_tensors
    dx, dw, db, _ = T.layernorm_bwd(dy, h, w, mean, rstd, dh, ctx.has_b, False)
    return dx, dx, dw, (db if ctx.has_b else None), None


register("add_layernorm", lambda x, residual, w, b, eps: (torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x)),
                                                          _f32(x, _rows(x))), _add_layernorm_backward, _add_layernorm_setup)


# y = LayerNorm(dropout(x, p) + residual) -> (y, h): BertSelfOutput / BertOutput in train mode
# (modeling_bert.py:289-293, :347-351); the keep mask is regenerated from (seed, element index) in the backward
def _dropout_add_layernorm_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[2], output[2], output[3], inputs[7])  # (+ the device seed, if any)
    ctx.has_b = inputs[3] is not None
    ctx.drop = (inputs[5], inputs[6])
    ctx.set_materialize_grads(False)


def _dropout_add_layernorm_backward(ctx, dy, dh, _dm, _dr):
    none = (None,) * 8
    if dy is None:
        if dh is None:
            return none
        raise TamdError("dropout_add_layernorm: only the pre-norm sum is differentiated; use ops.layernorm pieces")
    h, w, mean, rstd, seed_dev = ctx.saved_tensors
    dx, dxd, dw, db, _ = T.layernorm_dropout_bwd(dy, h, w, mean, rstd, ctx.drop[0], ctx.drop[1], dh, ctx.has_b, False, seed_dev)
    return (dxd, dx, dw, (db if ctx.has_b else None)) + none[4:]


register("dropout_add_layernorm", lambda x, residual, w, b, eps, dropout_p, seed, seed_dev=None: (
    torch.empty_like(x), torch.empty_like(x), _f32(x, _rows(x)), _f32(x, _rows(x))),
    _dropout_add_layernorm_backward, _dropout_add_layernorm_setup)


# y = act(x W^T + b) [+ residual] on the MFMA GEMM; dX and dW use the k-major operand modes (no HBM transposes).
# nn.Linear call sites: see csrc/gemm.hip header.
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
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        dy2, db_act = T.bias_act_bwd(pre, None, dy2, ctx.act, want_db)   # (+ its column sums: the bias gradient)
    else:
        db_act = None
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = T.gemm(dy2, w, False, True).view(x.shape)                      # dX = dY . W
    if ctx.needs_input_grad[1]:
        dw = T.gemm(dy2, _c(x).view(-1, k), True, True)                     # dW = dY^T . X
    if ctx.has_bias and ctx.needs_input_grad[2]:
        db = db_act if db_act is not None else T.colsum(dy2)
    return dx, dw, db, dres, None, None


register("linear", lambda x, w, bias, residual, act, train: (
    x.new_empty(*x.shape[:-1], w.shape[0]),
    x.new_empty(_rows(x), w.shape[0]) if (train and act != ACT_NONE) else _nothing(x)), _linear_backward, _linear_setup)


# y = x . Wf^T (+ bf), Wf = row-concatenation of the member weights (fused QKV / gate|up, fused_params.py).
# Gradients go to the member parameters: one fused dW GEMM, each member receives its row slice.
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


register("fused_linear", lambda x, wf, bf, members: x.new_empty(*x.shape[:-1], wf.shape[0]), _fused_linear_backward,
         _fused_linear_setup)


# y = x @ W + b with W stored [in, out] (GPT-2 Conv1D, pytorch_utils.py:117-121): the k-major B operand
def _conv1d_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.has_bias = inputs[2] is not None


def _conv1d_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy2 = _c(dy).view(-1, w.shape[1])
    dx = T.gemm(dy2, w).view(x.shape)                                       # dX = dY . W^T  (W is [N=in, K=out])
    dw = T.gemm(_c(x).view(-1, x.shape[-1]), dy2, True, True)               # dW[in,out] = X^T . dY
    return dx, dw, (T.colsum(dy2) if ctx.has_bias else None)


register("conv1d", lambda x, w, b: x.new_empty(*x.shape[:-1], w.shape[1]), _conv1d_backward, _conv1d_setup)


# Rotary embedding on the first `nheads` heads of a [B, S, row] projection output
# (models/llama/modeling_llama.py:130-160).  Out of place at this level (autograd needs the input intact).
def _rope_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1], inputs[2])
    ctx.meta = inputs[3:]


def _rope_backward(ctx, dy):
    cos, sin = ctx.saved_tensors
    nheads, head_dim, conj = ctx.meta
    return T.rope(dy, cos, sin, nheads, head_dim, not conj), None, None, None, None, None


register("rope", lambda x, cos, sin, nheads, head_dim, conj=False: torch.empty_like(x, memory_format=torch.contiguous_format),
         _rope_backward, _rope_setup)


# softmax(scale QK^T + mask) V on [B,S,H,D] views.  Reference: eager_attention_forward,
# models/llama/modeling_llama.py:191-213 and siblings.
def _attention_setup(ctx, inputs, output):
    q, k, v, key_valid, scale, causal, dropout_p, seed, q_start, _train, seed_dev = inputs
    ctx.save_for_backward(q, k, v, output[0], output[1], key_valid, q_start, seed_dev)
    ctx.meta = (scale, causal, dropout_p, seed)
    ctx.set_materialize_grads(False)


def _attention_backward(ctx, do, _dlse):
    none = (None,) * 11
    if do is None:
        return none
    q, k, v, o, lse, key_valid, q_start, seed_dev = ctx.saved_tensors
    if lse.numel() == 0:
        raise TamdError("attention was run with train=False but is being differentiated")
    scale, causal, dropout_p, seed = ctx.meta
    dq, dk, dv = T.attn_bwd(q, k, v, o, lse, do, scale, causal, key_valid, dropout_p, seed, q_start, None, None, seed_dev)
    return (dq, dk, dv) + none[3:]


register("attention", lambda q, k, v, key_valid, scale, causal, dropout_p, seed, q_start, train, seed_dev=None: (
    q.new_empty(q.shape), _f32(q, q.shape[0], q.shape[2], q.shape[1]) if train else _nothing(q)),
    _attention_backward, _attention_setup)


# act = silu(gate) * up on a fused [T, 2I] projection output (modeling_llama.py:174-176)
def _swiglu_backward(ctx, dact):
    (gu,) = ctx.saved_tensors
    gu2 = _c(gu).view(-1, gu.shape[-1])
    dgu, _ = T.swiglu_bwd(gu2, _c(dact).view(gu2.shape[0], -1), False)
    return dgu.view(gu.shape)


register("swiglu", lambda gu: gu.new_empty(*gu.shape[:-1], gu.shape[-1] // 2), _swiglu_backward,
         lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


def _bias_act_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.act = inputs[2]


def _bias_act_backward(ctx, dy):
    x, bias = ctx.saved_tensors
    dx, db = T.bias_act_bwd(x, bias, dy, ctx.act, bias is not None)
    return dx, (db if bias is not None else None), None


register("bias_act", lambda x, bias, act: torch.empty_like(x), _bias_act_backward, _bias_act_setup)


# nn.Embedding (models/llama/modeling_llama.py:381): bit-exact gather, sorted scatter-add backward
def _embedding_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.meta = (inputs[1].shape[0], inputs[2])


def _embedding_backward(ctx, dout):
    (ids,) = ctx.saved_tensors
    vocab, padding_idx = ctx.meta
    return None, T.embedding_bwd(ids, dout, vocab, padding_idx), None


register("embedding", lambda ids, table, padding_idx=-1: table.new_empty(*ids.shape, table.shape[1]), _embedding_backward,
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
    d_pre, dw, db, _ = T.layernorm_bwd(dy, pre, ln_w, mean, rstd)
    d_word = T.embedding_bwd(input_ids, d_pre, vocab, padding_idx)
    d_typ = T.embedding_bwd(token_type_ids, d_pre, tvocab, -1)
    d_pos = T.embedding_bwd(position_ids, d_pre, npos, -1)
    return (None, None, None, d_word, d_typ, d_pos, dw, db, None, None, None)


register("bert_embeddings",
         lambda input_ids, token_type_ids, position_ids, word, typ, pos, ln_w, ln_b, eps, padding_idx, train: (
             word.new_empty(*input_ids.shape, word.shape[1]),
             word.new_empty(*input_ids.shape, word.shape[1]) if train else _nothing(word),
             _f32(word, input_ids.numel()), _f32(word, input_ids.numel())),
         _bert_embeddings_backward, _bert_embeddings_setup)


# fixed_cross_entropy on `logits.float()` (loss/loss_utils.py:32-46) without materialising fp32 logits.
# Returns the SUM of per-token losses; the caller divides (mean over valid labels or num_items_in_batch).
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


register("cross_entropy_sum", lambda logits2d, labels, ignore_index=-100: (_f32(logits2d), _f32(logits2d, logits2d.shape[0])),
         _cross_entropy_sum_backward, _cross_entropy_sum_setup)


# lm_head + causal-LM loss without ever holding the [tokens, vocab] logits (reference:
# `logits = self.lm_head(hidden)` then ForCausalLMLoss, modeling_llama.py:477-484 / loss/loss_utils.py:49-71).
# Tokens are processed in chunks: logits_c = h_c W^T (MFMA GEMM) -> cross-entropy forward (lse, per-token loss) ->
# dlogits_c, already scaled by 1/normaliser -> dh_c = dlogits_c W and dW += dlogits_c^T h_c (accumulate epilogue).
# The same three GEMMs as the unfused path, no recomputation; the gradients are produced in the forward and only
# multiplied by the upstream scalar in the backward.  Peak extra memory: one chunk of logits instead of 2 x [T, V].
# dW accumulates in the storage dtype across chunks (<= 8 roundings at the default chunking).
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


register("linear_cross_entropy", lambda h2d, w, labels, normaliser, ignore_index, chunk, need_dh, need_dw: (
    _f32(h2d), torch.empty_like(h2d) if need_dh else _nothing(h2d), torch.empty_like(w) if need_dw else _nothing(w)),
    _linear_cross_entropy_backward, _linear_cross_entropy_setup)


# Vocabulary projection whose width is not a multiple of 8 (+ optional token-level cross-entropy): BERT's MLM head,
# `self.decoder(hidden)` then `CrossEntropyLoss()(scores.view(-1, V), labels.view(-1))` (models/bert/modeling_bert.py:
# 483-496, 970-975; bert-base: V = 30522).  The GEMM stores 16-byte row segments, so the weight / bias live in buffers
# padded with ZERO rows to Vp = a multiple of 64 (fused_params.PaddedRows: the tied parameters are row-slice views of them),
# the logits are computed as [M, Vp] and handed out as the [M, V] view of that buffer (row stride Vp), and the backward runs
# the dX / dW GEMMs on the K-padded gradient buffer the loss kernel writes (padding columns zero).
#   -> (loss_sum fp32 scalar: SUM of the per-token losses, 0 when labels is None; logits [M, V] (strided); lse [M] fp32)
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
    need = (ctx.needs_input_grad[0], ctx.needs_input_grad[3], ctx.has_bias and ctx.needs_input_grad[4])
    dh, dw, db = T.padded_vocab_head_bwd(g_loss if labels is not None else None, g_logits, h2d, w_pad, labels, logits, lse,
                                         ctx.ignore_index, *need)
    return (dh if need[0] else None, None, None, dw if need[1] else None, db if need[2] else None) + none[5:]


register("padded_vocab_head", _padded_vocab_head_fake, _padded_vocab_head_backward, _padded_vocab_head_setup)
register("padded_vocab_head_bwd",
         lambda g_loss, g_logits, h2d, w_pad, labels, logits, lse, ignore_index, need_dh, need_dw, need_db: (
             torch.empty_like(h2d) if need_dh else _nothing(h2d),
             h2d.new_empty(logits.shape[1], h2d.shape[1]) if need_dw else _nothing(h2d),
             h2d.new_empty(logits.shape[1]) if need_db else _nothing(h2d)))


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
    """LayerNorm(dropout(x, p) + residual) -> y, the dropout inside the norm kernel (seed from torch's generators unless
    given -- `dropout_seeds`: `torch.manual_seed` repeats it, activation checkpointing regenerates it, a captured step draws
    a fresh one per replay)."""
    seed_dev = None
    if seed is None:
        (seed,), seed_dev = dropout_seeds(1, x.device)
    return T.dropout_add_layernorm(x, residual, w, b, float(eps), float(dropout_p), int(seed), seed_dev)[0]


def hidden_dropout_keep_mask(seed: int, rows: int, cols: int, p: float) -> torch.Tensor:
    """The keep mask [rows, cols] (bool) of the hidden-state dropout inside the add + LayerNorm kernels, rebuilt on the host:
    element (row, col) is kept iff tamd_dropout_hash(seed, row * cols + col) >= p * 2^32 (csrc/dropout.h DropCtx)."""
    import numpy as np

    idx = np.arange(rows * cols, dtype=np.uint64)
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
    return torch.from_numpy((x >= thr).reshape(rows, cols))


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


def rope_inplace(x2d, cos, sin, seq, nheads, head_dim):
    """Inference only: the rotary embedding applied IN PLACE to the first `nheads` heads of every row of x2d [tokens, row]
    (a fresh q|k|v product nobody else holds: the cached forward, models/llama.py)."""
    T.rope_(x2d, cos, sin, int(seq), int(nheads), int(head_dim), False)
    return x2d


def linear_swiglu(x2, wgu):
    """Inference only: SiLU(x2 . Wg^T) * (x2 . Wu^T) for wgu = [Wg ; Wu] from ONE product (`gemm_swiglu_supported` shapes: the
    GEMM epilogue, or the weight-streaming kernel at M <= 16), gate | up themselves not kept."""
    return T.gemm_swiglu(x2, wgu, False)[1]


def bias_act(x, bias, act):
    return T.bias_act(x, bias, int(act))


def dropout_seed() -> int:
    """One 63-bit seed per attention call from torch's CPU generator: `torch.manual_seed` makes runs repeatable,
    and activation checkpointing (which restores the CPU RNG state before recomputing) regenerates the same mask."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def dropout_seed_tensor(n: int, device) -> torch.Tensor:
    """`n` 62-bit seeds as an int64 DEVICE tensor, drawn by torch's own RNG kernel on the current stream: no host round trip,
    and -- the point -- safe under HIP-graph capture: the generator's philox offset is part of the captured graph's state and
    advances on every replay, so every replay of a captured training step draws fresh masks (a host-drawn seed would be baked
    into the graph).  `torch.manual_seed` / `torch.cuda.manual_seed` repeat it; activation checkpointing restores the device
    RNG state before recomputing."""
    return torch.empty(n, dtype=torch.int64, device=device).random_(0, 2 ** 62)


def capturing(device) -> bool:
    """Is the current stream of `device` being captured into a HIP graph?"""
    dev = torch.device(device)
    return dev.type == "cuda" and torch.cuda.is_current_stream_capturing()


def dropout_seeds(n: int, device):
    """-> (host seeds [n ints], device seed tensor or None) for `n` dropout sites of one op.  Eagerly the seeds are host
    integers (no extra launch); while the current stream is being captured into a graph they live on the device."""
    if capturing(device):
        return [0] * n, dropout_seed_tensor(n, torch.device(device))
    return [dropout_seed() for _ in range(n)], None


def dropout_keep_mask(seed: int, batch: int, heads: int, seq_q: int, seq_k: int, p: float) -> torch.Tensor:
    """The attention kernels' keep mask [B,H,Sq,Sk] (bool), rebuilt on the host (csrc/dropout.h: the seed is mixed once by
    splitmix64; one 64-bit mix of 24-bit multiplies decides a 2 x 2 block -- query pair x key pair -- of the probability
    matrix, 16 bits per element; include/tamd.h tamd_attn_dropout_field is the same function element by element) -- for
    tests/debugging."""
    import numpy as np

    m64 = (1 << 64) - 1
    z = int(seed) & m64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m64
    z ^= z >> 31
    s0, s1 = np.uint32(z & 0xFFFFFFFF), np.uint32(z >> 32)
    csq, csk = (seq_q + 1) // 2, (seq_k + 1) // 2
    bh = np.arange(batch * heads, dtype=np.uint64)[:, None, None]
    qb = (np.arange(seq_q, dtype=np.uint64) >> np.uint64(1))[None, :, None]
    kb = (np.arange(seq_k, dtype=np.uint64) >> np.uint64(1))[None, None, :]
    idx = (bh * np.uint64(csq) + qb) * np.uint64(csk) + kb
    lo, hi = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)

    def mul24(a, k24):  # v_mul_u32_u24
        return (a & np.uint32(0xFFFFFF)) * np.uint32(k24)

    with np.errstate(over="ignore"):
        x = lo ^ s0
        x = mul24(x, 0x9E3779) + mul24(x >> np.uint32(8), 0x85EBCB) + (s1 + mul24(hi, 0x632BE5))
        x ^= x >> np.uint32(15)
        x = mul24(x, 0xC2B2AF) + mul24(x >> np.uint32(8), 0x27D4EB)
        x ^= x >> np.uint32(13)
        t = x ^ np.uint32(0x165667B1)
        y = mul24(t, 0xD3A265) + mul24(t >> np.uint32(8), 0x7F4A7D)  # the second word (odd keys)
        y ^= y >> np.uint32(16)
    k_odd = (np.arange(seq_k) & 1).astype(bool)[None, None, :]
    q_odd = (np.arange(seq_q) & 1).astype(bool)[None, :, None]
    w = np.where(k_odd, y, x)
    field = np.where(q_odd, w >> np.uint32(16), w & np.uint32(0xFFFF))
    thr16 = np.uint32(min(65535.0, float(np.float32(p)) * 65536.0))
    return torch.from_numpy((field >= thr16).reshape(batch, heads, seq_q, seq_k))


def attention(q, k, v, scale, causal, key_valid=None, dropout_p=0.0, seed=None, q_start=None, seed_dev=None):
    if dropout_p > 0.0 and seed is None and seed_dev is None:
        (seed,), seed_dev = dropout_seeds(1, q.device)
    return T.attention(q, k, v, key_valid, float(scale), bool(causal), float(dropout_p), int(seed or 0), q_start,
                       _wants_grad(q, k, v), seed_dev)[0]


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


def sliding_window_q_start(batch: int, seq: int, window: int, device) -> torch.Tensor:
    """The reference's sliding-window overlay on a causal mask (`kv_idx > q_idx - sliding_window`, masking_utils.py:92-101,
    134-138) as the kernels' two bound planes, int32 [2, B, S]: query q sees keys max(0, q - window + 1) .. q, key k is seen
    by queries k .. min(S - 1, k + window - 1).  Both planes are non-decreasing, which is all the kernels ask of them
    (include/tamd.h q_start): tiles left of the window are neither loaded nor visited."""
    if window < 1:
        raise TamdError(f"sliding_window must be >= 1, got {window}")
    pos = torch.arange(seq, device=device, dtype=torch.int64)
    planes = torch.stack(((pos - (window - 1)).clamp(min=0), (pos + (window - 1)).clamp(max=seq - 1)))
    return planes[:, None, :].expand(2, batch, seq).to(torch.int32).contiguous()


def chunked_q_start(batch: int, seq: int, chunk_size: int, left_padding, device) -> torch.Tensor:
    """The reference's chunked-attention overlay on a causal mask (`(kv_idx - left_padding[b]) // chunk_size ==
    (q_idx - left_padding[b]) // chunk_size`, masking_utils.py:104-113, 161-165): a packed batch whose sequence ids are the
    chunk numbers (floor division, so the left padding forms chunks of its own as in the reference)."""
    if chunk_size < 1:
        raise TamdError(f"chunk_size must be >= 1, got {chunk_size}")
    pos = torch.arange(seq, device=device, dtype=torch.int64)[None, :]
    lp = (torch.zeros(batch, dtype=torch.int64, device=device) if left_padding is None
          else torch.as_tensor(left_padding, device=device).to(torch.int64).reshape(-1))
    if lp.numel() == 1 and batch != 1:
        lp = lp.expand(batch)
    if lp.numel() != batch:
        raise TamdError(f"chunked attention: left_padding has {lp.numel()} entries for a batch of {batch}")
    return packed_q_start(torch.div(pos - lp[:, None], chunk_size, rounding_mode="floor"))


def intersect_q_start(a: Optional[torch.Tensor], b: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Bound planes of the AND of two masks that each have the form `q_start[q] <= k <= q` (packed sequences, a sliding
    window, chunks): the later first-visible key per query, the earlier last-seeing query per key."""
    if a is None or b is None:
        return a if b is None else b
    return torch.stack((torch.maximum(a[0], b[0]), torch.minimum(a[1], b[1]))).contiguous()


def q_start_from_cu_seqlens(cu_seq_lens: torch.Tensor, total: int) -> torch.Tensor:
    """The reference's varlen description of a flattened batch -- `cu_seq_lens_q` = cumulative sequence lengths [n + 1]
    (`FlashAttentionKwargs`, modeling_flash_attention_utils.py:575-590; DataCollatorWithFlattening) -- as the two bound
    planes the kernels take: int32 [2, 1, total] (first / last token of each token's sequence).  Tokens past the last
    boundary (padding of the flattened row) form one more sequence.  Device-side, no synchronisation."""
    cu = cu_seq_lens.to(torch.int64).reshape(-1)
    idx = torch.arange(total, device=cu.device)
    ends = torch.cat([cu[1:], cu.new_tensor([total])])            # exclusive end of every sequence (+ the padding tail)
    starts = torch.cat([cu[:-1], cu[-1:].clamp(max=total)])
    seg = torch.searchsorted(ends, idx, right=True).clamp(max=ends.numel() - 1)
    start, end = starts[seg], ends[seg] - 1
    return torch.stack((start, end)).to(torch.int32).view(2, 1, total).contiguous()


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
