"""`attn_implementation="tamd"`: the MI355X flash-attention kernel behind the reference's AttentionInterface.

Boundary (SURVEY.md §8b "B1"): `AttentionInterface.register(key, fn)` (src/transformers/modeling_utils.py:5092-5130)
and `AttentionMaskInterface.register(key, mask_fn)` (src/transformers/masking_utils.py:711-724).  The callee
signature is the documented one (docs/source/en/attention_interface.md:156-182; call sites
models/llama/modeling_llama.py:268-277, models/bert/modeling_bert.py:192-201, models/gpt2/modeling_gpt2.py:211-220,
models/clip/modeling_clip.py:321-330):

    fn(module, query[B,Hq,S,D], key[B,Hkv,Sk,D], value[B,Hkv,Sk,D], attention_mask, *, dropout, scaling, **kwargs)
        -> (attn_output[B,S,Hq,D] contiguous, None)
"""
from __future__ import annotations

import threading
from typing import Optional

import torch

from . import ops
from ._cabi import TamdError

ATTN_KEY = "tamd"


class TamdMask:
    """What `tamd_mask` hands to the attention function for packed sequences: the 2-D key-validity mask (or None) plus
    `q_start` (int32 [2, B, S]: first token of every query's sequence, last token of every key's sequence).  The
    decoder layers pass it along untouched as `attention_mask`."""

    __slots__ = ("key_valid", "q_start")

    def __init__(self, key_valid, q_start):
        self.key_valid, self.q_start = key_valid, q_start


def _mask_kv_len(attention_mask) -> Optional[int]:
    """Number of key slots a mask produced by `tamd_mask` covers.  For a prefill into a pre-allocated (static) KV cache
    `tamd_mask` returns a key-validity mask over the slots IN USE only -- shorter than the cache's key/value tensors --
    and the attention function slices K/V to it, so the kernels' bottom-right causal alignment is the reference's
    `kv_idx <= q_idx + q_offset`.  The length travels as the mask's own shape (it survives `.to(device)`, slicing of the
    batch, fake tensors and graph capture; round 2 carried it as a Python attribute on the tensor, which none of those
    keep)."""
    if isinstance(attention_mask, TamdMask):
        attention_mask = attention_mask.key_valid
    if attention_mask is None or not torch.is_tensor(attention_mask):
        return None
    return int(attention_mask.shape[-1])


def _closure_vars(fn):
    cells = getattr(fn, "__closure__", None) or ()
    return dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in cells))) if cells else {}


def _decompose_mask_function(fn):
    """Leaves of an `and_masks(...)` tree (masking_utils.py:48-59)."""
    inner = _closure_vars(fn).get("mask_functions") if getattr(fn, "__name__", "") == "and_mask" else None
    if inner is None:
        return [fn]
    out = []
    for f in inner:
        out.extend(_decompose_mask_function(f))
    return out


def tamd_mask(batch_size, q_length, kv_length, q_offset=0, kv_offset=0, mask_function=None, attention_mask=None,
              **kwargs):
    """Mask factory for AttentionMaskInterface: the kernels take a [B, kv_len] key-validity mask (or None) and, for
    packed sequences, the first visible key of every query.

    Same contract as `flash_attention_mask` (masking_utils.py:607-647) minus its host-side `.all()` sync: causality
    is a kernel flag, padding is the 2-D mask itself.  The reference describes everything else through
    `mask_function`: plain causal / bidirectional are the kernel flag; `and_masks(causal, packed_sequence_mask)`
    (masking_utils.py:973-974, position_ids restarting inside a row), the causal sliding window (:134-138) and chunked
    attention (:161-165) -- alone or AND-ed together -- become `q_start` (every one of them is `first[q] <= k <= q`); any
    other mask function (bidirectional windows, blockwise, user overlays) is refused instead of being silently ignored."""
    from transformers import masking_utils as mu

    # 2-D padding mask over the kv window [kv_offset, kv_offset + kv_length), padded with zeros on the right when it
    # is shorter than the window (pre-allocated caches), as prepare_padding_mask does (masking_utils.py:205-215)
    padding = attention_mask
    if padding is not None:
        if padding.dim() != 2:
            raise TamdError(f"attn_implementation='tamd' takes a 2-D attention_mask, got {tuple(padding.shape)}")
        if (short := kv_length + kv_offset - padding.shape[-1]) > 0:
            padding = torch.nn.functional.pad(padding, (0, short))
        padding = padding[:, kv_offset: kv_offset + kv_length] if padding.shape[-1] != kv_length else padding
    bidirectional = getattr(mu, "bidirectional_mask_function", None)
    if mask_function is not None and mask_function is bidirectional:
        # encoder self-attention and CROSS-attention (create_bidirectional_mask with encoder_hidden_states: q_offset = 0,
        # kv_length = encoder length != q_length, masking_utils.py:1000-1080): every query sees every valid key; there is
        # no cache geometry to decode
        return padding
    kv_len = None
    dynamic = not torch.is_tensor(q_offset) and kv_offset == 0 and kv_length == q_offset + q_length
    if not dynamic:
        # a cache whose key/value tensors are longer than what has been written (StaticCache: kv_length =
        # max_cache_len, q_offset = tokens already cached, cache_utils.py:489-497).  Visible keys are
        # kv_idx <= q_idx + q_offset (masking_utils.py:76-81 with the offsets of :193-202).
        if kv_offset != 0:
            raise TamdError("attn_implementation='tamd' does not implement sliding-window KV caches (kv_offset != 0); "
                            "use attn_implementation='sdpa' or 'eager'")
        if q_length == 1:  # decode: one query sees keys 0 .. q_offset; stays on the device (q_offset may be a tensor)
            dev = padding.device if padding is not None else kwargs.get("device")
            used = torch.arange(kv_length, device=dev)[None] <= torch.as_tensor(q_offset, device=dev)
            padding = used.expand(batch_size, -1) if padding is None else (padding.to(torch.bool) & used)
        else:              # prefill into the cache: slice K/V to the slots in use (host integer: one sync per forward)
            kv_len = int(q_offset) + q_length
            if kv_len > kv_length:
                raise TamdError(f"KV cache overflow: {kv_len} positions, cache holds {kv_length}")
            dev = padding.device if padding is not None else kwargs.get("device")
            padding = (torch.ones(batch_size, kv_len, dtype=torch.bool, device=dev) if padding is None
                       else padding[:, :kv_len].to(torch.bool))
    plain = (None, mu.causal_mask_function, getattr(mu, "bidirectional_mask_function", None))
    if mask_function in plain:
        return padding  # (a static-cache prefill mask is [B, kv_len]: shorter than the cache, see _mask_kv_len)
    packed_ids, window, chunk = None, None, None
    for leaf in _decompose_mask_function(mask_function):
        if leaf in plain:
            continue
        kind, cv = _overlay_kind(leaf), _closure_vars(leaf)
        if kind == "packed" and torch.is_tensor(cv.get("packed_sequence_mask")) and packed_ids is None:
            packed_ids = cv["packed_sequence_mask"]
            continue
        if kind == "sliding" and isinstance(cv.get("sliding_window"), int) and not isinstance(cv["sliding_window"], bool):
            window = cv["sliding_window"] if window is None else min(window, cv["sliding_window"])
            continue
        if kind == "chunked" and isinstance(cv.get("chunk_size"), int) and chunk is None:
            chunk = (cv["chunk_size"], cv.get("left_padding"))
            continue
        raise TamdError("attn_implementation='tamd' supports causal / bidirectional masks, 2-D padding, packed sequences and "
                        "the causal sliding-window / chunked overlays; the mask function "
                        f"{getattr(leaf, '__qualname__', leaf)!r} is not one of them "
                        "(bidirectional windows / blockwise / custom overlays need attn_implementation='sdpa' or 'eager')")
    if packed_ids is None and window is None and chunk is None:
        return padding
    if not dynamic or q_offset != 0 or q_length != kv_length:
        raise TamdError("packed sequences, sliding windows and chunked attention with a KV cache are not supported by "
                        "attn_implementation='tamd' (use 'sdpa' or 'eager' for cached generation with these masks)")
    # every one of these overlays AND-ed with the causal mask has the form `q_start[b, q] <= k <= q`: the kernels' two bound planes
    dev = (packed_ids.device if packed_ids is not None else padding.device if padding is not None
           else kwargs.get("device") or "cpu")
    bounds = None if packed_ids is None else ops.packed_q_start(packed_ids[:, -q_length:])
    if window is not None and window < kv_length:  # (a window as long as the row is the plain causal mask)
        bounds = ops.intersect_q_start(bounds, ops.sliding_window_q_start(batch_size, q_length, window, dev))
    if chunk is not None:
        bounds = ops.intersect_q_start(bounds, ops.chunked_q_start(batch_size, q_length, chunk[0], chunk[1], dev))
    return padding if bounds is None else TamdMask(padding, bounds)


def _overlay_kind(leaf) -> Optional[str]:
    """Which of the reference's mask overlays a leaf of the `and_masks` tree is, by the factory that made it
    (masking_utils.py:92-101 sliding_window_overlay, :104-113 chunked_overlay, :182-190 packed_sequence_mask_function)."""
    if getattr(leaf, "__name__", "") != "inner_mask":
        return None
    factory = getattr(leaf, "__qualname__", "").split(".<locals>")[0]
    return {"packed_sequence_mask_function": "packed", "sliding_window_overlay": "sliding",
            "chunked_overlay": "chunked"}.get(factory)


def split_mask(attention_mask, batch: int, kv_len: int):
    """(key_valid [B, kv_len] bool or None, q_start int32 [2, B, S] or None) from whatever reached the attention layer."""
    if isinstance(attention_mask, TamdMask):
        kv = None if attention_mask.key_valid is None else _key_valid_from_mask(attention_mask.key_valid, batch, kv_len)
        return kv, attention_mask.q_start
    if attention_mask is None:
        return None, None
    return _key_valid_from_mask(attention_mask, batch, kv_len), None


_varlen_cache = threading.local()  # the last (cu_seq_lens_q, cu_seq_lens_k, total) -> q_start of this thread


def _ver(t) -> Optional[int]:
    """The version counter of a tensor, or None when it has none: tensors created under `torch.inference_mode()` do not track
    versions (`t._version` raises), so an in-place update of one cannot be seen -- the per-forward caches below skip them."""
    return None if (t is None or t.is_inference()) else t._version


def _same_tensor(a, b) -> bool:
    """Same bytes by construction: the same object, or two views of one storage at the same version (the caller holds a
    reference to `a`, so its storage cannot have been recycled for `b`).  Inference tensors (no version counter) only match
    as the same object -- and the callers do not cache them at all."""
    if a is b:
        return True
    if a is None or b is None or _ver(a) is None or _ver(b) is None:
        return False
    return a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype and _ver(a) == _ver(b)


def varlen_q_start(q_start, kwargs, batch: int, sq: int, sk: int, causal: bool):
    """The reference's varlen kwargs (`cu_seq_lens_q / cu_seq_lens_k / max_length_q / max_length_k`,
    modeling_flash_attention_utils.py:575-590, consumed by its flash path :768-790) for a flattened batch: the same
    block-diagonal causal attention the mask factory derives from restarting `position_ids` -- used when the mask did not
    already carry it.  Different query / key boundaries (a KV cache under packing) are refused, not ignored.
    Every layer of a forward receives the same two tensors: the boundaries are validated (one comparison on the device,
    the only host synchronisation of this path) and turned into `q_start` ONCE per forward, the other layers reuse it."""
    cu_q = kwargs.get("cu_seq_lens_q")
    if q_start is not None or cu_q is None:
        return q_start
    cu_k = kwargs.get("cu_seq_lens_k")
    if batch != 1 or sq != sk or not causal:
        raise TamdError("attn_implementation='tamd' takes cu_seq_lens_q/k for one flattened, causal row without a KV cache "
                        "(equal query and key boundaries)")
    hit = getattr(_varlen_cache, "entry", None)
    vers = (_ver(cu_q), _ver(cu_k) if cu_k is not None else -1)
    cacheable = None not in vers  # (inference tensors carry no version: validated and converted on every call)
    if (cacheable and hit is not None and hit[2] == sq and _same_tensor(hit[0], cu_q)
            and (cu_k is None or _same_tensor(hit[1], cu_k)) and hit[4] == vers):  # (the SAME object updated in place)
        return hit[3]
    if cu_k is not None and cu_k is not cu_q and (cu_k.shape != cu_q.shape or not torch.equal(cu_k, cu_q)):
        raise TamdError("attn_implementation='tamd' takes cu_seq_lens_q/k for one flattened, causal row without a KV cache "
                        "(equal query and key boundaries)")
    qs = ops.q_start_from_cu_seqlens(cu_q, sq)
    if cacheable:
        _varlen_cache.entry = (cu_q, cu_k if cu_k is not None else cu_q, sq, qs, vers)
    return qs


_window_cache = threading.local()  # the last (batch, seq, window, device) -> q_start of this thread


def window_q_start(q_start, sliding_window, batch: int, sq: int, sk: int, causal: bool, device):
    """The `sliding_window` argument the reference's sliding-window models hand to every attention function
    (e.g. models/mistral/modeling_mistral.py `sliding_window=getattr(self.config, "sliding_window", None)`; its flash path
    consumes it in modeling_flash_attention_utils.py `_process_flash_attention_kwargs`) when the mask did not already carry
    the window (a mask built by another factory, or no mask at all).  Decode steps (one query) see the whole cropped cache;
    a prefill on top of a cache longer than the window is refused, not computed with the wrong mask."""
    if sliding_window is None or q_start is not None or sq == 1 or int(sliding_window) >= sk:
        return q_start
    if not causal or sq != sk:
        raise TamdError("attn_implementation='tamd' implements sliding_window for causal attention without a KV cache "
                        f"(got {sq} queries over {sk} keys, causal={causal}); use 'sdpa' or 'eager'")
    key = (batch, sq, int(sliding_window), torch.device(device))
    hit = getattr(_window_cache, "entry", None)
    if hit is None or hit[0] != key:
        hit = _window_cache.entry = (key, ops.sliding_window_q_start(batch, sq, int(sliding_window), device))
    return hit[1]


_kv_cache = threading.local()  # the last (mask tensor, batch, kv_len) -> key_valid of this thread


def _key_valid_from_mask(attention_mask, batch: int, kv_len: int) -> Optional[torch.Tensor]:
    """[B, kv_len] bool, contiguous.  Every layer of a forward receives the same mask tensor: it is converted ONCE per forward
    (two small launches per layer otherwise -- 64 per generated token of a 32-layer model, where a decode step is bound by
    launches).  The entry keeps the mask tensor alive, so `_same_tensor` cannot be fooled by a recycled allocation, and an
    in-place update of the mask changes its version."""
    src = attention_mask.key_valid if isinstance(attention_mask, TamdMask) else attention_mask
    hit = getattr(_kv_cache, "entry", None)
    ver = _ver(src) if torch.is_tensor(src) else None  # (None: an inference tensor -- converted on every call, never cached)
    if (hit is not None and ver is not None and hit[1] == batch and hit[2] == kv_len and _same_tensor(hit[0], src)
            and hit[4] == ver):
        if isinstance(attention_mask, TamdMask) and attention_mask.q_start is not None:
            raise TamdError("packed sequences reached a fused block that does not implement them (supported: the "
                            "Llama path and every model that goes through the registered attention function)")
        return hit[3]
    out = _key_valid_from_mask_uncached(attention_mask, batch, kv_len)
    if ver is not None and out is not None:
        _kv_cache.entry = (src, batch, kv_len, out, ver)
    return out


def _key_valid_from_mask_uncached(attention_mask, batch: int, kv_len: int) -> Optional[torch.Tensor]:
    if isinstance(attention_mask, TamdMask):  # a fused module path that only knows padding masks
        if attention_mask.q_start is not None:
            raise TamdError("packed sequences reached a fused block that does not implement them (supported: the "
                            "Llama path and every model that goes through the registered attention function)")
        attention_mask = attention_mask.key_valid
        if attention_mask is None:
            return None
    if attention_mask.dim() == 2:
        kv = attention_mask
    elif attention_mask.dim() == 4 and attention_mask.shape[1] == 1 and attention_mask.shape[2] == 1:
        kv = attention_mask[:, 0, 0, :]  # [B,1,1,Sk] broadcast padding mask
    else:
        raise TamdError(
            "attn_implementation='tamd' takes a 2-D padding mask [batch, kv_len] (register-time mask function "
            f"`tamd_mask`); got a mask of shape {tuple(attention_mask.shape)}. Arbitrary 4-D masks are not supported.")
    if kv.dtype.is_floating_point:
        kv = kv > (torch.finfo(kv.dtype).min / 2)  # additive float mask: 0 keeps, finfo.min / -inf drops
    elif kv.dtype != torch.bool:
        kv = kv != 0
    if kv.shape[0] != batch:
        kv = kv.expand(batch, -1)
    return kv[:, -kv_len:].contiguous()


def tamd_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0,
                           scaling: Optional[float] = None, is_causal: Optional[bool] = None, **kwargs):
    # reference: nn.functional.dropout(attn_weights, p=dropout, training=module.training) after the softmax
    # (modeling_llama.py:209); here the keep mask is generated inside the kernels (ops.dropout_keep_mask).
    drop_p = float(dropout) if (dropout and getattr(module, "training", True)) else 0.0
    if not 0.0 <= drop_p < 1.0:
        raise TamdError(f"attention dropout must be in [0, 1), got {drop_p}")
    if not query.is_cuda and ops.backend().name == "hip":
        raise TamdError("attn_implementation='tamd' needs GPU tensors (no CPU fallback); use 'eager' or 'sdpa' on CPU")
    b, hq, sq, d = query.shape
    sk = key.shape[2]
    if d not in (64, 128):
        raise TamdError(f"attn_implementation='tamd' supports head_dim 64 and 128, got {d}")
    if query.dtype == torch.float32 and torch.is_autocast_enabled():
        # mixed precision with fp32 master weights (Trainer(bf16=True)): the projections ran in the autocast dtype but
        # cos/sin are fp32, so q/k arrive in fp32 after the rotary -- autocast would cast them for torch's own attention
        adt = torch.get_autocast_dtype("cuda")
        query, key, value = query.to(adt), key.to(adt), value.to(adt)
    if query.dtype not in (torch.bfloat16, torch.float16):
        raise TamdError(f"attn_implementation='tamd' supports bf16/fp16 (or fp32 under autocast), got {query.dtype}: "
                        "load the model with dtype=torch.bfloat16 or select attn_implementation='sdpa'")
    if key.dtype != query.dtype or value.dtype != query.dtype:
        key, value = key.to(query.dtype), value.to(query.dtype)
    if scaling is None:
        scaling = d ** -0.5
    causal = is_causal if is_causal is not None else getattr(module, "is_causal", True)
    causal = bool(causal) and sq > 1
    # [B,H,S,D] -> [B,S,H,D] views (the projections produced [B,S,H,D]; this undoes the caller's transpose)
    q, k, v = query.transpose(1, 2), key.transpose(1, 2), value.transpose(1, 2)
    used = _mask_kv_len(attention_mask)
    if used is not None and used < sk:
        sk = used  # pre-allocated cache: only the first kv_len key slots are in use (strided views of the cache)
        k, v = k[:, :sk], v[:, :sk]
    key_valid, q_start = split_mask(attention_mask, b, sk)
    q_start = varlen_q_start(q_start, kwargs, b, sq, sk, causal)
    q_start = window_q_start(q_start, kwargs.get("sliding_window"), b, sq, sk, causal, q.device)
    if q.stride(3) != 1:
        q = q.contiguous()
    if k.stride(3) != 1:
        k = k.contiguous()
    if v.stride(3) != 1:
        v = v.contiguous()
    out = ops.attention(q, k, v, float(scaling), causal, key_valid, dropout_p=drop_p, q_start=q_start)
    return out, None


_registered = False


def register() -> None:
    """Register the backend with the reference's two global registries (idempotent)."""
    global _registered
    if _registered:
        return
    from transformers.masking_utils import AttentionMaskInterface
    from transformers.modeling_utils import AttentionInterface

    AttentionInterface.register(ATTN_KEY, tamd_attention_forward)
    AttentionMaskInterface.register(ATTN_KEY, tamd_mask)
    _registered = True
