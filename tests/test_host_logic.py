"""Property tests of the host-side logic (no kernels): packed-sequence bounds, the split-K policy mirror, the mask
function decoder.  Randomised with hypothesis; the oracle's restatements are the reference."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
import oracle as orc  # noqa: E402

from transformers_amd import _cabi, build, ops  # noqa: E402
from transformers_amd.attention import TamdMask, tamd_mask  # noqa: E402


@settings(max_examples=60, deadline=None)
@given(st.lists(st.lists(st.integers(1, 9), min_size=1, max_size=6), min_size=1, max_size=3), st.integers(0, 3))
def test_packed_bounds_match_the_oracle(rows, seed):
    """ops.packed_q_start (device-side cummax/cummin construction) == the oracle's run scan, for arbitrary packings,
    and both describe the reference's and_masks(causal, packed) mask."""
    s = max(sum(r) for r in rows)
    pos = []
    for r in rows:
        r = list(r)
        r[-1] += s - sum(r)  # pad the last sequence so every row has length s
        pos.append(np.concatenate([np.arange(n) for n in r]))
    pos = np.stack(pos)
    ids = orc.packed_sequence_ids(pos)
    lo_hi = ops.packed_q_start(torch.from_numpy(ids)).numpy()
    assert np.array_equal(lo_hi, orc.packed_bounds(ids))
    mask = orc.packed_attention_mask_bool(ids)[:, 0]
    qi, ki = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    for b in range(len(rows)):
        assert np.array_equal((lo_hi[0, b][:, None] <= ki) & (ki <= qi), mask[b])
        assert np.array_equal((ki <= qi) & (qi <= lo_hi[1, b][None, :]), mask[b])


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 40000), st.integers(1, 5000).map(lambda n: n * 4), st.integers(1, 600).map(lambda k: k * 8),
       st.sampled_from([ops.EPI_NONE, ops.EPI_ACCUM, ops.EPI_BIAS, ops.EPI_RESIDUAL]))
def test_split_k_policy_mirror(m, n, k, epi):
    lib = _cabi.TamdLib(build.build())
    assert ops.gemm_workspace_bytes(m, n, k, epi) == lib.tamd_gemm_workspace_bytes(m, n, k, 0, epi)
    assert ops.gemm_workspace_bytes(m, n, k, epi, 2) == lib.tamd_gemm_workspace_bytes(m, n, k, 2, epi)


def test_mask_function_decoder():
    from transformers import masking_utils as mu

    am = torch.ones(2, 12, dtype=torch.long)
    am[0, 9:] = 0
    # plain causal / bidirectional: the padding mask itself (or None)
    assert tamd_mask(2, 12, 12, mask_function=mu.causal_mask_function) is None
    assert torch.equal(tamd_mask(2, 12, 12, mask_function=mu.causal_mask_function, attention_mask=am), am)
    if hasattr(mu, "bidirectional_mask_function"):
        assert tamd_mask(2, 12, 12, mask_function=mu.bidirectional_mask_function) is None
    # packed: decoded into TamdMask with the same bounds as the oracle
    pos = torch.tensor([[0, 1, 2, 3, 0, 1, 0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]])
    ids = mu.find_packed_sequence_indices(pos)
    fn = mu.and_masks(mu.causal_mask_function, mu.packed_sequence_mask_function(ids))
    m = tamd_mask(2, 12, 12, mask_function=fn, attention_mask=am)
    assert isinstance(m, TamdMask) and torch.equal(m.key_valid, am)
    assert np.array_equal(m.q_start.numpy(), orc.packed_bounds(ids.numpy()))
    # nested and_masks, order irrelevant
    fn2 = mu.and_masks(mu.and_masks(mu.packed_sequence_mask_function(ids)), mu.causal_mask_function)
    assert np.array_equal(tamd_mask(2, 12, 12, mask_function=fn2).q_start.numpy(), m.q_start.numpy())
    # anything else is refused, as is packing with a KV cache
    with pytest.raises(ops.TamdError):
        tamd_mask(2, 12, 12, mask_function=mu.and_masks(mu.causal_mask_function, mu.sliding_window_bidirectional_overlay(4)))
    with pytest.raises(ops.TamdError):
        tamd_mask(2, 12, 12, mask_function=lambda b, h, q, k: q >= k)
    with pytest.raises(ops.TamdError):
        tamd_mask(2, 4, 12, q_offset=8, mask_function=fn)


def _reference_bool_mask(fn, batch, s):
    """The reference's own evaluation of a mask function: [B, S, S] bool (masking_utils.py sdpa_mask, no skips)."""
    from transformers import masking_utils as mu

    m = mu.sdpa_mask(batch, s, s, 0, 0, mask_function=fn, attention_mask=None, allow_is_causal_skip=False)
    return m[:, 0].numpy()


def _bounds_as_bool(planes, s):
    """The mask the kernels compute from the two planes -- both forms (the forward / dQ kernels use plane 0, dK/dV plane 1)."""
    qi, ki = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    a = (planes[0][:, :, None] <= ki[None]) & (ki <= qi)[None]
    b = (ki <= qi)[None] & (qi[None] <= planes[1][:, None, :])
    return a, b


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 40), st.integers(1, 48), st.integers(1, 3), st.integers(0, 2), st.integers(1, 17), st.integers(0, 9))
def test_sliding_window_and_chunked_overlays_become_bounds(s, window, batch, mode, chunk, lp0):
    """masking_utils.py:92-113, 134-138, 161-165: the causal sliding-window and chunked masks (alone, together, and AND-ed
    with packed sequences) decoded by `tamd_mask` into the kernels' two bound planes describe exactly the mask the
    reference evaluates from the same mask function; both planes are non-decreasing (what the kernels' tile skipping
    relies on)."""
    from transformers import masking_utils as mu

    left = torch.tensor([(lp0 + 3 * i) % max(s, 1) for i in range(batch)])
    if mode == 0:
        fn = mu.sliding_window_causal_mask_function(window)
    elif mode == 1:
        fn = mu.chunked_causal_mask_function(chunk, left)
    else:
        ids = torch.div(torch.arange(s)[None].expand(batch, -1) + torch.arange(batch)[:, None], max(1, s // 3),
                        rounding_mode="floor")
        fn = mu.and_masks(mu.sliding_window_causal_mask_function(window), mu.chunked_overlay(chunk, left),
                          mu.packed_sequence_mask_function(ids))
    want = _reference_bool_mask(fn, batch, s)
    m = tamd_mask(batch, s, s, mask_function=fn, device="cpu")
    if m is None:  # a window that covers the row: the plain causal mask
        assert mode == 0 and window >= s
        assert np.array_equal(want, np.broadcast_to(np.tril(np.ones((s, s), bool)), want.shape))
        return
    assert isinstance(m, TamdMask) and m.key_valid is None and m.q_start.dtype == torch.int32
    planes = m.q_start.numpy()
    assert planes.shape == (2, batch, s)
    a, b = _bounds_as_bool(planes, s)
    assert np.array_equal(a, want) and np.array_equal(b, want)
    assert (np.diff(planes, axis=-1) >= 0).all()
    with pytest.raises(ops.TamdError):  # ... not with a KV cache
        tamd_mask(batch, 1, s + 1, q_offset=s, mask_function=fn, device="cpu")


def test_sliding_window_argument_of_the_attention_function():
    """The `sliding_window` keyword (models/mistral/modeling_mistral.py: handed to every attention function) when the mask
    does not carry the window: planes for the square causal case, nothing for decode steps and windows covering the row,
    a refusal for a prefill over a longer cache."""
    from transformers_amd.attention import window_q_start

    dev = torch.device("cpu")
    assert window_q_start(None, None, 2, 8, 8, True, dev) is None
    assert window_q_start(None, 8, 2, 8, 8, True, dev) is None and window_q_start(None, 4, 2, 1, 8, True, dev) is None
    qs = window_q_start(None, 3, 2, 8, 8, True, dev)
    assert np.array_equal(qs.numpy(), ops.sliding_window_q_start(2, 8, 3, dev).numpy())
    assert qs[0, 1].tolist() == [0, 0, 0, 1, 2, 3, 4, 5] and qs[1, 0].tolist() == [2, 3, 4, 5, 6, 7, 7, 7]
    assert window_q_start(None, 3, 2, 8, 8, True, dev) is qs  # (one construction per forward, not per layer)
    keep = torch.zeros(2, 2, 8, dtype=torch.int32)
    assert window_q_start(keep, 3, 2, 8, 8, True, dev) is keep  # the mask already carried it
    with pytest.raises(ops.TamdError):
        window_q_start(None, 3, 2, 4, 8, True, dev)
    with pytest.raises(ops.TamdError):
        window_q_start(None, 3, 2, 8, 8, False, dev)


def test_mask_for_cross_attention_is_the_padding_mask():
    """create_bidirectional_mask with encoder_hidden_states calls the mask factory with q_offset = 0
    and kv_length = encoder length != q_length (masking_utils.py: `create_bidirectional_mask`).  That is not a cache:
    every query sees every valid encoder key -- for q_length < kv_length, q_length == 1 and q_length > kv_length alike."""
    from transformers import masking_utils as mu

    bi = getattr(mu, "bidirectional_mask_function", None)
    if bi is None:
        pytest.skip("reference without bidirectional_mask_function")
    enc = torch.ones(2, 10, dtype=torch.long)
    enc[1, 7:] = 0
    for q_len in (4, 1, 16):
        m = tamd_mask(2, q_len, 10, q_offset=0, mask_function=bi, attention_mask=enc)
        assert torch.equal(m, enc) and m.shape == (2, 10)
        assert tamd_mask(2, q_len, 10, q_offset=0, mask_function=bi) is None
    # a causal prefill into a pre-allocated cache still yields the mask over the slots in use: its LENGTH carries kv_len
    m = tamd_mask(2, 4, 32, q_offset=3, mask_function=mu.causal_mask_function, device=torch.device("cpu"))
    assert m.shape == (2, 7) and bool(m.all())
    from transformers_amd.attention import _mask_kv_len

    assert _mask_kv_len(m) == 7 and _mask_kv_len(m.to(torch.int32)[:1]) == 7 and _mask_kv_len(None) is None
    with pytest.raises(ops.TamdError):
        tamd_mask(2, 4, 6, q_offset=3, mask_function=mu.causal_mask_function, device=torch.device("cpu"))


def test_varlen_kwargs_give_the_packed_bounds():
    """The reference's varlen kwargs (cu_seq_lens_q/k, modeling_flash_attention_utils.py:575-590) describe the same
    block-diagonal structure as restarting position_ids: `q_start_from_cu_seqlens` reproduces `packed_q_start` of the
    corresponding sequence ids (the oracle's packed_bounds), including a padding tail after the last boundary."""
    from transformers import masking_utils as mu

    from transformers_amd.attention import varlen_q_start

    pos = torch.tensor([[0, 1, 2, 3, 0, 1, 0, 1, 2, 3, 4, 5]])
    ids = mu.find_packed_sequence_indices(pos)
    cu = torch.tensor([0, 4, 6, 12], dtype=torch.int32)
    got = ops.q_start_from_cu_seqlens(cu, 12)
    assert got.dtype == torch.int32 and got.shape == (2, 1, 12)
    assert np.array_equal(got.numpy(), ops.packed_q_start(ids).numpy())
    assert np.array_equal(got.numpy(), orc.packed_bounds(ids.numpy()))
    tail = ops.q_start_from_cu_seqlens(torch.tensor([0, 4, 6, 10]), 12)  # two padding tokens after the last sequence
    assert tail[0, 0].tolist() == [0, 0, 0, 0, 4, 4, 6, 6, 6, 6, 10, 10] and tail[1, 0].tolist()[-2:] == [11, 11]
    # the attention-layer hook: used when the mask carried no packing, refused where it cannot be honoured
    assert varlen_q_start(None, {}, 1, 12, 12, True) is None
    assert torch.equal(varlen_q_start(None, {"cu_seq_lens_q": cu, "cu_seq_lens_k": cu}, 1, 12, 12, True), got)
    with pytest.raises(ops.TamdError):
        varlen_q_start(None, {"cu_seq_lens_q": cu, "cu_seq_lens_k": cu}, 2, 12, 12, True)       # not one flattened row
    with pytest.raises(ops.TamdError):
        varlen_q_start(None, {"cu_seq_lens_q": cu, "cu_seq_lens_k": torch.tensor([0, 5, 6, 12])}, 1, 12, 12, True)


def test_key_validity_mask_is_converted_once_per_forward():
    """Every layer of a forward hands the attention function the same mask tensor: `split_mask` converts it to the kernels'
    [B, kv_len] bool form once and the other layers get that tensor back; another mask, an in-place update of the same one,
    another batch or key length are converted afresh (a stale hit would attend to padding)."""
    from transformers_amd.attention import split_mask

    am = torch.tensor([[1, 1, 1, 0], [1, 1, 1, 1]])
    kv1, _ = split_mask(am, 2, 4)
    kv2, _ = split_mask(am, 2, 4)
    assert kv1 is kv2 and kv1.dtype == torch.bool and kv1.tolist() == [[True, True, True, False], [True, True, True, True]]
    am[0, 3] = 1                                   # in place: the version moves
    kv3, _ = split_mask(am, 2, 4)
    assert kv3 is not kv1 and kv3.all()
    other = am.clone()
    other[1, 0] = 0
    kv4, _ = split_mask(other, 2, 4)
    assert kv4.tolist()[1][0] is False
    kv5, _ = split_mask(am, 2, 3)                  # the last 3 key slots
    assert kv5.shape == (2, 3)
    kv6, _ = split_mask(TamdMask(am, None), 2, 4)  # the mask factory's wrapper around the same tensor
    assert kv6.all() and kv6.shape == (2, 4)
    add = torch.zeros(2, 1, 1, 4)
    add[0, 0, 0, 0] = torch.finfo(torch.float32).min
    kv7, _ = split_mask(add, 2, 4)                 # the reference's additive 4-D padding mask
    assert kv7.tolist()[0] == [False, True, True, True] and split_mask(add, 2, 4)[0] is kv7


def test_mask_caches_under_inference_mode():
    """tensors created inside `torch.inference_mode()` have no version counter (`t._version` raises).  The
    per-forward mask / varlen caches skip them -- converted on every call, never a stale hit -- instead of crashing every
    BERT / CLIP / Llama forward with a padding mask under inference_mode."""
    from transformers_amd.attention import _key_valid_from_mask, split_mask, varlen_q_start

    with torch.inference_mode():
        am = torch.ones(2, 8, dtype=torch.long)
        am[0, 6:] = 0
        kv = _key_valid_from_mask(am, 2, 8)
        assert kv.dtype == torch.bool and kv.tolist()[0] == [True] * 6 + [False] * 2 and kv[1].all()
        am[0, 6:] = 1                               # an in-place update nothing could detect: must not be served from a cache
        assert _key_valid_from_mask(am, 2, 8).all()
        kv2, _ = split_mask(TamdMask(am, None), 2, 8)
        assert kv2.all()
        cu = torch.tensor([0, 4, 6, 12], dtype=torch.int32)
        a = varlen_q_start(None, {"cu_seq_lens_q": cu, "cu_seq_lens_k": cu}, 1, 12, 12, True)
        cu[1] = 3
        b = varlen_q_start(None, {"cu_seq_lens_q": cu, "cu_seq_lens_k": cu}, 1, 12, 12, True)
        assert a[0, 0].tolist()[:6] == [0, 0, 0, 0, 4, 4] and b[0, 0].tolist()[:6] == [0, 0, 0, 3, 3, 3]
    outside = torch.ones(2, 8, dtype=torch.long)    # ordinary tensors keep the once-per-forward behaviour
    assert split_mask(outside, 2, 8)[0] is split_mask(outside, 2, 8)[0]


def test_fused_layer_paths_respect_accelerate_wrappers():
    """the fused layer paths read their children's weights directly, so a forward wrapper accelerate installs on
    a CHILD for `device_map` / offload (`_hf_hook`) would not run and the weight would still sit on `meta`.  `_placement_ok`
    sends such layers to the reference module (whose children are called); a wrapper on the layer itself is fine."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer

    from transformers_amd.models.common import _placement_ok

    cfg = LlamaConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                      num_key_value_heads=1, head_dim=64)
    layer = LlamaDecoderLayer(cfg, 0)
    cpu = torch.device("cpu")
    probe = (layer.self_attn.q_proj.weight, layer.mlp.down_proj.weight)
    assert _placement_ok(layer, cpu, *probe)
    layer._hf_hook = object()                       # a wrapper on the layer itself has run before its forward is entered
    layer.__dict__.pop("_tamd_placement")
    assert _placement_ok(layer, cpu, *probe)
    layer.mlp.down_proj._hf_hook = object()         # ... on a child it has not
    assert _placement_ok(layer, cpu, *probe)        # (cached on the probe weights' identity)
    layer.__dict__.pop("_tamd_placement")
    assert not _placement_ok(layer, cpu, *probe)
    del layer.mlp.down_proj._hf_hook
    # a multi-GPU `device_map="auto"` puts a plain execution-device AlignDevicesHook on EVERY submodule, also of
    # layers that sit wholly on one device -- only hooks that offload or execute elsewhere send the layer to the reference path
    from accelerate.hooks import AlignDevicesHook, SequentialHook

    for hook, want in ((AlignDevicesHook(execution_device="cpu", io_same_device=False), True),
                       (AlignDevicesHook(execution_device="cpu", offload=True), False),
                       (AlignDevicesHook(execution_device="meta"), False),
                       (SequentialHook(AlignDevicesHook(execution_device="cpu"), AlignDevicesHook(offload=True,
                                                                                                   execution_device="cpu")), False)):
        layer.mlp.down_proj._hf_hook = hook
        layer.__dict__.pop("_tamd_placement")
        assert _placement_ok(layer, cpu, *probe) is want, hook
    del layer.mlp.down_proj._hf_hook
    layer.__dict__.pop("_tamd_placement")
    # offload replaces the parameter object (meta) -- that alone invalidates the cached verdict
    layer.mlp.down_proj.weight = torch.nn.Parameter(torch.empty_like(layer.mlp.down_proj.weight, device="meta"))
    probe = (layer.self_attn.q_proj.weight, layer.mlp.down_proj.weight)
    assert not _placement_ok(layer, cpu, *probe)


def test_weight_gradient_cut_packs_the_dispatch_rounds():
    """`gemm_dw_balanced` (csrc/torch_binding.cpp): a weight-gradient product whose 256 x 256 tile grid ends in a mostly empty
    dispatch round is cut into a whole-rounds part and a remainder of at most half a round (which split-K fills) instead of
    splitting the whole product along K.  Llama-3-8B at 32768 tokens: q|k|v (384 tiles) at the q | k|v boundary, down_proj (896)
    at 48 of its 56 tile columns, lm_head (8016) at 496 of 501 tile rows; grids that already pack stay one launch."""
    from transformers_amd import _native

    t = 32768
    assert _native.dw_cut(6144, 4096, t) == (0, 4096)        # 256 tiles = 1 round | 128 tiles -> 2 splits
    assert _native.dw_cut(4096, 14336, t) == (1, 12288)      # 768 tiles = 3 rounds | 128 tiles
    assert _native.dw_cut(128256, 4096, t) == (0, 126976)    # 31 rounds | 80 tiles
    for m, n in ((28672, 4096), (4096, 4096), (2048, 4096), (4096, 11008), (2304, 768)):
        assert _native.dw_cut(m, n, t)[0] == -1, (m, n)      # whole rounds, at most one round, or no cut of <= half a round
    assert _native.dw_cut(6144, 4096, t, cus=304) == (0, 4864)  # (a round is the DEVICE's CU count -- MI300X: 304 tiles | 80)
    assert _native.dw_cut(6144, 4096, 1000)[0] == -1         # ragged token counts (the ping-pong kernel): untouched
    assert _native.dw_cut(6100, 4096, t)[0] == -1            # ragged outputs: untouched
    for m, n in ((6144, 4096), (4096, 14336), (128256, 4096)):
        axis, at = _native.dw_cut(m, n, t)
        tm, tn = m // 256, n // 256
        main = (at // 256) * (tn if axis == 0 else tm)
        assert main % 256 == 0 and 0 < tm * tn - main <= 128
