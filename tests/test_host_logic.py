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
        tamd_mask(2, 12, 12, mask_function=mu.and_masks(mu.causal_mask_function, mu.sliding_window_overlay(4)))
    with pytest.raises(ops.TamdError):
        tamd_mask(2, 12, 12, mask_function=lambda b, h, q, k: q >= k)
    with pytest.raises(ops.TamdError):
        tamd_mask(2, 4, 12, q_offset=8, mask_function=fn)
