"""Attention under the reference's sliding-window / chunked masks (masking_utils.py:92-113) at the Llama-3-8B shape
(batch 8 x 4096, 32 query / 8 key-value heads, head_dim 128): forward and backward times of the kernels with the bound
planes against the plain causal launch, and against the work the mask leaves (visible (q, k) pairs / causal pairs) --
tiles left of the window are neither loaded nor visited, so the time should follow the visible area plus the masked edge
tiles.  One JSON line per mask.

    python tools/attn_window_bench.py [--seq 4096] [--batch 8]
"""
import argparse
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seq", type=int, default=4096)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
b, s, hq, hkv, d = args.batch, args.seq, 32, 8, 128


def timeit(fn, iters=args.iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


torch.manual_seed(0)
q = torch.randn(b, s, hq, d, device=dev).bfloat16()
k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
scale = 1 / math.sqrt(d)
causal_pairs = s * (s + 1) / 2
base = None
masks = [("causal", None)]
for w in (2048, 1024, 512, 128):
    if w < s:
        masks.append((f"window {w}", ops.sliding_window_q_start(b, s, w, dev)))
for c in (2048, 1024):
    if c < s:
        masks.append((f"chunk {c}", ops.chunked_q_start(b, s, c, None, dev)))
for name, planes in masks:
    o, lse = ops.raw_attn_fwd(q, k, v, scale, True, q_start=planes)
    do = torch.randn_like(o)
    tf = timeit(lambda: ops.raw_attn_fwd(q, k, v, scale, True, q_start=planes))
    tb = timeit(lambda: ops.raw_attn_bwd(q, k, v, o, lse, do, scale, True, q_start=planes))
    pairs = causal_pairs if planes is None else float((torch.arange(s, device=dev) - planes[0, 0].long() + 1).sum())
    rec = {"mask": name, "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4), "visible_share_of_causal": round(pairs / causal_pairs, 4),
           "fwd_TFLOPs": round(4.0 * b * hq * pairs * d / (tf * 1e-3) / 1e12, 1),
           "bwd_TFLOPs": round(10.0 * b * hq * pairs * d / (tb * 1e-3) / 1e12, 1)}
    if base is None:
        base = (tf, tb)
    else:
        rec["fwd_vs_causal"], rec["bwd_vs_causal"] = round(tf / base[0], 4), round(tb / base[1], 4)
    print(json.dumps(rec), flush=True)
