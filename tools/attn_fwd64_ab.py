"""A/B of the experimental forward attention kernels of the diagnostic library against the product kernel at the
Llama-3-8B shape: interleaved rounds, ms and TFLOP/s, bit-identity check.
  fwd64  64 query rows per wave (csrc/attention_fwd64.inc, tamd_attn_set_fwd64)
  pair   causal only: two query tiles per workgroup, heaviest + lightest (attn_fwd_pair_kernel, tamd_attn_set_pair)"""
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402
from transformers_amd import ops  # noqa: E402

lib = _diag.use_diag()
dev = torch.device("cuda:0")
for name, b, s, hq, hkv, causal in [("llama3-8b causal", 8, 4096, 32, 8, True), ("llama3-8b bidirectional", 8, 4096, 32, 8, False)]:
    d = 128
    q = torch.randn(b, s, hq, d, device=dev).bfloat16()
    k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    scale = 1 / math.sqrt(d)
    fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
    arms = ["fwd32", "fwd64"] + (["pair"] if causal else [])

    def select(key):
        lib.tamd_attn_set_fwd64(int(key == "fwd64"))
        lib.tamd_attn_set_pair(int(key == "pair"))

    o0, l0 = ops.raw_attn_fwd(q, k, v, scale, causal)
    same = {}
    for key in arms[1:]:
        select(key)
        o1, l1 = ops.raw_attn_fwd(q, k, v, scale, causal)
        same[key] = bool(torch.equal(o0, o1) and torch.equal(l0, l1))
    select("fwd32")
    res = {"shape": name, "bit_identical": same, "ms": {key: [] for key in arms}}
    for rnd in range(3):
        for key in arms:
            select(key)
            ops.raw_attn_fwd(q, k, v, scale, causal)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.raw_attn_fwd(q, k, v, scale, causal)
            e1.record()
            torch.cuda.synchronize()
            res["ms"][key].append(round(e0.elapsed_time(e1) / 5, 4))
    select("fwd32")
    res["TFLOPs"] = {kk: round(fl / (min(vv) * 1e-3) / 1e12) for kk, vv in res["ms"].items()}
    print(json.dumps(res), flush=True)

# ---- backward: the dQ kernel with two query tiles per workgroup (tamd_attn_set_pair bit 1); whole-backward ms
# (delta + dQ + dK/dV -- only dQ differs between the arms), causal Llama-3-8B shape, interleaved rounds
b, s, hq, hkv, d = 8, 4096, 32, 8, 128
q = torch.randn(b, s, hq, d, device=dev).bfloat16()
k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
do = torch.randn(b, s, hq, d, device=dev).bfloat16()
scale = 1 / math.sqrt(d)
o, lse = ops.raw_attn_fwd(q, k, v, scale, True)
dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
ref = [t.clone() for t in ops.raw_attn_bwd(q, k, v, o, lse, do, scale, True, None)]
lib.tamd_attn_set_pair(2)
got = ops.raw_attn_bwd(q, k, v, o, lse, do, scale, True, None)
res = {"shape": "llama3-8b causal backward", "bit_identical": all(bool(torch.equal(x, y)) for x, y in zip(got, ref)),
       "ms": {"bwd": [], "bwd_dq_pair": []}}
for rnd in range(3):
    for on, key in ((0, "bwd"), (2, "bwd_dq_pair")):
        lib.tamd_attn_set_pair(on)
        ops.raw_attn_bwd(q, k, v, o, lse, do, scale, True, None, dq=dq, dk=dk, dv=dv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.raw_attn_bwd(q, k, v, o, lse, do, scale, True, None, dq=dq, dk=dk, dv=dv)
        e1.record()
        torch.cuda.synchronize()
        res["ms"][key].append(round(e0.elapsed_time(e1) / 5, 4))
lib.tamd_attn_set_pair(0)
print(json.dumps(res), flush=True)
