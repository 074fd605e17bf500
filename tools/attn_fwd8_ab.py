"""A/B of the 8-wave attention kernels (256 query rows per workgroup, one workgroup per CU, 4-deep K/V ring requested three
tiles ahead; bit 0 = forward, bit 1 = dQ) against the 4-wave product kernels (128 rows, double-buffered, two workgroups per
CU): interleaved rounds in one process, ms and TFLOP/s, bit-identity check.

    python tools/attn_fwd8_ab.py > gpurun_out/r03d_attn_fwd8_ab.jsonl"""
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


def timed(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, b, s, hq, hkv, d, causal in [("llama3-8b causal", 8, 4096, 32, 8, 128, True),
                                       ("llama3-8b bidirectional", 8, 4096, 32, 8, 128, False),
                                       ("llava prompt causal (1088 x 32 MHA heads)", 1, 1088, 32, 32, 128, True),
                                       ("bert-base bidirectional", 32, 512, 12, 12, 64, False)]:
    q = torch.randn(b, s, hq, d, device=dev).bfloat16()
    k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    do = torch.randn(b, s, hq, d, device=dev).bfloat16()
    scale = 1 / math.sqrt(d)
    fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
    lib.tamd_attn_set_fwd8(0)
    o0, l0 = ops.raw_attn_fwd(q, k, v, scale, causal)
    g0 = [t.clone() for t in ops.raw_attn_bwd(q, k, v, o0, l0, do, scale, causal)]
    lib.tamd_attn_set_fwd8(3)
    o1, l1 = ops.raw_attn_fwd(q, k, v, scale, causal)
    g1 = ops.raw_attn_bwd(q, k, v, o0, l0, do, scale, causal)
    same = {"fwd": bool(torch.equal(o0, o1) and torch.equal(l0, l1)), "bwd": all(bool(torch.equal(x, y)) for x, y in zip(g0, g1))}
    dq, dk, dv = (torch.empty_like(t) for t in (q, k, v))
    res = {"shape": name, "bit_identical": same, "fwd_ms": {"4w": [], "8w": []}, "bwd_ms": {"4w": [], "8w_dq": []}}
    for rnd in range(3):
        for key, sw in (("4w", 0), ("8w", 1)):
            lib.tamd_attn_set_fwd8(sw)
            res["fwd_ms"][key].append(round(timed(lambda: ops.raw_attn_fwd(q, k, v, scale, causal)), 4))
        for key, sw in (("4w", 0), ("8w_dq", 2)):
            lib.tamd_attn_set_fwd8(sw)
            res["bwd_ms"][key].append(round(timed(lambda: ops.raw_attn_bwd(q, k, v, o0, l0, do, scale, causal, None,
                                                                             dq=dq, dk=dk, dv=dv), iters=5), 4))
    lib.tamd_attn_set_fwd8(0)
    res["fwd_TFLOPs"] = {kk: round(fl / (min(vv) * 1e-3) / 1e12) for kk, vv in res["fwd_ms"].items()}
    print(json.dumps(res), flush=True)
