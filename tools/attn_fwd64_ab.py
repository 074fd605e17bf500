"""A/B of the experimental 64-rows-per-wave forward attention kernel (csrc/attention_fwd64.inc, diagnostic library)
against the product kernel at the Llama-3-8B shape: interleaved rounds, ms and TFLOP/s, bit-identity check."""
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
    o0, l0 = ops.raw_attn_fwd(q, k, v, scale, causal)
    lib.tamd_attn_set_fwd64(1)
    o1, l1 = ops.raw_attn_fwd(q, k, v, scale, causal)
    lib.tamd_attn_set_fwd64(0)
    res = {"shape": name, "bit_identical": bool(torch.equal(o0, o1) and torch.equal(l0, l1)), "ms": {"fwd32": [], "fwd64": []}}
    for rnd in range(3):
        for on, key in ((0, "fwd32"), (1, "fwd64")):
            lib.tamd_attn_set_fwd64(on)
            ops.raw_attn_fwd(q, k, v, scale, causal)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.raw_attn_fwd(q, k, v, scale, causal)
            e1.record()
            torch.cuda.synchronize()
            res["ms"][key].append(round(e0.elapsed_time(e1) / 5, 4))
    lib.tamd_attn_set_fwd64(0)
    res["TFLOPs"] = {kk: round(fl / (min(vv) * 1e-3) / 1e12) for kk, vv in res["ms"].items()}
    print(json.dumps(res), flush=True)
