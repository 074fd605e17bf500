"""A/B of the forward attention kernel with 64 query rows per wave (csrc/attention_fwd64.hip, selected through
tamd_attn_set_fwd64 of the diagnostic library) against the 32-rows-per-wave kernel: interleaved rounds, ms and TFLOP/s,
bit-identity check, on random data (zero-filled operands clock ~20 % higher: MI355X_MICROARCH.md, DVFS)."""
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
SHAPES = [("llama3-8b causal", 8, 4096, 32, 8, True), ("llama3-8b bidirectional", 8, 4096, 32, 8, False),
          ("llama2-7b causal (MHA)", 4, 4096, 32, 32, True), ("llava prompt 1088 causal", 1, 1088, 32, 32, True),
          ("seq 2048 causal", 16, 2048, 32, 8, True), ("seq 8192 causal", 2, 8192, 32, 8, True)]
for name, b, s, hq, hkv, causal in SHAPES:
    d = 128
    q = torch.randn(b, s, hq, d, device=dev).bfloat16()
    k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    scale = 1 / math.sqrt(d)
    fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
    arms = ["fwd32", "fwd64"]

    def select(key):
        lib.tamd_attn_set_fwd64(int(key == "fwd64"))

    select("fwd32")
    o0, l0 = ops.raw_attn_fwd(q, k, v, scale, causal)
    select("fwd64")
    o1, l1 = ops.raw_attn_fwd(q, k, v, scale, causal)
    same = bool(torch.equal(o0, o1) and torch.equal(l0, l1))
    res = {"shape": name, "bit_identical": same, "ms": {key: [] for key in arms}}
    if not same:
        res["max_abs_diff"] = float((o0.float() - o1.float()).abs().max())
        res["nan"] = bool(torch.isnan(o1.float()).any())
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
