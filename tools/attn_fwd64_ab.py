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
for si, (name, b, s, hq, hkv, causal) in enumerate(SHAPES):
    full = si < 2  # the ablation arms only at the Llama-3-8B shape
    d = 128
    q = torch.randn(b, s, hq, d, device=dev).bfloat16()
    k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    scale = 1 / math.sqrt(d)
    fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
    # variants of attn_fwd64_launch (attention_fwd64.hip); 5-7 are ablations (wrong results).  The records r03f..m were
    # taken with the first kernel and its schedule variants still in the library (arms v1..v12, g2 = today's fwd64)
    VARIANTS = {"fwd64": 1, "fwd64_split": 2, "abl_no_dma": 5, "abl_no_softmax": 6, "abl_no_dma_no_softmax": 7}
    arms = ["fwd32"] + [a for a in VARIANTS if full or not a.startswith("abl")]

    def select(key):
        lib.tamd_attn_set_fwd64(VARIANTS.get(key, 0))

    select("fwd32")
    o0, l0 = ops.raw_attn_fwd(q, k, v, scale, causal)
    same = {}
    for key in arms[1:]:
        if key.startswith("abl"):
            continue
        select(key)
        o1, l1 = ops.raw_attn_fwd(q, k, v, scale, causal)
        same[key] = bool(torch.equal(o0, o1) and torch.equal(l0, l1))
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
