"""A/B of the full-line GEMM's schedule variants (TAMD_GEMM_SCHED_FL_* hints) on the Llama-3-8B shapes, every layout,
interleaved rounds (DVFS: never compare separate processes).  TFLOP/s, bf16, random normal operands."""
import json
import sys

import torch

sys.path.insert(0, ".")
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SCHEDS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fl", "fl_stag", "fl_nt", "fl_persist", "fl_persist_stag", "fl_all"]
ROUNDS, ITERS = 3, 8


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS * 1e-3


T = 32768
for name, m, n, k in [("qkv", T, 6144, 4096), ("o_proj", T, 4096, 4096), ("gate_up", T, 28672, 4096),
                      ("down", T, 4096, 14336)]:
    fl = 2.0 * m * n * k
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    dy = torch.randn(m, n, device=dev).bfloat16()
    calls = {"fwd": lambda s: ops.raw_gemm(x, w, sched=s), "dx": lambda s: ops.raw_gemm(dy, w, b_kn=True, sched=s),
             "dw": lambda s: ops.raw_gemm(dy, x, a_km=True, b_kn=True, sched=s)}
    for lay, fn in calls.items():
        res = {s: [] for s in SCHEDS}
        for _ in range(ROUNDS):
            for s in SCHEDS:
                res[s].append(round(fl / timeit(lambda: fn(s)) / 1e12))
        ref = {"fwd": lambda: torch.mm(x, w.t()), "dx": lambda: torch.mm(dy, w), "dw": lambda: torch.mm(dy.t(), x)}[lay]
        print(json.dumps({"shape": name, "layout": lay, "TF": res, "torch": round(fl / timeit(ref) / 1e12)}), flush=True)
    del x, w, dy
