"""The gate|up GEMM with the SiLU*up epilogue (tamd_gemm_swiglu) against plain GEMM + swiglu kernel at prompt-sized M (the LLaVA
language model: 1088 tokens, Llama-2-7B dims; short Llama-3-8B prompts), forward-only (no gate|up kept) and train mode.
   python tools/gemm_swiglu_small_ab.py > gpurun_out/<tag>_swiglu_small_ab.jsonl"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, m, inter, k in [("llava lm 1088", 1088, 11008, 4096), ("llama3 1024", 1024, 14336, 4096), ("llama3 2048", 2048, 14336, 4096),
                          ("llama3 4096", 4096, 14336, 4096), ("llama3 512", 512, 14336, 4096)]:
    torch.manual_seed(0)
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(2 * inter, k, device=dev) * 0.02).bfloat16()
    arms = {
        "fused_nogu": lambda: ops.raw_gemm_swiglu(x, w, need_gu=False),
        "fused_gu": lambda: ops.raw_gemm_swiglu(x, w, need_gu=True),
        "gemm_then_kernel": lambda: ops.raw_swiglu_fwd(ops.raw_gemm(x, w)),
        "gemm_only": lambda: ops.raw_gemm(x, w),
    }
    t = {a: [] for a in arms}
    for _ in range(3):
        for a, fn in arms.items():
            t[a].append(timeit(fn))
    same = bool(torch.equal(ops.raw_gemm_swiglu(x, w, need_gu=False)[1], ops.raw_swiglu_fwd(ops.raw_gemm(x, w))))
    rec = {"case": name, "M": m, "I": inter, "K": k, "tiles": -(-m // 256) * -(-2 * inter // 256), "same_bits": same}
    rec.update({a + "_us": round(min(v), 1) for a, v in t.items()})
    print(json.dumps(rec), flush=True)
