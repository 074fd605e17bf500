"""Prompt-sized GEMMs (the LLaVA language model at 1088 tokens) with COLD weights: in the model every layer's weights come
from HBM, in a micro-benchmark that re-uses one weight they come from the 256 MB Infinity Cache -- in the forward the
gate|up / q|k|v launches run 30-38 % longer than in tools/gemm_small_ab.py (profiles/r04k_llava_kernel_stats.csv).  Here each
arm cycles through enough distinct weights to exceed the cache (footprint >= 1.5 GB), next to the same arm on ONE weight.
   python tools/gemm_cold_ab.py > gpurun_out/<tag>_gemm_cold_ab.jsonl"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1088


def timeit(fn, n, iters, warm):
    for i in range(warm):
        fn(i % n)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i % n)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


CASES = [("lm qkv", 12288, 4096, "none"), ("lm o", 4096, 4096, "res"), ("lm gate_up+swiglu", 22016, 4096, "swiglu"),
         ("lm down", 4096, 11008, "res"), ("lm head", 32064, 4096, "none")]
for name, n, k, epi in CASES:
    torch.manual_seed(0)
    copies = max(2, int(1.6e9 // (n * k * 2)) + 1)
    ws = [(torch.randn(n, k, device=dev) * 0.02).bfloat16() for _ in range(copies)]
    x = torch.randn(M, k, device=dev).bfloat16()
    res = torch.randn(M, n, device=dev).bfloat16()

    def arm(sched):
        if epi == "swiglu":
            if sched is None:
                return lambda i: ops.raw_gemm_swiglu(x, ws[i], need_gu=False)
            return lambda i: ops.raw_swiglu_fwd(ops.raw_gemm(x, ws[i], sched=sched))
        kw = dict(residual=res, epilogue=ops.EPI_RESIDUAL) if epi == "res" else {}
        return lambda i: ops.raw_gemm(x, ws[i], sched=sched, **kw)

    arms = {"default": arm(None), "fl": arm("fl"), "sm": arm("sm"),
            "torch": (lambda i: torch.nn.functional.linear(x, ws[i]))}
    rec = {"case": name, "M": M, "N": n, "K": k, "epi": epi, "weights_cycled": copies, "footprint_GB": round(copies * n * k * 2 / 1e9, 2)}
    for a, fn in arms.items():
        cold = min(timeit(fn, copies, 3 * copies, copies) for _ in range(2))
        warm = min(timeit(fn, 1, 24, 4) for _ in range(2))
        rec[a + "_cold_us"] = round(cold, 1)
        rec[a + "_warm_us"] = round(warm, 1)
    print(json.dumps(rec), flush=True)
