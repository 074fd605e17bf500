"""The M = batch products of a cached decode step (csrc/gemv.hip: VALU kernel for M <= 4, MFMA kernel for 5 .. 16) at Llama-3-8B
dimensions, weights cycled through > 1 GB so that they come from HBM as in the model: us per call and weight bytes / time for
ours, for the 256 x 256 tile kernels (schedule hint) and for torch (hipBLASLt / rocBLAS).
   python tools/gemv_bench.py > gpurun_out/<tag>_gemv_bench.jsonl"""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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


CASES = [("q|k|v", 6144, 4096, "none"), ("o_proj + residual", 4096, 4096, "res"), ("gate|up + SiLU*up", 28672, 4096, "swiglu"),
         ("down_proj + residual", 4096, 14336, "res"), ("lm_head", 128256, 4096, "none")]
for name, n, k, epi in CASES:
    copies = max(2, int(1.2e9 // (n * k * 2)) + 1)
    ws = [(torch.randn(n, k, device=dev) * 0.02).bfloat16() for _ in range(copies)]
    for m in [int(v) for v in os.environ.get("GEMV_BENCH_ROWS", "1,2,4,8,16").split(",")]:
        x = torch.randn(m, k, device=dev).bfloat16()
        res = torch.randn(m, n, device=dev).bfloat16()
        if epi == "swiglu":
            arms = {"ours": lambda i: ops.raw_gemm_swiglu(x, ws[i], need_gu=False),
                    "tiles": lambda i: ops.raw_swiglu_fwd(ops.raw_gemm(x, ws[i], sched="fl")),
                    "torch": lambda i: torch.nn.functional.silu((y := torch.nn.functional.linear(x, ws[i]))[:, :n // 2]) * y[:, n // 2:]}
        else:
            kw = dict(residual=res, epilogue=ops.EPI_RESIDUAL) if epi == "res" else {}
            arms = {"ours": lambda i: ops.raw_gemm(x, ws[i], **kw), "tiles": lambda i: ops.raw_gemm(x, ws[i], sched="fl", **kw),
                    "torch": (lambda i: torch.nn.functional.linear(x, ws[i]) + res) if epi == "res" else (lambda i: torch.nn.functional.linear(x, ws[i]))}
        rec = {"case": name, "M": m, "N": n, "K": k, "weight_MB": round(n * k * 2 / 1e6, 1)}
        for a, fn in arms.items():
            t = min(timeit(fn, copies, 3 * copies, copies) for _ in range(2))
            rec[a + "_us"] = round(t, 1)
            rec[a + "_TBps"] = round(n * k * 2 / t / 1e6, 2)
        print(json.dumps(rec), flush=True)
    del ws
    torch.cuda.empty_cache()
