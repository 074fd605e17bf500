"""The 128 x 128 GEMM tile (gemm_sm_kernel, schedule hint "sm") against the 256 x 256 full-line kernel ("fl") and the
library default (which may take split-K) on forward products whose 256 x 256 grid cannot fill the GPU: the CLIP-L tower's
four projections (577 tokens), the LLaVA language model's at 1088 tokens, and a sweep over the tile count.  Interleaved,
min of 3 rounds; TFLOP/s = 2*M*N*K / time.   python tools/gemm_sm_ab.py > gpurun_out/<tag>_gemm_sm_ab.jsonl"""
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
    return s.elapsed_time(e) / iters * 1e3  # us


CASES = [  # name, M, N, K, epilogue
    ("clip qkv", 577, 3072, 1024, "bias"), ("clip o", 577, 1024, 1024, "bias_res"), ("clip fc1", 577, 4096, 1024, "bias_act"),
    ("clip fc2", 577, 1024, 4096, "bias_res"), ("proj1", 576, 4096, 1024, "bias_act"), ("proj2", 576, 4096, 4096, "bias"),
    ("lm qkv", 1088, 12288, 4096, "none"), ("lm o", 1088, 4096, 4096, "res"), ("lm down", 1088, 4096, 11008, "res"),
    ("lm gate_up", 1088, 22016, 4096, "none"), ("bert qkv b4", 2048, 2304, 768, "bias"), ("bert fc1 b4", 2048, 3072, 768, "bias_act"),
    ("gpt2 b1", 128, 2304, 768, "bias"),
] + [(f"sweep {t}x256", 1024, 256 * t // 4, 2048, "none") for t in (8, 16, 32, 48, 64, 96, 128, 192, 256)]

for name, m, n, k, epi in CASES:
    torch.manual_seed(0)
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev).bfloat16()
    res = torch.randn(m, n, device=dev).bfloat16()
    kw = {"none": {}, "bias": dict(bias=bias, epilogue=ops.EPI_BIAS), "res": dict(residual=res, epilogue=ops.EPI_RESIDUAL),
          "bias_res": dict(bias=bias, residual=res, epilogue=ops.EPI_RESIDUAL),
          "bias_act": dict(bias=bias, epilogue=ops.EPI_BIAS_ACT, act=ops.ACT_QUICK_GELU)}[epi]
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    ref = ops.raw_gemm(x, w, sched="fl", **kw)
    same = bool(torch.equal(ops.raw_gemm(x, w, sched="sm", **kw), ref))
    t = {"fl": [], "sm": [], "default": [], "torch": []}
    lin = (lambda: torch.nn.functional.linear(x, w, bias if "bias" in epi else None))
    for _ in range(3):
        t["fl"].append(timeit(lambda: ops.raw_gemm(x, w, sched="fl", out=out, **kw)))
        t["sm"].append(timeit(lambda: ops.raw_gemm(x, w, sched="sm", out=out, **kw)))
        t["default"].append(timeit(lambda: ops.raw_gemm(x, w, out=out, **kw)))
        t["torch"].append(timeit(lin))
    fl = 2.0 * m * n * k
    row = {"case": name, "M": m, "N": n, "K": k, "epi": epi, "tiles256": -(-m // 256) * -(-n // 256),
           "tiles128": -(-m // 128) * -(-n // 128), "sm_bit_identical": same}
    for key, v in t.items():
        row[key + "_us"] = round(min(v), 1)
        row[key + "_TF"] = round(fl / min(v) / 1e6)
    print(json.dumps(row), flush=True)
