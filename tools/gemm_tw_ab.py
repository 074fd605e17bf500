"""The 256 x 128 two-workgroups-per-CU GEMM kernel (gemm_tw_kernel, schedule hint "tw") against the 256 x 256 full-line kernel
("fl"), the 128 x 128 tile ("sm"), the library default and torch (hipBLASLt) on forward products: bert-base at batch 32,
the LLaVA language model at 1088 tokens, the CLIP-L tower, the five Llama-3-8B forward shapes at 32768 tokens.
Interleaved, min of 3 rounds; TFLOP/s = 2*M*N*K / time.   python tools/gemm_tw_ab.py [group ...] > gpurun_out/<tag>_gemm_tw_ab.jsonl"""
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


GROUPS = {
    "bert": [("bert qkv", 16384, 2304, 768, "bias"), ("bert o", 16384, 768, 768, "bias"), ("bert fc1", 16384, 3072, 768, "bias"),
             ("bert fc2", 16384, 768, 3072, "bias"), ("bert fc1+gelu", 16384, 3072, 768, "bias_act"),
             ("bert mlm", 16384, 30528, 768, "bias")],
    "llava": [("lm qkv", 1088, 12288, 4096, "none"), ("lm o", 1088, 4096, 4096, "res"), ("lm down", 1088, 4096, 11008, "res"),
              ("lm gate_up", 1088, 22016, 4096, "none"), ("lm head", 1088, 32064, 4096, "none"),
              ("clip qkv", 577, 3072, 1024, "bias"), ("clip fc1", 577, 4096, 1024, "bias_act"), ("clip fc2", 577, 1024, 4096, "bias_res")],
    "llama": [("qkv", 32768, 6144, 4096, "none"), ("o_proj", 32768, 4096, 4096, "res"), ("gate_up", 32768, 28672, 4096, "none"),
              ("down", 32768, 4096, 14336, "res"), ("sq8k", 8192, 8192, 8192, "none")],
    "mid": [("m4096 qkv", 4096, 6144, 4096, "none"), ("m4096 down", 4096, 4096, 14336, "res"), ("m2048 gate_up", 2048, 28672, 4096, "none")],
}
which = sys.argv[1:] or list(GROUPS)
for grp in which:
    for name, m, n, k, epi in GROUPS[grp]:
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
        same = bool(torch.equal(ops.raw_gemm(x, w, sched="tw", **kw), ref))
        arms = ["fl", "tw", "default", "torch"] + (["sm"] if m * n <= 1088 * 32064 else [])
        t = {a: [] for a in arms}
        lin = (lambda: torch.nn.functional.linear(x, w, bias if "bias" in epi else None))
        iters = 10 if m * n * k > 2e12 else 30
        for _ in range(3):
            for a in arms:
                if a == "torch":
                    t[a].append(timeit(lin, iters))
                elif a == "default":
                    t[a].append(timeit(lambda: ops.raw_gemm(x, w, out=out, **kw), iters))
                else:
                    t[a].append(timeit(lambda: ops.raw_gemm(x, w, sched=a, out=out, **kw), iters))
        fl = 2.0 * m * n * k
        row = {"case": name, "M": m, "N": n, "K": k, "epi": epi, "tiles256": -(-m // 256) * -(-n // 256),
               "tiles_tw": -(-m // 256) * -(-n // 128), "tw_bit_identical": same}
        for key, v in t.items():
            row[key + "_us"] = round(min(v), 1)
            row[key + "_TF"] = round(fl / min(v) / 1e6)
        print(json.dumps(row), flush=True)
