"""The SiLU*up backward as the way out of the down projection's dX GEMM (tamd_gemm_swiglu_bwd) against the two kernels it
replaces (tamd_gemm with a k-major B, then tamd_swiglu_bwd), interleaved in one process, with a bit-identity check.
JSON lines to stdout:   python tools/gemm_swiglu_bwd_ab.py > gpurun_out/<tag>_gemm_swiglu_bwd_ab.jsonl"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    shapes = [("llama3-8b", 32768, 14336, 4096), ("llama2-7b", 32768, 11008, 4096), ("llama3-8b b2", 8192, 14336, 4096)]
    for name, t, inter, k in shapes:
        torch.manual_seed(0)
        dy = torch.randn(t, k, device=dev).bfloat16()
        wd = (torch.randn(k, inter, device=dev) * k ** -0.5).bfloat16()
        gu = torch.randn(t, 2 * inter, device=dev).bfloat16()
        assert ops.gemm_swiglu_bwd_supported(dy, wd, gu)

        def two():
            return ops.raw_swiglu_bwd(gu, ops.raw_gemm(dy, wd, b_kn=True))[0]

        def fused():
            return ops.raw_gemm_swiglu_bwd(dy, wd, gu)

        same = torch.equal(two(), fused())
        rec = {"shape": name, "t": t, "inter": inter, "k": k, "same_bits": same}
        for rnd in range(3):  # interleaved: the box's clock drifts with its temperature
            rec.setdefault("gemm_ms", []).append(timeit(lambda: ops.raw_gemm(dy, wd, b_kn=True)))
            rec.setdefault("two_kernels_ms", []).append(timeit(two))
            rec.setdefault("fused_ms", []).append(timeit(fused))
        fl = 2.0 * t * inter * k
        rec["gemm_TF"] = fl / min(rec["gemm_ms"]) / 1e9
        rec["fused_TF_of_the_product"] = fl / min(rec["fused_ms"]) / 1e9
        rec["saved_ms"] = min(rec["two_kernels_ms"]) - min(rec["fused_ms"])
        print(json.dumps(rec), flush=True)
        del dy, wd, gu


if __name__ == "__main__":
    main()
