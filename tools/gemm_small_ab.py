"""GEMM at bert-base shapes (tile grids near or below one wave of 256x256 tiles): ours (default dispatch, split-K where
the policy picks it) vs torch.mm (hipBLASLt).  TFLOP/s."""
import json, sys, torch
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for name, m, n, k in [("attn.out", 16384, 768, 768), ("qkv", 16384, 2304, 768), ("ffn1", 16384, 3072, 768), ("ffn2", 16384, 768, 3072)]:
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * 0.02).bfloat16(); dy = torch.randn(m, n, device=dev).bfloat16()
    fl = 2.0 * m * n * k
    r = {"shape": name, "mnk": [m, n, k]}
    r["fwd"] = [round(fl / timeit(lambda: ops.raw_gemm(x, w)) / 1e12), round(fl / timeit(lambda: torch.mm(x, w.t())) / 1e12)]
    r["dx"] = [round(fl / timeit(lambda: ops.raw_gemm(dy, w, b_kn=True)) / 1e12), round(fl / timeit(lambda: torch.mm(dy, w)) / 1e12)]
    r["dw"] = [round(fl / timeit(lambda: ops.raw_gemm(dy, x, a_km=True, b_kn=True)) / 1e12), round(fl / timeit(lambda: torch.mm(dy.t(), x)) / 1e12)]
    r["dw_unsplit"] = round(fl / timeit(lambda: ops.raw_gemm(dy, x, a_km=True, b_kn=True, sched="fl")) / 1e12)
    print(json.dumps(r), flush=True)
