"""A/B the GEMM kernel variants (one process per variant: the variant is latched at first use)."""
import json, os, subprocess, sys
code = r'''
import sys, torch, json, os
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
out = {}
for name, m, n, k in [("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096), ("down", 32768, 4096, 14336)]:
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    t = timeit(lambda: ops.raw_gemm(x, w)); out[name] = round(2.0*m*n*k/t/1e12, 1)
    del x, w
print(json.dumps({"TAMD_GEMM": os.environ.get("TAMD_GEMM", "v2"), "VAR": os.environ.get("TAMD_GEMM_VAR", "0"), **out}))
'''
variants = sys.argv[1:] or ["v1", "0", "2", "10", "18", "26"]
for v in variants:
    env = {"TAMD_GEMM": v} if v.startswith("v") else {"TAMD_GEMM": "v2", "TAMD_GEMM_VAR": v}
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
