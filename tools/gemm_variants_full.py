"""ping-pong (pp) vs one-wave-per-SIMD (w4) kernel on forward and backward (k-major) shapes; one process per variant."""
import os, subprocess, sys
code = r'''
import sys, torch, json, os
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=15, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
out = {}
for name, m, n, k in [("qkv", 32768, 6144, 4096), ("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096), ("down", 32768, 4096, 14336)]:
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * 0.02).bfloat16(); dy = torch.randn(m, n, device=dev).bfloat16()
    fl = 2.0*m*n*k
    out[name] = [round(fl/timeit(lambda: ops.raw_gemm(x, w))/1e12), round(fl/timeit(lambda: ops.raw_gemm(dy, w, b_kn=True))/1e12), round(fl/timeit(lambda: ops.raw_gemm(dy, x, a_km=True, b_kn=True))/1e12)]
    ref = x[:300].float() @ w[:300].float().t()
    c = ops.raw_gemm(x, w)[:300, :300].float()
    out[name].append(round(((c - ref).norm() / ref.norm()).item(), 5))
    del x, w, dy
print(json.dumps({"TAMD_GEMM": os.environ.get("TAMD_GEMM", "v2"), "fwd/dx/dw/relerr": out}))
'''
for v in sys.argv[1:] or ["pp", "w4"]:
    e = dict(os.environ); e["TAMD_GEMM"] = v
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
