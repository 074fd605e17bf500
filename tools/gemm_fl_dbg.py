"""Ablation timing of gemm_fl_kernel (TAMD_GEMM_DBG bit mask: 1 no LDS-DMA after the prologue, 2 no LDS fragment
reads, 4 no vmcnt wait, 8 no barrier; results are wrong by construction) on forward shapes."""
import json, os, subprocess, sys
code = r'''
import sys, torch, json, os
sys.path.insert(0, ".")
from transformers_amd import ops
sys.path.insert(0, "tools")
import _diag
_diag.use_diag()
dev = torch.device("cuda:0")
out = {"dbg": os.environ.get("TAMD_GEMM_DBG", "0")}
for name, m, n, k in [("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096), ("down", 32768, 4096, 14336)]:
    x = torch.randn(m, k, device=dev).bfloat16(); w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    for _ in range(3): ops.raw_gemm(x, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.raw_gemm(x, w)
    e.record(); torch.cuda.synchronize()
    out[name] = round(2.0 * m * n * k / (s.elapsed_time(e) / 10 * 1e-3) / 1e12)
print(json.dumps(out))
'''
for v in sys.argv[1:] or ["0", "1", "2", "12", "15"]:
    e = dict(os.environ); e["TAMD_GEMM_DBG"] = v
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
