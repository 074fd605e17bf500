"""Forward/backward attention throughput on
the Llama-3-8B and BERT-base shapes.  FLOP counts: forward 4*B*H*Sq*Sk*D (x0.5 causal), backward 2.5x."""
import json
import os
import subprocess
import sys

code = r'''
import sys, torch, json, os, math
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
out = {"variant": os.environ.get("TAMD_ATTN_DKDV", "new")}
for name, b, s, hq, hkv, d, causal in [("llama3-8b", 8, 4096, 32, 8, 128, True), ("bert-base", 32, 512, 12, 12, 64, False),
                                        ("clip-l", 16, 577, 16, 16, 64, False)]:
    torch.manual_seed(0)
    q = torch.randn(b, s, hq, d, device=dev).bfloat16(); k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    scale = 1 / math.sqrt(d)
    o, lse = ops.raw_attn_fwd(q, k, v, scale, causal)
    do = torch.randn_like(o)
    fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
    tf = timeit(lambda: ops.raw_attn_fwd(q, k, v, scale, causal))
    tb = timeit(lambda: ops.raw_attn_bwd(q, k, v, o, lse, do, scale, causal))
    dq, dk, dv = ops.raw_attn_bwd(q, k, v, o, lse, do, scale, causal)
    out[name] = {"fwd_TF": round(fl / tf / 1e12), "bwd_TF": round(2.5 * fl / tb / 1e12), "bwd_ms": round(tb * 1e3, 3),
                 "dk_sum": dk.float().abs().sum().item(), "dv_sum": dv.float().abs().sum().item()}
print(json.dumps(out))
'''
for v in sys.argv[1:] or ["new"]:
    e = dict(os.environ)
    e["TAMD_ATTN_DKDV"] = v
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:], flush=True)
