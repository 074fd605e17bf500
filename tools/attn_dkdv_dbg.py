"""Ablation timing of the dK/dV kernel (TAMD_DKDV_DBG bit mask: 1 no softmax math, 2 no LDS fragment reads,
4 no MFMA, 8 no tile loads, 16 no barrier; results are wrong by construction) at the Llama-3-8B shape."""
import json, os, subprocess, sys
code = r'''
import sys, torch, json, os, math
sys.path.insert(0, ".")
from transformers_amd import ops
sys.path.insert(0, "tools")
import _diag
_diag.use_diag()
dev = torch.device("cuda:0")
b, s, hq, hkv, d = 8, 4096, 32, 8, 128
torch.manual_seed(0)
q = torch.randn(b, s, hq, d, device=dev).bfloat16(); k = torch.randn(b, s, hkv, d, device=dev).bfloat16(); v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
scale = 1 / math.sqrt(d)
o, lse = ops.raw_attn_fwd(q, k, v, scale, True)
do = torch.randn_like(o)
def run(): ops.raw_attn_bwd(q, k, v, o, lse, do, scale, True)
for _ in range(3): run()
torch.cuda.synchronize()
st, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print(json.dumps({"dbg": os.environ.get("TAMD_DKDV_DBG", "0"), "bwd_ms": round(st.elapsed_time(e) / 10, 3)}))
'''
for v in sys.argv[1:] or ["0", "1", "2", "4", "8", "16", "3", "6", "7"]:
    e = dict(os.environ); e["TAMD_DKDV_DBG"] = v
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
