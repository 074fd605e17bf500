"""Where does the attention forward kernel's time go?  Per-phase shader-clock sums of workgroup 0 (the heaviest causal
query tile: 64 K/V tiles) from the diagnostic build's s_memtime stamps (include/tamd_diag.h tamd_attn_set_trace)."""
import ctypes
import json
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402
from transformers_amd import ops  # noqa: E402

lib = _diag.use_diag()
dev = torch.device("cuda:0")
names = ["load issue", "K.Q^T", "mask+softmax", "P.V", "vmcnt wait", "barrier"]
for name, b, s, hq, hkv, d, causal in [("llama3-8b", 8, 4096, 32, 8, 128, True)]:
    q = torch.randn(b, s, hq, d, device=dev).bfloat16()
    k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
    scale = 1 / math.sqrt(d)
    for _ in range(3):
        ops.raw_attn_fwd(q, k, v, scale, causal)
    buf = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.tamd_attn_set_trace(ctypes.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.raw_attn_fwd(q, k, v, scale, causal)
    e1.record()
    torch.cuda.synchronize()
    lib.tamd_attn_set_trace(ctypes.c_void_p(0))
    t = buf.cpu().view(4, 8).double()
    ntiles = s // 64
    out = {"shape": name, "kernel_ms": round(e0.elapsed_time(e1), 3), "tiles": ntiles}
    for w in range(4):
        out[f"wave{w}_cycles_per_tile"] = {names[i]: round(t[w, i].item() / ntiles) for i in range(6)}
        out[f"wave{w}_total_per_tile"] = round(t[w, :6].sum().item() / ntiles)
    print(json.dumps(out), flush=True)
