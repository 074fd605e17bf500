"""What clock do the 4-wave GEMM kernels run at, and how busy is the matrix pipe?  (diagnostic build: every workgroup
stamps the shader clock and the 100 MHz real-time counter around its K loop; include/tamd_diag.h)

    clock  = shader ticks / real-time ticks * 0.1 GHz
    busy   = stages * 64 MFMAs * 32 cycles / shader ticks          (one wave per SIMD: its MFMAs are the SIMD's)
"""
import ctypes
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402
from transformers_amd import ops  # noqa: E402

lib = _diag.use_diag()
dev = torch.device("cuda:0")
T = 32768
scheds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fl", "fr"]
for name, m, n, k in [("o_proj", T, 4096, 4096), ("gate_up", T, 28672, 4096), ("down", T, 4096, 14336)]:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    dy = torch.randn(m, n, device=dev).bfloat16()
    for lay, call in (("fwd", lambda s: ops.raw_gemm(x, w, sched=s)), ("dx", lambda s: ops.raw_gemm(dy, w, b_kn=True, sched=s)),
                      ("dw", lambda s: ops.raw_gemm(dy, x, a_km=True, b_kn=True, sched=s))):
        mm, nn, kk = {"fwd": (m, n, k), "dx": (m, k, n), "dw": (n, k, m)}[lay]
        wgs = (mm // 256) * (nn // 256)
        for s in scheds:
            buf = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
            for _ in range(3):
                call(s)
            torch.cuda.synchronize()
            assert lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(s)
            e1.record()
            torch.cuda.synchronize()
            lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(0))
            t = buf.cpu().view(wgs, 2).double()
            ghz = (t[:, 0] / t[:, 1] * 0.1)
            busy = (kk // 64) * 64 * 32 / t[:, 0]
            ms = e0.elapsed_time(e1)
            print(json.dumps({"shape": name, "layout": lay, "sched": s, "ms": round(ms, 3),
                              "TF": round(2.0 * m * n * k / ms / 1e9), "clock_GHz_mean": round(ghz.mean().item(), 3),
                              "clock_GHz_min_max": [round(ghz.min().item(), 3), round(ghz.max().item(), 3)],
                              "mfma_busy_mean": round(busy.mean().item(), 3),
                              "kloop_us_mean": round((t[:, 1].mean() * 0.01).item(), 2),
                              "kloop_share_of_kernel": round((t[:, 1].sum() * 0.01 / 256 / (ms * 1e3)).item(), 3)}), flush=True)
    del x, w, dy

# data dependence (DVFS): the same launches on zero-filled operands, ours and hipBLASLt's
m, n, k = T, 28672, 4096
for fill in ("randn", "zeros"):
    x = (torch.randn(m, k, device=dev) if fill == "randn" else torch.zeros(m, k, device=dev)).bfloat16()
    w = ((torch.randn(n, k, device=dev) * 0.02) if fill == "randn" else torch.zeros(n, k, device=dev)).bfloat16()
    out = {"shape": "gate_up", "fill": fill}
    for nm, fn in [(s, (lambda s=s: ops.raw_gemm(x, w, sched=s))) for s in scheds] + [("torch", lambda: torch.mm(x, w.t()))]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[nm] = round(2.0 * m * n * k * 8 / e0.elapsed_time(e1) / 1e9)
    print(json.dumps(out), flush=True)
