"""Ablation timing of gemm_fl_kernel (tamd_gemm_set_dbg bit mask: 1 no LDS-DMA after the prologue, 2 no LDS fragment
reads, 4 no vmcnt wait at the hand-off, 8 no barrier; results are wrong by construction) on forward shapes, interleaved
rounds in one process, with the clock probe: TFLOP/s, clock, MFMA busy."""
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
variants = [int(v) for v in sys.argv[1:]] or [0, 4, 8, 12, 1, 2, 15]
for name, m, n, k in [("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096), ("down", 32768, 4096, 14336)]:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    wgs = (m // 256) * (n // 256)
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            lib.tamd_gemm_set_dbg(v)
            for _ in range(2):
                ops.raw_gemm(x, w)
            buf = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(6):
                ops.raw_gemm(x, w)
            e.record()
            torch.cuda.synchronize()
            lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(buf.data_ptr()))
            ops.raw_gemm(x, w)
            torch.cuda.synchronize()
            lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(0))
            t = buf.cpu().view(wgs, 2).double()
            res[v].append((round(2.0 * m * n * k * 6 / s.elapsed_time(e) / 1e9), round((t[:, 0] / t[:, 1]).mean().item() * 0.1, 2),
                           round(((k // 64) * 2048 / t[:, 0]).mean().item(), 3)))
    lib.tamd_gemm_set_dbg(0)
    print(json.dumps({"shape": name, "TF_clock_busy_by_dbg": {str(v): res[v] for v in variants}}), flush=True)
