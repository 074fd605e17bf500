"""Which bf16 MFMA shape does more work inside MI355X's power budget?  Register-only MFMA streams (no LDS, no memory
traffic) on random operand bits and on zeros, 256 workgroups x 4 waves (one per SIMD), interleaved rounds."""
import ctypes
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402

lib = _diag.use_diag()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
blocks, iters = 256, 20000
flop = blocks * 4 * iters * 32 * 2 * 32 * 32 * 16
sink = torch.zeros(4, device=dev)
for fill in ("randn", "zeros", "randn"):
    src = (torch.randn(61 * 512, device=dev).bfloat16() if fill == "randn" else torch.zeros(61 * 512, device=dev).bfloat16())
    src = src.view(torch.int32)
    for rnd in range(3):
        for mode, name in ((0, "32x32x16"), (1, "16x16x32")):
            clk = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.tamd_mfma_power(P(src), iters, mode, blocks, P(clk), P(sink), st) == 0
            e1.record()
            torch.cuda.synchronize()
            t = clk.cpu().view(blocks, 2).double()
            ms = e0.elapsed_time(e1)
            busy = iters * 32 * 32 / t[:, 0]  # 32 MFMA-equivalents of 32 cycles per round
            print(json.dumps({"fill": fill, "mfma": name, "ms": round(ms, 2), "TF": round(flop / ms / 1e9),
                              "clock_GHz": round((t[:, 0] / t[:, 1]).mean().item() * 0.1, 3),
                              "issue_busy": round(busy.mean().item(), 3)}), flush=True)
