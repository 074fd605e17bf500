"""Timeline of one gemm_fl_kernel launch (diagnostic build): every workgroup stamps the 100 MHz real-time counter at entry and
exit plus its XCC id.  Reports, per shape: kernel time, the sum of workgroup busy time / (256 CUs x makespan) = how full the
machine was, the ramp (first start -> last first-round start), the tail (first CU idle -> last exit), and per XCD the time
its last workgroup finished -- do the eight XCDs finish together?      python tools/gemm_timeline.py > gpurun_out/<tag>_gemm_timeline.jsonl"""
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
SHAPES = [("qkv", T, 6144, 4096), ("o_proj", T, 4096, 4096), ("gate_up", T, 28672, 4096), ("down", T, 4096, 14336),
          ("bert fc1", 16384, 3072, 768), ("bert mlm", 16384, 30528, 768)]
for name, m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    wgs = -(-m // 256) * -(-n // 256)
    buf = torch.zeros(3 * wgs, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.raw_gemm(x, w, sched="fl")
    torch.cuda.synchronize()
    assert lib.tamd_gemm_set_timeline_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.raw_gemm(x, w, sched="fl")
    e1.record()
    torch.cuda.synchronize()
    lib.tamd_gemm_set_timeline_buffer(ctypes.c_void_p(0))
    t = buf.cpu().view(wgs, 3)
    start, end, xcc = t[:, 0].double() * 0.01, t[:, 1].double() * 0.01, t[:, 2]  # us
    t0 = start.min()
    start, end = start - t0, end - t0
    span = end.max().item()
    busy = (end - start).sum().item()
    # when does the machine stop being full: the (wgs - 255)-th largest end time is when the first CU runs out of work
    ends = end.sort().values
    first_idle = ends[-256].item() if wgs >= 256 else 0.0
    starts = start.sort().values
    row = {"shape": name, "m": m, "n": n, "k": k, "wgs": wgs, "kernel_us_events": round(e0.elapsed_time(e1) * 1e3, 1),
           "makespan_us": round(span, 1), "fill": round(busy / (256 * span), 4),
           "wg_us_mean": round((end - start).mean().item(), 2), "wg_us_min_max": [round((end - start).min().item(), 2), round((end - start).max().item(), 2)],
           "ramp_us": round(starts[min(255, wgs - 1)].item(), 2), "tail_us": round(span - first_idle, 2),
           "xcd_last_end_us": [round(end[xcc == i].max().item(), 1) if (xcc == i).any() else None for i in range(8)],
           "xcd_wgs": [int((xcc == i).sum()) for i in range(8)],
           "xcd_wg_us_mean": [round((end - start)[xcc == i].mean().item(), 2) if (xcc == i).any() else None for i in range(8)]}
    print(json.dumps(row), flush=True)
    del x, w
