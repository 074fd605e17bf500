"""Leading-dimension padding of A, B and C on the forward products of a Llama-3-8B layer (row strides that are multiples of 4 KiB
put the same column chunk of every row on one memory channel): TFLOP/s per (pad A, pad B, pad C) in elements, interleaved.
    python tools/gemm_ld_ab.py > gpurun_out/<tag>_gemm_ld_ab.jsonl"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def padded(rows, cols, pad, scale=1.0):
    t = (torch.randn(rows, cols + pad, device=dev) * scale).bfloat16()
    return t[:, :cols]


arms = [(0, 0, 0), (64, 0, 0), (0, 64, 0), (0, 0, 64), (64, 64, 0), (64, 64, 64), (128, 128, 128), (0, 0, 0)]
for name, m, n, k in [("qkv", 32768, 6144, 4096), ("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096),
                      ("down", 32768, 4096, 14336)]:
    fl = 2.0 * m * n * k
    rec = {"shape": name, "mnk": [m, n, k], "arms (pad A, B, C)": {}}
    for i, (pa, pb, pc) in enumerate(arms):
        x, w = padded(m, k, pa), padded(n, k, pb, 0.02)
        c = padded(m, n, pc)
        t = timeit(lambda: ops.raw_gemm(x, w, out=c))
        rec["arms (pad A, B, C)"][f"{pa},{pb},{pc}" + ("" if i < len(arms) - 1 else " again")] = round(fl / t / 1e12)
        del x, w, c
    print(json.dumps(rec), flush=True)
