"""A/B of the staggered K start (tamd_gemm_set_stagger, include/tamd_diag.h; DBG=16 instantiations of gemm_fl_kernel)
against the product schedule on the Llama-3-8B GEMM shapes, every layout, interleaved rounds in one process.

Why: hipBLASLt's gfx950 256x256x64 kernel (the one torch.mm picks for these shapes) starts each workgroup's K loop at a
different offset and wraps (its StaggerU code is visible at the top of its main loop); ours sweep k in lockstep, so all
256 workgroups ask for lines with the same low address bits at the same moment.

    python tools/gemm_stagger_ab.py [--rounds 3] [--iters 6] [--shapes o_proj,gate_up,down,qkv]

One JSON line per (shape, layout): TFLOP/s per round for every configuration "mode/units/stride" ("off" = product);
a trailing "e" = with the LDS-DMA pieces issued early in every k-step (tamd_gemm_set_dbg(32)), a trailing "b" = with a
second barrier per k-step (tamd_gemm_set_dbg(64)); row-major layout only: the other differences to hipBLASLt's loop, see
profiles/r02_gemm_variants.md section 5.  Every arm gives correct results."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402
from transformers_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--shapes", default="o_proj,gate_up,down,qkv")
ap.add_argument("--configs", default="2/32/2,3/32/2,1/8/0,1/8/1,1/8/4,2/8/1,2/8/2,2/32/1,2/16/0,3/4/2,3/16/1,4/8/1,4/32/2")
args = ap.parse_args()
lib = _diag.use_diag()
dev = torch.device("cuda:0")
T = 32768
SHAPES = {"qkv": (T, 6144, 4096), "o_proj": (T, 4096, 4096), "gate_up": (T, 28672, 4096), "down": (T, 4096, 14336)}
configs = [None] + [tuple(int(v) for v in c.split("/")) for c in args.configs.split(",")]
EARLY = ["e", "2/32/2e", "3/32/2e", "1/8/1e", "b", "2/32/2b"]              # forward layout only ("b": second barrier)


def key_of(c):
    return "off" if c is None else c if isinstance(c, str) else "/".join(map(str, c))


def select(c):
    early = isinstance(c, str)
    st = tuple(int(v) for v in c[:-1].split("/")) if early and len(c) > 1 else (c if not early and c else (0, 0, 0))
    lib.tamd_gemm_set_stagger(*st)
    lib.tamd_gemm_set_dbg((64 if c.endswith("b") else 32) if early else 0)


def time_ms(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / args.iters


for name in args.shapes.split(","):
    m, n, k = SHAPES[name]
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    dy = torch.randn(m, n, device=dev).bfloat16()
    legs = {"fwd": lambda: ops.raw_gemm(x, w), "dX": lambda: ops.raw_gemm(dy, w, b_kn=True),
            "dW": lambda: ops.raw_gemm(dy, x, a_km=True, b_kn=True)}
    for leg, fn in legs.items():
        arms = configs + (EARLY if leg == "fwd" else [])
        res = {key_of(c): [] for c in arms}
        for _ in range(args.rounds):
            for c in arms:
                select(c)
                res[key_of(c)].append(round(2.0 * m * n * k / time_ms(fn) / 1e9))
        select(None)
        best = max(res, key=lambda c: sorted(res[c])[len(res[c]) // 2])
        print(json.dumps({"shape": name, "layout": leg, "tflops": res, "best_median": best}), flush=True)
    del x, w, dy
