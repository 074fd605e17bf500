"""A/B of the two LDS-DMA piece placements of the full-line GEMM kernel on the Llama-3-8B shapes, interleaved rounds in one
process (TFLOP/s per arm and round):

  fl              the product kernel (early pieces whenever A is row-major: forward and dX; late for dW)
  early / late    the other placement (diagnostic library: tamd_gemm_set_dbg 32 / 128)

(profiles/r03b_gemm_persist_ab.jsonl was written by this tool when it still carried the two arms of the persistent,
XCD-aligned walk -- profiles/r03b_gemm_persist.patch.)

Split-K products (q|k|v and down dW) go through tamd_gemm_ws with the workspace the policy asks for.

    python tools/gemm_piece_ab.py [--rounds 3] [--iters 6] [--shapes qkv,o_proj,gate_up,down,lm_head] [--legs fwd,dX,dW]
"""
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
ap.add_argument("--shapes", default="qkv,o_proj,gate_up,down,lm_head")
ap.add_argument("--legs", default="fwd,dX,dW")
ap.add_argument("--no-diag", action="store_true", help="product library only (no early / late arms)")
ap.add_argument("--placements", action="store_true", help="also the round-5 placements (tamd_gemm_set_dbg 64 / 256 / 512)")
ap.add_argument("--three-barrier", action="store_true", help="round 6: hipBLASLt's loop structure (tamd_gemm_set_dbg 1024), every layout")
ap.add_argument("--only", default="", help="comma-separated arms to keep beside fl (e.g. b3)")
args = ap.parse_args()
lib = None if args.no_diag else _diag.use_diag()
be = ops.backend()
dev = torch.device("cuda:0")
T = 32768
SHAPES = {"qkv": (T, 6144, 4096), "o_proj": (T, 4096, 4096), "gate_up": (T, 28672, 4096), "down": (T, 4096, 14336),
          "lm_head": (T, 128256, 4096)}


def gemm(a, b, flags, m, n, k, out, ws):
    be.lib.check(be.lib.tamd_gemm_ws(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, m, n, k, a.stride(0),
                                     b.stride(0), out.stride(0), 0, flags, ops.EPI_NONE, 0, ops._code(a),
                                     ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                     be.stream(a)), "tamd_gemm_ws")


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
    # (operands, layout flags, GEMM M N K) of the three products of a linear layer
    legs = {"fwd": (x, w, 0, m, n, k), "dX": (dy, w, 2, m, k, n), "dW": (dy, x, 3, n, k, m)}
    for leg in args.legs.split(","):
        a, b, lay, gm, gn, gk = legs[leg]
        out = torch.empty(gm, gn, dtype=torch.bfloat16, device=dev)
        ws_bytes = be.lib.tamd_gemm_workspace_bytes(gm, gn, gk, lay, ops.EPI_NONE)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        arms = ["fl"]
        if lib is not None and ws is None:
            arms.append("early" if lay == 3 else "late")
            if lay != 3:
                arms.append("early")  # (forward layout: the product schedule until round 5; dX: still the product schedule)
            if lay != 3 and args.placements:  # round 5: pieces behind the even pairs / pieces first / pieces first + split hand-off
                arms += ["p1", "p2", "p3"]
            if args.three_barrier:
                arms += ["b3", "b1", "b3p", "b3s", "b3c", "pw"]  # (three barriers (vendor table, one MFMA per gap) / the one-barrier ring, forced: one of them is what "fl" runs; the table rounded to MFMA pairs / SPREAD)
            if args.only:
                arms = ["fl"] + [c for c in arms[1:] if c in args.only.split(",")]
        res = {c: [] for c in arms}
        ref = None
        for rnd in range(args.rounds):
            for c in arms:
                if lib is not None:
                    lib.tamd_gemm_set_dbg({"late": 128, "early": 32, "p1": 64, "p2": 256, "p3": 512, "b3": 1024, "b1": 2048, "b3p": 1024 + 4096, "b3c": 1024 + 16384, "b3s": 1024 + 8192}.get(c, 0))
                flags = lay
                fn = lambda: gemm(a, b, flags, gm, gn, gk, out, ws)  # noqa: E731
                res[c].append(round(2.0 * gm * gn * gk / time_ms(fn) / 1e9))
                if rnd == 0:
                    if ref is None:
                        ref = out.clone()
                    elif not torch.equal(out, ref):
                        res[c].append("MISMATCH")
        if lib is not None:
            lib.tamd_gemm_set_dbg(0)
        med = {c: sorted(v for v in vals if not isinstance(v, str))[len(vals) // 2] for c, vals in res.items()}
        print(json.dumps({"shape": name, "leg": leg, "mnk": [gm, gn, gk], "split_k": bool(ws_bytes), "tflops": res,
                          "median_vs_fl": {c: round(med[c] / med["fl"] - 1, 4) for c in arms}}), flush=True)
        del out, ws
    del x, w, dy
