"""Weight-gradient products dW[M, N] = dY[T, M]^T . X[T, N] of Llama-3-8B at T = 32768 tokens, two ways, interleaved in one process:

  whole     ONE product under the library's split-K policy (`torch.ops.tamd.gemm_out`: q|k|v 384 tiles and down_proj 896 tiles are
            split in two along K -- every workgroup half the tokens, fp32 partial tiles of the whole output, a reduction launch)
  balanced  `torch.ops.tamd.gemm` = gemm_dw_balanced (csrc/torch_binding.cpp, round 5): a part whose tile grid is a whole number of
            dispatch rounds, unsplit, and a remainder of at most half a round, which split-K fills

    python tools/gemm_dw_cut_ab.py            -> one JSON line per product (us, TFLOP/s, max |difference|)"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import _native, ops  # noqa: E402,F401

dev = torch.device("cuda:0")
T = 32768


def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, m, n in (("dW q|k|v", 6144, 4096), ("dW down", 4096, 14336), ("dW lm_head", 128256, 4096), ("dW gate|up", 28672, 4096),
                   ("dW o", 4096, 4096)):
    torch.manual_seed(0)
    dy = torch.randn(T, m, device=dev).bfloat16()
    x = torch.randn(T, n, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    whole = lambda: torch.ops.tamd.gemm_out(out, dy, x, True, True)  # noqa: E731
    bal = lambda: torch.ops.tamd.gemm(dy, x, True, True)  # noqa: E731
    whole()
    got = bal()
    torch.cuda.synchronize()
    diff = (got.float() - out.float()).abs().max().item()
    tw, tb = [], []
    for _ in range(3):
        tw.append(timeit(whole))
        tb.append(timeit(bal))
    fl = 2.0 * m * n * T
    print(json.dumps({"product": name, "M": m, "N": n, "K": T, "cut": _native.dw_cut(m, n, T), "whole_us": round(min(tw), 1),
                      "balanced_us": round(min(tb), 1), "whole_TF": round(fl / min(tw) / 1e6), "balanced_TF": round(fl / min(tb) / 1e6),
                      "balanced_over_whole": round(min(tb) / min(tw), 4), "max_abs_diff": diff,
                      "all_us": {"whole": [round(v, 1) for v in tw], "balanced": [round(v, 1) for v in tb]}}), flush=True)
    del dy, x, out, got
