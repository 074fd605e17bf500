"""The weight gradients of one layer's dense layers as ONE grouped launch (tamd_gemm_group / torch.ops.tamd.gemm_dw_group)
against one product each (the default dispatch: split-K where the policy picks it) and torch (hipBLASLt, `dy.t() @ x`):
bert-base / bert-large at batch 32 x 512, the CLIP-L tower at 4 images, GPT-2 at batch 8 x 1024.  Interleaved, min of 3 rounds.
   python tools/gemm_group_ab.py > gpurun_out/<tag>_gemm_group_ab.jsonl"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def layer(h, inter, tokens):  # (M = out features, N = in features, K = tokens) of o, fc1, fc2, q|k|v
    return [(h, h, tokens), (inter, h, tokens), (h, inter, tokens), (3 * h, h, tokens)]


CASES = {"bert-base 32x512": layer(768, 3072, 16384), "bert-large 32x512": layer(1024, 4096, 16384),
         "clip-L 4x577 (padded 2368)": layer(1024, 4096, 2368), "gpt2 8x1024": layer(768, 3072, 8192),
         "bert-base 8x128": layer(768, 3072, 1024)}
for name, shapes in CASES.items():
    torch.manual_seed(0)
    dys = [torch.randn(k, m, device=dev).bfloat16() for (m, n, k) in shapes]
    xs = [(torch.randn(k, n, device=dev) * 0.1).bfloat16() for (m, n, k) in shapes]
    flops = sum(2.0 * m * n * k for (m, n, k) in shapes)
    arms = {
        "group": lambda: torch.ops.tamd.gemm_dw_group(dys, xs),
        "singles": lambda: [ops.raw_gemm(a, b, a_km=True, b_kn=True) for a, b in zip(dys, xs)],
        "torch": lambda: [a.t() @ b for a, b in zip(dys, xs)],
    }
    t = {a: [] for a in arms}
    for _ in range(3):
        for a, fn in arms.items():
            t[a].append(timeit(fn))
    g = torch.ops.tamd.gemm_dw_group(dys, xs)
    s1 = [ops.raw_gemm(a, b, a_km=True, b_kn=True, sched="fl") for a, b in zip(dys, xs)]
    err = max(float(((u.float() - v.float()).norm() / v.float().norm())) for u, v in zip(g, s1))
    rec = {"case": name, "shapes": shapes, "gflop": round(flops / 1e9, 1), "max_rel_diff_vs_unsplit": err}
    for a in arms:
        rec[a + "_us"] = round(min(t[a]), 1)
        rec[a + "_TF"] = round(flops / min(t[a]) / 1e6, 1)
    print(json.dumps(rec), flush=True)
