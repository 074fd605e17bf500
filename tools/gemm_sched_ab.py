"""A/B of the GEMM schedules (ops.raw_gemm(..., sched=...)) on the Llama-3-8B shapes for the forward product and the
two backward products (k-major operand layouts), with torch.mm (hipBLASLt) beside them; interleaved rounds so
clock/thermal drift hits every variant alike.
    python tools/gemm_sched_ab.py [rounds] [fwd,dx,dw]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from transformers_amd import ops

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = (sys.argv[2] if len(sys.argv) > 2 else "fwd,dx,dw").split(",")


def timeit(fn, iters=10, warm=2):
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


SHAPES = [("qkv", 32768, 6144, 4096), ("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096),
          ("down", 32768, 4096, 14336), ("lm_head", 32768, 128256, 4096)]
for name, m, n, k in SHAPES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    dy = torch.randn(m, n, device=dev).bfloat16()
    fl = 2.0 * m * n * k
    prods = {
        "fwd": (("pp", "fl"), lambda s: ops.raw_gemm(x, w, sched=s), lambda: torch.mm(x, w.t())),
        "dx": (("pp", "fl"), lambda s: ops.raw_gemm(dy, w, b_kn=True, sched=s), lambda: torch.mm(dy, w)),
        "dw": (("pp", "fl"), lambda s: ops.raw_gemm(dy, x, a_km=True, b_kn=True, sched=s), lambda: torch.mm(dy.t(), x)),
    }
    for prod in which:
        scheds, ours, ref = prods[prod]
        res = {}
        for r in range(rounds):
            for sched in scheds:
                res.setdefault(sched, []).append(round(fl / timeit(lambda: ours(sched)) / 1e12))
            res.setdefault("torch", []).append(round(fl / timeit(ref) / 1e12))
        base = ours("pp")
        same = {s: bool(torch.equal(ours(s), base)) for s in scheds[1:]}
        err = ((base.float() - ref().float()).norm() / ref().float().norm()).item()
        print(json.dumps({"shape": name, "product": prod, "mnk": [m, n, k], "TFLOPs": res, "bit_identical_to_pp": same,
                          "rel_err_vs_torch": round(err, 5)}), flush=True)
        del base
    del x, w, dy
