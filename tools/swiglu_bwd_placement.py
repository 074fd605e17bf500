"""Does the placement of its operands decide which regime the fused SwiGLU-backward GEMM runs in?
(profiles/r02_regression_note.md: `tamd_gemm_swiglu_bwd` takes 3.6 ms on fresh tensors and in a one-layer loop, 5.7 ms
inside the 32-layer model on most boxes; same ISA, same arguments.)

One process, one box, the Llama-3-8B shape (T=32768, hidden 4096, I=14336).  Legs, each timed with HIP events over
`--iters` launches and reported next to the two-kernel path (GEMM + swiglu_bwd_kernel) on the same tensors:

  fresh       operands allocated first thing (what the micro-benchmark measures)
  ballast     --ballast-gb of 1.88 GB blocks allocated first (the 32 layers' saved gate|up), operands after them
  holes       every other ballast block freed, emptied from the caching allocator, operands allocated into the holes
  interleave  one ballast block between every two operand allocations
  rotate      a different (gu, d_gu, act) set per launch, cycling over 8 sets (what the backward of 32 layers does)

    python tools/swiglu_bwd_placement.py [--ballast-gb 120] [--iters 12] [--legs fresh,ballast,...]

Under `rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum` (tools/gpu_r03_tlb.sh) the same legs give
the address-translation miss rate per launch."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ballast-gb", type=float, default=120.0)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--legs", default="fresh,ballast,holes,interleave,rotate")
args = ap.parse_args()
dev = torch.device("cuda:0")
T, H, I = 32768, 4096, 14336
BLOCK = T * 2 * I * 2                                                     # one saved gate|up, bytes


def operands(between=None):
    """(dy, wd, gu) in allocation order; `between()` is called between the allocations."""
    out = []
    for shape, scale in (((T, H), 1.0), ((H, I), 0.02), ((T, 2 * I), 1.0)):
        out.append(torch.empty(*shape, device=dev, dtype=torch.bfloat16).normal_(0.0, scale))
        if between:
            between()
    return out


def time_pair(sets, label):
    """sets: list of (dy, wd, gu); launches cycle over them."""
    def run(fn):
        for i in range(3):
            fn(*sets[i % len(sets)])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(args.iters):
            fn(*sets[i % len(sets)])
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / args.iters

    def two_kernel(dy, wd, gu):
        return ops.raw_swiglu_bwd(gu, ops.raw_gemm(dy, wd, b_kn=True), want_act=True)

    fused = run(ops.raw_gemm_swiglu_bwd)
    split = run(two_kernel)
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"leg": label, "fused_ms": round(fused, 3), "gemm_plus_swiglu_bwd_ms": round(split, 3),
                      "sets": len(sets), "hbm_in_use_gb": round((total - free) / 2 ** 30, 1),
                      "gu_ptr": hex(sets[0][2].data_ptr())}), flush=True)


legs = args.legs.split(",")
nblocks = int(args.ballast_gb * 2 ** 30 / BLOCK)
if "fresh" in legs:
    s = operands()
    time_pair([s], "fresh")
    del s
    torch.cuda.empty_cache()
ballast = [torch.empty(BLOCK, dtype=torch.uint8, device=dev) for _ in range(nblocks)]
for b in ballast:
    b.zero_()                                                              # touch: physical pages behind every block
if "ballast" in legs:
    s = operands()
    time_pair([s], "ballast")
    del s
    torch.cuda.empty_cache()
if "holes" in legs:
    kept = ballast[0::2]
    ballast = None
    torch.cuda.empty_cache()                                               # holes of one block each go back to the driver
    s = operands()
    time_pair([s], "holes")
    del s
    ballast = kept + [torch.empty(BLOCK, dtype=torch.uint8, device=dev) for _ in range(nblocks - len(kept))]
    torch.cuda.empty_cache()
if "interleave" in legs:
    extra = []
    s = operands(lambda: extra.append(torch.empty(BLOCK // 7, dtype=torch.uint8, device=dev).zero_()))
    time_pair([s], "interleave")
    del s, extra
    torch.cuda.empty_cache()
if "rotate" in legs:
    del ballast[8:]
    torch.cuda.empty_cache()
    dy, wd, gu = operands()
    sets = [(dy, wd, gu)] + [(dy, wd, torch.empty_like(gu).normal_()) for _ in range(7)]
    time_pair(sets, "rotate")
