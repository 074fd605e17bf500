"""A/B of tamd_gemm of TWO builds of the C-ABI library in one process, interleaved on one GPU:
    python tools/gemm_lib_ab.py [base.so] [new.so]        (defaults: tools/ab/libtamd_base.so, transformers_amd/libtamd.so)
The Llama-3-8B products at 32768 tokens in the three layouts of a training step (forward x.W^T, dX = dY.W, dW = dY^T.X) and the
residual epilogue; per shape the time and TFLOP/s of each build (min of 3 interleaved rounds) and whether the results agree
bit for bit.  The base library is a build of an earlier commit (tools/build_base_lib.sh): no second code arm in the sources."""
import ctypes
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from transformers_amd._cabi import TamdLib, TAMD_BF16  # noqa: E402

base = TamdLib(Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "tools" / "ab" / "libtamd_base.so", accept_abi=(7, 8, 9))
new = TamdLib(Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "transformers_amd" / "libtamd.so")
dev = torch.device("cuda:0")
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def timeit(fn, iters, warm=2):
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


T = 32768
CASES = [  # name, layout flags, M, N, K, epilogue
    ("fwd qkv", 0, T, 6144, 4096, 0), ("fwd o+res", 0, T, 4096, 4096, 2), ("fwd down+res", 0, T, 4096, 14336, 2),
    ("dX qkv", 2, T, 4096, 6144, 0), ("dX gate|up", 2, T, 4096, 28672, 0), ("dX down", 2, T, 14336, 4096, 0),
    ("dW qkv", 3, 6144, 4096, T, 0), ("dW o", 3, 4096, 4096, T, 0), ("dW gate|up", 3, 28672, 4096, T, 0), ("dW down", 3, 4096, 14336, T, 0),
]
for name, flags, m, n, k, epi in CASES:
    torch.manual_seed(0)
    a = (torch.randn(k, m, device=dev) if flags & 1 else torch.randn(m, k, device=dev)).bfloat16()
    b = ((torch.randn(k, n, device=dev) if flags & 2 else torch.randn(n, k, device=dev)) * 0.05).bfloat16()
    res = torch.randn(m, n, device=dev).bfloat16() if epi == 2 else None
    outs = {}
    t = {"base": [], "new": []}

    def call(lib, c):
        st = lib.tamd_gemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), None, res.data_ptr() if res is not None else None, m, n, k,
                           a.stride(0), b.stride(0), n, n if res is not None else 0, flags, epi, 0, TAMD_BF16, stream)
        assert st == 0, (name, st)

    for key, lib in (("base", base), ("new", new)):
        outs[key] = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        call(lib, outs[key])
    torch.cuda.synchronize()
    iters = 6 if m * n * k > 2e12 else 12
    for _ in range(3):
        for key, lib in (("base", base), ("new", new)):
            t[key].append(timeit(lambda: call(lib, outs[key]), iters))
    fl = 2.0 * m * n * k
    print(json.dumps({"case": name, "M": m, "N": n, "K": k, "same_bits": bool(torch.equal(outs["base"], outs["new"])),
                      "base_us": round(min(t["base"]), 1), "new_us": round(min(t["new"]), 1),
                      "base_TF": round(fl / min(t["base"]) / 1e6, 1), "new_TF": round(fl / min(t["new"]) / 1e6, 1),
                      "new_over_base": round(min(t["new"]) / min(t["base"]), 4)}), flush=True)

# the gate|up projection with SiLU(gate) * up in its epilogue (tamd_gemm_swiglu): GU == NULL (forward only) and GU written (training)
I, K = 14336, 4096
torch.manual_seed(0)
x = torch.randn(T, K, device=dev).bfloat16()
wgu = (torch.randn(2 * I, K, device=dev) * 0.05).bfloat16()
for name, keep in (("fwd gate|up+SwiGLU (no GU)", False), ("fwd gate|up+SwiGLU (GU kept)", True)):
    outs, t = {}, {"base": [], "new": []}

    def call(lib, act, gu):
        st = lib.tamd_gemm_swiglu(x.data_ptr(), wgu.data_ptr(), gu.data_ptr() if gu is not None else None, act.data_ptr(), T, I, K,
                                  K, K, 2 * I, I, TAMD_BF16, stream)
        assert st == 0, (name, st)

    for key, lib in (("base", base), ("new", new)):
        outs[key] = (torch.empty(T, I, device=dev, dtype=torch.bfloat16),
                     torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16) if keep else None)
        call(lib, *outs[key])
    torch.cuda.synchronize()
    for _ in range(3):
        for key, lib in (("base", base), ("new", new)):
            t[key].append(timeit(lambda: call(lib, *outs[key]), 6))
    fl = 2.0 * T * 2 * I * K
    d = (outs["base"][0].float() - outs["new"][0].float()).abs()
    print(json.dumps({"case": name, "M": T, "N": 2 * I, "K": K, "same_bits": bool(torch.equal(outs["base"][0], outs["new"][0])),
                      "act_mismatch_fraction": float((outs["base"][0] != outs["new"][0]).float().mean()),
                      "act_max_abs_diff": float(d.max()),
                      "base_us": round(min(t["base"]), 1), "new_us": round(min(t["new"]), 1),
                      "base_TF": round(fl / min(t["base"]) / 1e6, 1), "new_TF": round(fl / min(t["new"]) / 1e6, 1),
                      "new_over_base": round(min(t["new"]) / min(t["base"]), 4)}), flush=True)
    del outs
