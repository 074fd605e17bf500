"""Effect of leading-dimension padding (L2 channel spreading) on the full-line GEMM: row strides that are multiples of
4 KiB put the same column chunk of every row on one L2 channel.  TFLOP/s for pad in elements on A / B / both."""
import json, sys, torch
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
def padded(rows, cols, pad, scale=1.0):
    t = (torch.randn(rows, cols + pad, device=dev) * scale).bfloat16()
    return t[:, :cols]
for name, m, n, k in [("o_proj", 32768, 4096, 4096), ("gate_up", 32768, 28672, 4096), ("down", 32768, 4096, 14336)]:
    fl = 2.0 * m * n * k
    out = {"shape": name}
    for pa, pb in [(0, 0), (128, 0), (0, 128), (128, 128), (64, 64), (256, 256)]:
        x, w, dy = padded(m, k, pa), padded(n, k, pb, 0.02), padded(m, n, pa)
        out[f"fwd a{pa} b{pb}"] = round(fl / timeit(lambda: ops.raw_gemm(x, w)) / 1e12)
        out[f"dx a{pa} b{pb}"] = round(fl / timeit(lambda: ops.raw_gemm(dy, w, b_kn=True)) / 1e12)
        out[f"dw a{pa} b{pb}"] = round(fl / timeit(lambda: ops.raw_gemm(dy, x, a_km=True, b_kn=True)) / 1e12)
        del x, w, dy
    print(json.dumps(out), flush=True)
