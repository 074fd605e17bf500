import sys, torch, json
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
m, n, k = 32768, 4096, 4096
for pad in (0, 8, 32, 64, 128, 264):
    x = torch.randn(m, k + pad, device=dev).bfloat16()[:, :k]
    w = (torch.randn(n, k + pad, device=dev) * 0.02).bfloat16()[:, :k]
    t = timeit(lambda: ops.raw_gemm(x, w))
    tt = timeit(lambda: torch.mm(x, w.t()))
    print(json.dumps({"pad": pad, "tamd_tflops": round(2.0*m*n*k/t/1e12, 1), "torch_mm_tflops": round(2.0*m*n*k/tt/1e12, 1)}), flush=True)
