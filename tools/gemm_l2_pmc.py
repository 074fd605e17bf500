"""A handful of launches of the gate|up forward GEMM (ours, then torch.mm) for a rocprofv3 --pmc pass:
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/l2 -o l2 -- python tools/gemm_l2_pmc.py"""
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
m, n, k = 32768, 28672, 4096
x = torch.randn(m, k, device=dev).bfloat16()
w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
dy = torch.randn(m, n, device=dev).bfloat16()
for _ in range(3):
    ops.raw_gemm(x, w)
    ops.raw_gemm(dy, w, b_kn=True)
    ops.raw_gemm(dy, x, a_km=True, b_kn=True)
    torch.mm(x, w.t())
torch.cuda.synchronize()
