"""Launch a few GEMMs of one shape (ours and torch.mm) for rocprofv3."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 4096, 4096)))
mode = sys.argv[4] if len(sys.argv) > 4 else "nt"
dev = torch.device("cuda:0")
x = torch.randn(m, k, device=dev).bfloat16()
w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
dy = torch.randn(m, n, device=dev).bfloat16()
for _ in range(6):
    if mode == "nt":
        ops.raw_gemm(x, w)
        torch.mm(x, w.t())
    elif mode == "dx":
        ops.raw_gemm(dy, w, b_kn=True)
    elif mode == "dw":
        ops.raw_gemm(dy, x, a_km=True, b_kn=True)
torch.cuda.synchronize()
