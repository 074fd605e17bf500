"""A handful of launches of our GEMM in each layout (and torch.mm) for a rocprofv3 --pmc pass:
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/l2 -o l2 -- python tools/gemm_l2_pmc.py
Round 2 measured forward 79.8 % / dX 65.1 % / dW 64.2 % L2 hits on the gate|up shape -- but there the dX / dW products
also have the long K (28672 / 32768 against 4096), so "k-major operand" and "long K loop: workgroups of a patch drift
apart over hundreds of stages and stop sharing through a 4 MiB L2" are confounded.  This version separates them: every
layout at a short and a long K (tools/gpu_l2.sh keys its table on kernel name AND grid size; the launch order is
printed)."""
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T = 32768
#        label            M      N      K
CASES = [("gate_up", T, 28672, 4096),       # fwd K=4096 (1792 x 8 tiles... grid 14336) | dX K=28672 | dW K=32768
         ("o_proj", T, 4096, 4096),         # fwd K=4096 (grid 2048) | dX K=4096 | dW K=32768
         ("down", T, 4096, 14336)]          # fwd K=14336 (grid 2048) | dX K=4096 (grid 7168) | dW K=32768
for label, m, n, k in CASES:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    dy = torch.randn(m, n, device=dev).bfloat16()
    print(f"{label}: fwd grid {(m // 256) * (n // 256)} K={k} | dX grid {(m // 256) * (k // 256)} K={n} | "
          f"dW grid {(n // 256) * (k // 256)} K={m} (split-K may multiply the dW grid)", flush=True)
    for _ in range(3):
        ops.raw_gemm(x, w)
        ops.raw_gemm(dy, w, b_kn=True)
        ops.raw_gemm(dy, x, a_km=True, b_kn=True)
        if label == "gate_up":
            torch.mm(x, w.t())
    torch.cuda.synchronize()
    del x, w, dy
