"""Per-phase cycle breakdown of the ping-pong GEMM (workgroup 0) from in-kernel s_memtime stamps."""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import _diag  # noqa: E402
from transformers_amd import ops  # noqa: E402

_diag.use_diag()

m, n, k = 32768, 4096, 4096
dev = torch.device("cuda:0")
x = torch.randn(m, k, device=dev).bfloat16()
w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
trace = torch.zeros(8 * 32 * 8, dtype=torch.int64, device=dev)
lib = ops.backend().lib
P = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    rc = lib.tamd_gemm_trace(P(x), P(w), P(c), m, n, k, P(trace), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
t = trace.cpu().view(8, 32, 8).double()
names = ["ds_read issue", "glds issue", "vmcnt(8) wait", "lgkmcnt(0) wait", "barrier A", "16 MFMA", "barrier B"]
for wv in (0, 3, 4, 7):
    d = t[wv, 8:30, 1:] - t[wv, 8:30, :-1]
    tot = (t[wv, 9:31, 0] - t[wv, 8:30, 0]).mean().item()
    print(f"wave {wv}: " + ", ".join(f"{nm} {d[:, i].mean().item():.0f}" for i, nm in enumerate(names)) + f" | sub-tile period {tot:.0f} ticks")
ref = x[:256].float() @ w[:256].float().t()
print("check", ((c[:256, :256].float() - ref).norm() / ref.norm()).item())
