"""Launch the attention kernels at the Llama-3-8B shape for rocprofv3."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
b, s, hq, hkv, d = 8, 4096, 32, 8, 128
q = torch.randn(b, s, hq, d, device=dev).bfloat16()
k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
for _ in range(4):
    o, lse = ops.raw_attn_fwd(q, k, v, d ** -0.5, True)
    do = torch.randn_like(o)
    ops.raw_attn_bwd(q, k, v, o, lse, do, d ** -0.5, True)
torch.cuda.synchronize()
