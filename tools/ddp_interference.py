"""What does sharing the GPU with the gradient all-reduce cost the backward?  (no 8-GPU node exists,
and RCCL at world size 1 moves nothing, so the all-reduce is stood in for by what it IS on the compute side: a handful of
workgroups -- one per RCCL channel -- that sit on their CUs for the whole transfer and stream gradient bytes.)

The GEMM takes a whole CU per workgroup (160 KiB LDS, 512 registers per lane: nothing co-resides), so every CU a
channel occupies is a CU the tile grid does not get.  This tool measures that on ONE GPU:

    proxy      a 2 GiB device-to-device copy loop on a side stream created with hipExtStreamCreateWithCUMask(N CUs)
    compute    one Llama-3-8B decoder layer forward+backward (batch 8 x 4096), the bench's unit of work

legs: alone | with the proxy on N = 8, 16, 32 masked CUs | the same with the compute stream masked to the other 256 - N
CUs ("--reserve-cus": the GEMM never waits for a CU a channel holds; its grid is quantised over 256 - N CUs instead).

    python tools/ddp_interference.py [--iters 6] > gpurun_out/r03_ddp_interference.jsonl
"""
import argparse
import ctypes
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--cus", default="8,16,32")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(bits):
    """A HIP stream whose kernels may only run on the CUs whose bit is set (bit i = CU i of the device enumeration)."""
    words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def build_layer():
    import transformers_amd
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding
    from transformers_amd.patch import _tables

    cfg = LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=1,
                      num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, max_position_embeddings=8192,
                      rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, attn_implementation="eager")
    torch.manual_seed(0)
    with torch.device(dev):
        layer = LlamaDecoderLayer(cfg, 0).bfloat16()
        rot = LlamaRotaryEmbedding(cfg)
    for p in layer.parameters():
        torch.nn.init.normal_(p, std=0.02) if p.dim() > 1 else None
    x = torch.randn(8, 4096, 4096, device=dev).bfloat16().requires_grad_(True)
    pe = rot(x, torch.arange(4096, device=dev)[None])
    transformers_amd.attention.register()
    cfg._attn_implementation = "tamd"
    for m in layer.modules():
        r = _tables().get(type(m))
        if r is not None:
            m.__class__ = r

    def step():
        y = layer(x, position_embeddings=pe)
        y.backward(x.detach())
        layer.zero_grad(set_to_none=True)
        x.grad = None

    return step


step = build_layer()
src = torch.empty(1 << 30, dtype=torch.bfloat16, device=dev)  # 2 GiB
dst = torch.empty_like(src)


def measure(compute_stream, proxy_stream):
    """ms per layer fwd+bwd on `compute_stream` while `proxy_stream` (or nothing) copies; + the proxy's GB/s."""
    torch.cuda.synchronize()
    with torch.cuda.stream(compute_stream):
        for _ in range(2):
            step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    copies = 0
    if proxy_stream is not None:
        with torch.cuda.stream(proxy_stream):
            p0.record()
            for _ in range(64):  # (more than the compute leg can outlast; the tail is cut by the final sync below)
                dst.copy_(src, non_blocking=True)
                copies += 1
            p1.record()
    with torch.cuda.stream(compute_stream):
        e0.record()
        for _ in range(args.iters):
            step()
        e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    torch.cuda.synchronize()
    gbps = (copies * 2 * src.numel() * 2 / (p0.elapsed_time(p1) * 1e-3) / 1e9) if proxy_stream is not None else None
    return ms, gbps


main = torch.cuda.current_stream()
base, _ = measure(main, None)
print(json.dumps({"leg": "alone", "ms_per_layer_fwd_bwd": base, "cus": NCU}), flush=True)
for n in (int(v) for v in args.cus.split(",")):
    bits = [i * (NCU // n) for i in range(n)]  # spread over the device enumeration (all XCDs)
    try:
        proxy = masked_stream(bits)
        ms, gbps = measure(main, proxy)
        print(json.dumps({"leg": f"proxy on {n} CUs", "ms_per_layer_fwd_bwd": ms, "slowdown": ms / base - 1,
                          "proxy_GBps_rw": gbps}), flush=True)
        rest = masked_stream([i for i in range(NCU) if i not in set(bits)])
        ms2, gbps2 = measure(rest, proxy)
        print(json.dumps({"leg": f"proxy on {n} CUs, compute masked to the other {NCU - n}", "ms_per_layer_fwd_bwd": ms2,
                          "slowdown": ms2 / base - 1, "proxy_GBps_rw": gbps2}), flush=True)
        ms3, _ = measure(rest, None)
        print(json.dumps({"leg": f"compute alone on {NCU - n} CUs", "ms_per_layer_fwd_bwd": ms3,
                          "slowdown": ms3 / base - 1}), flush=True)
    except Exception as e:  # diagnostics must not take the visit down
        print(json.dumps({"leg": f"{n} CUs", "error": repr(e)}), flush=True)
