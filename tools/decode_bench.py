"""Decode attention (tamd_attn_decode: split-KV over a KV cache) on one MI355X.
  kernels   per shape: the split-KV schedule vs the training kernel (tamd_attn_fwd on the same call) vs torch SDPA -- us per call
            and the K+V bytes of the cache / time (the HBM roofline of a decode step's attention: every cached key and value
            is read once)
  generate  `model.generate` of a Llama-3-8B-shaped model (DECODE_BENCH_LAYERS layers, default 8, random init) after a 4096-token
            prompt: new tokens / s with attn_implementation="tamd" (and, in child processes, with the M = batch projections kept
            on the MFMA tiles, TAMD_GEMM=x, and with the training attention kernel, TAMD_DECODE_KERNEL=0) and "sdpa"
    python tools/decode_bench.py [kernels] [generate] > gpurun_out/<tag>_decode_bench.jsonl"""
import ctypes
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import transformers_amd  # noqa: E402
from transformers_amd import _cabi, ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def kernels():
    be = ops.backend()
    lib = be.lib
    shapes = [("llama3-8b b1 4k", 1, 1, 4096, 32, 8, 128), ("llama3-8b b1 16k", 1, 1, 16384, 32, 8, 128),
              ("llama3-8b b1 64k", 1, 1, 65536, 32, 8, 128), ("llama3-8b b8 8k", 8, 1, 8192, 32, 8, 128),
              ("llama3-8b b64 4k", 64, 1, 4096, 32, 8, 128), ("llama2-7b b1 4k (MHA)", 1, 1, 4096, 32, 32, 128),
              ("llama3-8b b1 4k, 8 rows", 1, 8, 4096, 32, 8, 128), ("gpt2 b1 1k", 1, 1, 1024, 12, 12, 64)]
    for name, b, sq, sk, hq, hkv, d in shapes:
        torch.manual_seed(0)
        q = torch.randn(b, sq, hq, d, device=dev).bfloat16()
        k = torch.randn(b, sk, hkv, d, device=dev).bfloat16()
        v = torch.randn(b, sk, hkv, d, device=dev).bfloat16()
        o = torch.empty_like(q)
        scale = d ** -0.5
        ap = _cabi.AttnParams()
        ap.q, ap.k, ap.v, ap.o, ap.lse, ap.key_valid, ap.q_start = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, None, None
        ap.batch, ap.seq_q, ap.heads_q, ap.head_dim, ap.seq_k, ap.heads_kv = b, sq, hq, d, sk, hkv
        for nm, t in (("q", q), ("k", k), ("v", v), ("o", o)):
            setattr(ap, f"{nm}_stride_b", t.stride(0))
            setattr(ap, f"{nm}_stride_s", t.stride(1))
            setattr(ap, f"{nm}_stride_h", t.stride(2))
        ap.scale, ap.causal, ap.dtype, ap.dropout_p, ap.dropout_seed, ap.q_prescaled = scale, 1, _cabi.TAMD_BF16, 0.0, 0, 0
        nbytes = lib.tamd_attn_decode_workspace_bytes(ctypes.byref(ap))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        stream = ctypes.c_void_p(be.stream(q))
        t_dec = timeit(lambda: lib.tamd_attn_decode(ctypes.byref(ap), ws.data_ptr(), nbytes, stream))
        o_dec = o.clone()
        t_trn = timeit(lambda: lib.tamd_attn_fwd(ctypes.byref(ap), stream), iters=10 if sk * b > 100000 else 50)
        qs, ks, vs = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        t_sdpa = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=False, enable_gqa=hq != hkv))
        ref = torch.nn.functional.scaled_dot_product_attention(qs.float(), ks.float(), vs.float(), enable_gqa=hq != hkv).transpose(1, 2) if sq == 1 else None
        kv_bytes = 2.0 * b * sk * hkv * d * 2
        print(json.dumps({"bench": "kernels", "shape": name, "b": b, "sq": sq, "sk": sk, "hq": hq, "hkv": hkv, "d": d,
                          "decode_us": round(t_dec, 1), "train_kernel_us": round(t_trn, 1), "sdpa_us": round(t_sdpa, 1),
                          "decode_GBps": round(kv_bytes / t_dec / 1e3), "sdpa_GBps": round(kv_bytes / t_sdpa / 1e3),
                          "err_vs_fp32": None if ref is None else round(((o_dec.float() - ref).norm() / ref.norm()).item(), 5)}), flush=True)


def generate(arm):
    from transformers import LlamaConfig, LlamaForCausalLM

    layers = int(os.environ.get("DECODE_BENCH_LAYERS", "8"))
    cfg = LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                      num_key_value_heads=8, max_position_embeddings=8192, attn_implementation="sdpa" if arm == "sdpa" else "eager")
    torch.manual_seed(0)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        model = LlamaForCausalLM(cfg).eval()
    torch.set_default_dtype(old)
    if arm != "sdpa":
        transformers_amd.accelerate(model)
    for b in [int(v) for v in os.environ.get("DECODE_BENCH_BATCHES", "1,8").split(",")]:
        ids = torch.randint(0, 128000, (b, 4096), device=dev)
        new = 48
        with torch.no_grad():
            gen = dict(do_sample=False, pad_token_id=0)
            if os.environ.get("DECODE_BENCH_CACHE"):  # e.g. "static": a pre-allocated cache (no torch.cat per layer and step)
                gen["cache_implementation"] = os.environ["DECODE_BENCH_CACHE"]
            model.generate(ids[:, :256], max_new_tokens=4, **gen)  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(ids, max_new_tokens=1, **gen)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.generate(ids, max_new_tokens=new + 1, min_new_tokens=new + 1, **gen)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        dec = (t2 - t1) - (t1 - t0)  # the decode steps alone (both runs include the same prefill)
        print(json.dumps({"bench": "generate", "arm": arm, "batch": b, "prompt": 4096, "new_tokens": new, "layers": layers, "cache": os.environ.get("DECODE_BENCH_CACHE", "dynamic"),
                          "prefill_s": round(t1 - t0, 4), "decode_ms_per_token": round(dec / new * 1e3, 3),
                          "new_tokens_per_s": round(b * new / dec, 1),
                          "fallbacks": transformers_amd.fallback_calls() if arm != "sdpa" else None}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["kernels", "generate"]
    if "kernels" in which:
        kernels()
    if "generate" in which:
        arm = os.environ.get("DECODE_BENCH_ARM")
        if arm:
            if "TAMD_GEMM" in os.environ:  # (schedule switches exist only in the diagnostic library since round 6)
                import _diag

                _diag.use_diag()
            generate(arm)
        else:
            # (TAMD_GEMM=x keeps the M = batch projections on the 256 x 256 MFMA tiles instead of csrc/gemv.hip)
            for arm, env in (("tamd", {}), ("tamd, TAMD_GEMM=x (tile GEMMs)", {"TAMD_GEMM": "x"}),
                             ("tamd, TAMD_DECODE_KERNEL=0", {"TAMD_DECODE_KERNEL": "0"}), ("sdpa", {})):
                e = dict(os.environ, DECODE_BENCH_ARM=arm, **env)
                subprocess.run([sys.executable, __file__, "generate"], env=e, timeout=600)
