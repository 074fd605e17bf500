"""Per-kernel micro-benchmarks on one MI355X: TFLOP/s for the MFMA kernels (next to torch.mm = hipBLASLt),
GB/s for the HBM-bound kernels.  Writes JSON lines to stdout (redirect into gpurun_out/)."""
import json
import math
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from transformers_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_gemm():
    T = 32768
    # the first shape measured used to read 10-15 % low (q|k|v 1257-1303 TFLOP/s where the same call runs 1490 once the part is warm:
    # profiles/r06s_gemm_ld_ab.jsonl): half a second of GEMM before anything is timed
    xw, ww = torch.randn(8192, 8192, device=dev).bfloat16(), torch.randn(8192, 8192, device=dev).bfloat16()
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(20):
            ops.raw_gemm(xw, ww)
        torch.cuda.synchronize()
    del xw, ww
    shapes = [("qkv", T, 6144, 4096), ("o_proj", T, 4096, 4096), ("gate_up", T, 28672, 4096),
              ("down", T, 4096, 14336), ("lm_head", T, 128256, 4096), ("sq8k", 8192, 8192, 8192),
              ("bert_qkv", 16384, 2304, 768), ("bert_ffn1", 16384, 3072, 768)]
    for name, m, n, k in shapes:
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
        fl = 2.0 * m * n * k
        t = timeit(lambda: ops.raw_gemm(x, w))
        tt = timeit(lambda: torch.mm(x, w.t()))
        emit(kernel="gemm_nt", shape=name, m=m, n=n, k=k, ms=t * 1e3, tflops=fl / t / 1e12,
             torch_mm_ms=tt * 1e3, torch_mm_tflops=fl / tt / 1e12)
        if name in ("qkv", "gate_up", "down"):
            dy = torch.randn(m, n, device=dev).bfloat16()
            t = timeit(lambda: ops.raw_gemm(dy, w, b_kn=True))          # dX [m,k]
            tt = timeit(lambda: torch.mm(dy, w))
            emit(kernel="gemm_dx(b_kn)", shape=name, ms=t * 1e3, tflops=fl / t / 1e12, torch_mm_tflops=fl / tt / 1e12)
            t = timeit(lambda: ops.raw_gemm(dy, x, a_km=True, b_kn=True))  # dW [n,k]
            tt = timeit(lambda: torch.mm(dy.t(), x))
            emit(kernel="gemm_dw(a_km|b_kn)", shape=name, ms=t * 1e3, tflops=fl / t / 1e12,
                 torch_mm_tflops=fl / tt / 1e12)
            del dy
        del x, w


def bench_attn():
    for name, b, s, hq, hkv, d, causal in [("llama3-8b", 8, 4096, 32, 8, 128, True), ("bert-base", 32, 512, 12, 12, 64, False)]:
        q = torch.randn(b, s, hq, d, device=dev).bfloat16()
        k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
        v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
        fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
        t = timeit(lambda: ops.raw_attn_fwd(q, k, v, d ** -0.5, causal))
        o, lse = ops.raw_attn_fwd(q, k, v, d ** -0.5, causal)
        do = torch.randn_like(o)
        tb = timeit(lambda: ops.raw_attn_bwd(q, k, v, o, lse, do, d ** -0.5, causal))
        qt, kt, vt = (x.transpose(1, 2) for x in (q, k, v))
        ts = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qt, kt, vt, is_causal=causal, enable_gqa=hq != hkv))
        emit(kernel="attn_fwd", shape=name, ms=t * 1e3, tflops=fl / t / 1e12, sdpa_ms=ts * 1e3, sdpa_tflops=fl / ts / 1e12)
        emit(kernel="attn_bwd", shape=name, ms=tb * 1e3, tflops=2.0 * fl / tb / 1e12, note="algorithmic 2x fwd flops")


def bench_hbm():
    T, H, I = 32768, 4096, 14336
    x = torch.randn(T, H, device=dev).bfloat16()
    w = torch.ones(H, device=dev).bfloat16()
    t = timeit(lambda: ops.raw_rmsnorm_fwd(x, w, 1e-5))
    emit(kernel="rmsnorm_fwd", ms=t * 1e3, gbps=2 * x.numel() * 2 / t / 1e9)
    y, h, rstd = ops.raw_rmsnorm_fwd(x, w, 1e-5)
    t = timeit(lambda: ops.raw_rmsnorm_bwd(y, x, w, rstd, dres=x))
    emit(kernel="rmsnorm_bwd(+dres)", ms=t * 1e3, gbps=4 * x.numel() * 2 / t / 1e9)
    t = timeit(lambda: ops.raw_add(x, y))
    emit(kernel="add", ms=t * 1e3, gbps=3 * x.numel() * 2 / t / 1e9)
    qkv = torch.randn(T, 6144, device=dev).bfloat16()
    cos = torch.randn(1, 4096, 128, device=dev).bfloat16()
    t = timeit(lambda: ops.raw_rope_(qkv, cos, cos, 4096, 40, 128))
    emit(kernel="rope_inplace", ms=t * 1e3, gbps=2 * T * 5120 * 2 / t / 1e9)
    gu = torch.randn(T, 2 * I, device=dev).bfloat16()
    t = timeit(lambda: ops.raw_swiglu_fwd(gu))
    emit(kernel="swiglu_fwd", ms=t * 1e3, gbps=3 * T * I * 2 / t / 1e9)
    dact = torch.randn(T, I, device=dev).bfloat16()
    t = timeit(lambda: ops.raw_swiglu_bwd(gu, dact, want_act=True))
    emit(kernel="swiglu_bwd(+act)", ms=t * 1e3, gbps=6 * T * I * 2 / t / 1e9)
    del gu, dact
    table = torch.randn(128256, H, device=dev).bfloat16()
    ids = torch.randint(0, 128256, (8, 4096), device=dev)
    t = timeit(lambda: ops.raw_embedding_fwd(ids, table))
    emit(kernel="embedding_fwd", ms=t * 1e3, gbps=2 * T * H * 2 / t / 1e9)
    t = timeit(lambda: ops.raw_embedding_bwd(ids, x.view(8, 4096, H), 128256), iters=5)
    emit(kernel="embedding_bwd(+zero+sort)", ms=t * 1e3)
    del table
    t = timeit(lambda: ops.raw_transpose(x))
    emit(kernel="transpose", ms=t * 1e3, gbps=2 * x.numel() * 2 / t / 1e9)
    logits = torch.randn(8192, 128256, device=dev).bfloat16()
    labels = torch.randint(0, 128256, (8192,), device=dev)
    t = timeit(lambda: ops.raw_cross_entropy_fwd(logits, labels), iters=5)
    emit(kernel="cross_entropy_fwd", ms=t * 1e3, gbps=logits.numel() * 2 / t / 1e9)
    lse, _ = ops.raw_cross_entropy_fwd(logits, labels)
    gs = torch.ones(1, device=dev)
    t = timeit(lambda: ops.raw_cross_entropy_bwd(logits, labels, lse, gs), iters=5)
    emit(kernel="cross_entropy_bwd", ms=t * 1e3, gbps=2 * logits.numel() * 2 / t / 1e9)
    n = 1 << 29  # one 1 GiB bf16 tensor (the 8B model is 16 of these)
    pw = torch.randn(n, device=dev).bfloat16()
    pg = torch.randn(n, device=dev).bfloat16()
    pm = torch.zeros(n, device=dev).bfloat16()
    pv = torch.zeros(n, device=dev).bfloat16()
    kw = dict(lr=1e-4, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.1, step=3)
    t = timeit(lambda: ops.raw_adamw_step_(pw, pg, pm, pv, **kw), iters=5)
    emit(kernel="adamw_step(bf16 states)", ms=t * 1e3, gbps=7 * n * 2 / t / 1e9)
    ta = torch.nn.Parameter(pw.clone())
    ta.grad = pg
    topt = torch.optim.AdamW([ta], lr=1e-4, betas=(0.9, 0.95), weight_decay=0.1, fused=True)
    t = timeit(lambda: topt.step(), iters=5)
    emit(kernel="torch AdamW(fused=True), same tensor", ms=t * 1e3, gbps=7 * n * 2 / t / 1e9)
    del pw, pg, pm, pv, ta, topt
    t = timeit(lambda: x.clone())
    emit(kernel="torch_clone(ref copy)", ms=t * 1e3, gbps=2 * x.numel() * 2 / t / 1e9)


def bench_layer():
    import transformers_amd
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    cfg = LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=1,
                      num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, max_position_embeddings=8192,
                      rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, attn_implementation="eager")
    torch.manual_seed(0)
    with torch.device(dev):
        layer = LlamaDecoderLayer(cfg, 0).bfloat16()
        rot = LlamaRotaryEmbedding(cfg)
    for p in layer.parameters():
        torch.nn.init.normal_(p, std=0.02) if p.dim() > 1 else None
    B, S = 8, 4096
    x = torch.randn(B, S, 4096, device=dev).bfloat16().requires_grad_(True)
    pos = torch.arange(S, device=dev)[None]
    pe = rot(x, pos)
    transformers_amd.attention.register()
    cfg._attn_implementation = "tamd"
    from transformers_amd.patch import _tables

    for m in layer.modules():
        r = _tables().get(type(m))
        if r is not None:
            m.__class__ = r
    fl = 15.393e12

    def fwd():
        with torch.no_grad():
            return layer(x, position_embeddings=pe)

    t = timeit(fwd, iters=5, warm=2)
    emit(kernel="llama3_8b_layer_fwd(B8,S4096)", ms=t * 1e3, tflops=fl / t / 1e12, frac_of_2500=fl / t / 2.5e15)

    def fwdbwd():
        y = layer(x, position_embeddings=pe)
        y.backward(x.detach())
        layer.zero_grad(set_to_none=True)
        x.grad = None

    t = timeit(fwdbwd, iters=5, warm=2)
    emit(kernel="llama3_8b_layer_fwd+bwd", ms=t * 1e3, tflops=3 * fl / t / 1e12, frac_of_2500=3 * fl / t / 2.5e15)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "hbm", "layer"]
    for w in which:
        try:
            globals()["bench_" + w]()
        except Exception as e:  # keep going: one failing kernel must not hide the others
            emit(kernel=w, error=repr(e))
        torch.cuda.empty_cache()
