#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): fwd+bwd tokens/s of Llama-3-8B at seq 4096 on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          # self-launching: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # (what the driver does; both forms are the same run)

A step = one forward + backward of `LlamaForCausalLM` (random-init Llama-3-8B architecture, bf16 weights,
synthetic token ids, labels = ids, loss included) through the unchanged reference model classes with
`transformers_amd.accelerate(model)` applied; per-GPU batch 8 x 4096 tokens (weak scaling: global batch 8N).
With N > 1 the model is wrapped in torch DDP (what Trainer/accelerate do, src/transformers/trainer.py:1609-1635)
and gradients are all-reduced over RCCL/xGMI, overlapped with the backward.

`--config` selects the workload; the other BASELINE.json configurations print the same JSON schema:
    llama3-8b (default)  BASELINE config 3/4, the headline metric
    bert-base            BASELINE config 2: BertForMaskedLM fwd+bwd, batch 32 x seq 512, bf16, train mode (dropout 0.1)
    llava                BASELINE config 5: LLaVA-1.5-7B-shaped forward, one 336x336 image + 512 text tokens
    llama-tiny           debugging only
`--fused-lm-head-loss` runs the Llama step with `accelerate(model, fused_lm_head_loss=True)` (SURVEY section 8 row f1:
lm_head GEMM + loss chunk by chunk, no [tokens, vocab] logits): same FLOPs, same loss, ~17 GB less memory.

`--train-step` (Llama configurations) times the whole optimizer step of `Trainer` instead of fwd+bwd alone: forward, backward,
global gradient-norm clip (max_grad_norm 1.0, training_args.py:856) and AdamW (`transformers_amd.TamdAdamW(max_grad_norm=1.0)`:
SURVEY section 8 row f2; `--optimizer torch` = torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW(fused=True) for the A/B).

The default single-GPU headline run also reports, under `secondary`, the other single-GPU configurations of BASELINE.json and
the training step -- each in its OWN process after the headline's timed region (this same script with `--config bert-base`,
`--config llava`, `--train-step`), so that nothing they do can take the headline number down; `--no-secondary` skips them.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel family (the MFMA GEMM, csrc/gemm.hip): algorithmic FLOPs of its launches
                  in the timed region / their summed HIP-event durations, against the 2.5 PFLOP/s dense bf16 peak;
  cpu_baseline -- the reference's own eager path (pip transformers, byte-identical hot-path files) timed on the
                  host cores on a bounded sample: one Llama-3-8B decoder layer, batch 1 x seq 4096, bf16.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

LLAMA3_8B = dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, max_position_embeddings=8192,
                 rope_parameters={"rope_type": "default", "rope_theta": 500000.0})
CONFIGS = {
    # SURVEY.md section 8 "L3": public Llama-3-8B config.json values
    "llama3-8b": dict(kind="llama", model=LLAMA3_8B, batch=8, seq=4096),
    # debugging only (never reported as the headline metric)
    "llama-tiny": dict(kind="llama", batch=4, seq=2048,
                       model=dict(vocab_size=4096, hidden_size=1024, intermediate_size=2816, num_hidden_layers=4,
                                  num_attention_heads=8, num_key_value_heads=2, rms_norm_eps=1e-5,
                                  max_position_embeddings=4096,
                                  rope_parameters={"rope_type": "default", "rope_theta": 500000.0})),
    # SURVEY.md section 8 "BB": BertConfig() defaults = bert-base-uncased
    "bert-base": dict(kind="bert", model={}, batch=32, seq=512),
    # SURVEY.md section 8 "LV": CLIP ViT-L/14-336 + Llama-2-7B-shaped LLM, 576 image + 512 text positions, forward only
    "llava": dict(kind="llava", batch=1, seq=1088,
                  model=dict(vision=dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                         num_attention_heads=16, image_size=336, patch_size=14),
                             text=dict(vocab_size=32064, hidden_size=4096, intermediate_size=11008,
                                       num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=32,
                                       max_position_embeddings=4096, rms_norm_eps=1e-5))),
}

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
GEMM_TRAFFIC_FILES = ("r06_gemm_traffic.json", "r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json", "r02_gemm_traffic.json", "r01_gemm_traffic.json")


def llama_flops_per_token(m, seq, backward=True) -> float:
    """Algorithmic FLOPs per token: fwd+bwd = 3 x forward; causal attention counted at half."""
    h, i, l, v = m["hidden_size"], m["intermediate_size"], m["num_hidden_layers"], m["vocab_size"]
    d = h // m["num_attention_heads"]
    nq, nkv = m["num_attention_heads"] * d, m["num_key_value_heads"] * d
    per_layer = 2 * h * (nq + 2 * nkv) + 2 * nq * h + 3 * 2 * h * i + 4 * nq * seq / 2
    return (3.0 if backward else 1.0) * (l * per_layer + 2 * h * v)


def bert_flops_per_token(seq, h=768, i=3072, layers=12, vocab=30522) -> float:
    """bert-base MLM fwd+bwd: 4 h^2 projections + 2 h*i MLP + bidirectional attention + the MLM head."""
    per_layer = 2 * (4 * h * h + 2 * h * i) + 4 * h * seq
    return 3.0 * (layers * per_layer + 2 * h * h + 2 * h * vocab)


def llava_flops(c) -> float:
    v, t = c["model"]["vision"], c["model"]["text"]
    n_img = (v["image_size"] // v["patch_size"]) ** 2 + 1
    hv, iv = v["hidden_size"], v["intermediate_size"]
    clip = n_img * v["num_hidden_layers"] * (2 * (4 * hv * hv + 2 * hv * iv) + 4 * hv * n_img)
    proj = (n_img - 1) * 2 * (hv * t["hidden_size"] + t["hidden_size"] ** 2)
    return clip + proj + c["seq"] * llama_flops_per_token(t, c["seq"], backward=False)


def gemm_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); written by
    tools/prof_traffic.py into profiles/.  None when no profile of the current kernel is committed."""
    for name in GEMM_TRAFFIC_FILES:
        try:
            with open(ROOT / "profiles" / name) as f:
                t = json.load(f)
            return {"unit": "bytes/launch", "hbm_bytes": t["hbm_bytes_per_launch"],
                    "fetch_bytes": t["fetch_bytes_per_launch"], "write_bytes": t["write_bytes_per_launch"],
                    "source": f"profiles/{name}",
                    "from": "committed rocprofv3 PMC profile of this command (an earlier run), NOT measured by this run"}
        except (OSError, KeyError, ValueError):
            continue
    return None


def _swiglu_bwd_choices():
    try:
        from transformers_amd import _native

        return _native.swiglu_bwd_choices()
    except Exception as e:  # (never take the line down)
        return f"unavailable: {e}"


def fused_ways_out(gs, steps):
    """The launches of the log that carry elementwise work of the backward in their way out (tamd_gemm_swiglu_bwd: the SiLU*up
    backward inside the down projection's dX GEMM, one per decoder layer -- 3.76 GB of gate / up / d_gate / d_up traffic and ~18
    VALU instructions per element where a plain way out stores 0.94 GB): `achieved` above counts their whole duration against the
    product's FLOPs only; `frac_of_the_other_launches` is the same figure without them."""
    n = gs.get("fused_bwd_launches", 0)
    if not n:
        return None
    rest_ms, rest_fl = gs["ms"] - gs["fused_bwd_ms"], gs["flops"] - gs["fused_bwd_flops"]
    return {"what": "SiLU*up backward in the way out of the down projection's dX GEMM (replaces swiglu_bwd_kernel, 0.87-0.94 ms "
                    "per layer, and the d_act round trip through HBM)",
            "launches_per_step": n // steps, "avg_launch_ms": gs["fused_bwd_ms"] / n,
            "achieved_on_the_product_flops": gs["fused_bwd_flops"] / (gs["fused_bwd_ms"] * 1e-3) / 1e12,
            "frac_of_the_other_launches": rest_fl / (rest_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}


class GemmTimer:
    """HIP-event timing of every MFMA-GEMM launch inside the timed region.  The launches are issued by the compiled ops
    (csrc/torch_binding.cpp), so the event pairs are recorded there -- on the launch stream, around each tamd_gemm /
    tamd_gemm_swiglu call -- and read back through transformers_amd._native."""

    def __init__(self):
        self._on = False
        self._summary = None

    @property
    def enabled(self):
        return self._on

    @enabled.setter
    def enabled(self, on):
        from transformers_amd import _native

        if on and not self._on:
            _native.gemm_log(True)
        elif not on and self._on:
            _native.gemm_log(False)
            self._summary = _native.gemm_log_summary()
        self._on = bool(on)

    def install(self):
        pass

    def summary(self):
        s = self._summary
        return s if s and s["launches"] else None


def clock_probe(dev, m=32768, n=28672, k=4096):
    """The clock the GEMM runs at and how busy its matrix pipe is: the gate|up forward GEMM of the workload under the
    diagnostic build's probe (libtamd_diag.so, include/tamd_diag.h: every workgroup stamps s_memtime and the 100 MHz
    s_memrealtime around its K loop).  MI355X lowers its clock to stay inside the power budget: 2.5 PFLOP/s is the
    2.4 GHz figure; `peak_at_clock` is the same arithmetic at the clock measured here."""
    import ctypes

    import torch

    sys.path.insert(0, str(ROOT / "tools"))
    import _diag
    from transformers_amd import ops

    prev = ops.backend()
    lib = _diag.use_diag()
    try:
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
        wgs = (m // 256) * (n // 256)
        for _ in range(4):
            ops.raw_gemm(x, w)
        buf = torch.zeros(2 * wgs, dtype=torch.int64, device=dev)
        lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(buf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.raw_gemm(x, w)
        e1.record()
        torch.cuda.synchronize()
        lib.tamd_gemm_set_clock_buffer(ctypes.c_void_p(0))
        t = buf.cpu().view(wgs, 2).double()
        ghz = (t[:, 0] / t[:, 1]).mean().item() * 0.1
        return {"kernel": f"gemm_fl_kernel {m}x{n}x{k} (gate|up forward), diagnostic build", "clock_GHz": ghz,
                "mfma_busy_in_k_loop": ((k // 64) * 2048 / t[:, 0]).mean().item(),
                "k_loop_share_of_kernel": (t[:, 1].sum() * 1e-8 / 256 / (e0.elapsed_time(e1) * 1e-3)).item(),
                "TFLOPs": 2.0 * m * n * k / (e0.elapsed_time(e1) * 1e-3) / 1e12,
                "peak_at_clock_TFLOPs": PEAK_BF16_TFLOPS * ghz / 2.4}
    finally:
        ops._set_backend(prev)


def layer_forward_probe(model, dev, batch, seq, iters=10, warm=3):
    """The north star's own figure (BASELINE.json): ONE `LlamaDecoderLayer.forward`
    (modeling_llama.py:295-324 -- RMSNorm, fused q|k|v projection, rotary embedding, causal GQA attention, o_proj + residual,
    RMSNorm, gate|up + SiLU*up, down_proj + residual) of the benchmarked model at (batch, seq, hidden) as a fraction of the
    dense bf16 MFMA peak.  Layer 0 of the model `main()` has just timed, its weights, the same op (`torch.ops.tamd.llama_layer`);
    HIP events on the launch stream around `iters` forwards, after the timed region and outside it.  `ms` is the forward
    alone (no_grad: the gate|up pre-activations are not written), `ms_training_forward` the forward of a training step
    (everything the backward needs is saved)."""
    import torch

    layer = model.model.layers[0]
    mcfg = model.config
    h, i = mcfg.hidden_size, mcfg.intermediate_size
    d = h // mcfg.num_attention_heads
    nq, nkv = mcfg.num_attention_heads * d, mcfg.num_key_value_heads * d
    flop = batch * seq * (2 * h * (nq + 2 * nkv) + 2 * nq * h + 3 * 2 * h * i + 4 * nq * seq / 2)
    x = torch.randn(batch, seq, h, device=dev).to(next(layer.parameters()).dtype)
    pos = torch.arange(seq, device=dev)[None]
    with torch.no_grad():
        pe = model.model.rotary_emb(x, pos)

    def timed(fn):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def fwd():
        with torch.no_grad():
            return layer(x, position_embeddings=pe)

    xg = x.clone().requires_grad_(True)

    def fwd_train():
        return layer(xg, position_embeddings=pe)

    ms = timed(fwd)
    ms_t = timed(fwd_train)
    return {"what": f"one LlamaDecoderLayer.forward at ({batch}, {seq}, {h}): {flop / 1e12:.3f} TFLOP algorithmic "
                    "(causal attention at half), layer 0 of the timed model, HIP events after the timed region",
            "ms": ms, "tflops": flop / ms / 1e9, "frac_of_2500": flop / ms / 1e9 / PEAK_BF16_TFLOPS,
            "ms_training_forward": ms_t, "frac_of_2500_training_forward": flop / ms_t / 1e9 / PEAK_BF16_TFLOPS,
            "iters": iters}


def cpu_baseline(model_cfg, seq, layers, threads=None, iters=1):
    """Reference eager path on the host cores: ONE decoder layer at (1, seq, hidden), bf16, fwd+bwd, averaged."""
    import torch
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    cfg = LlamaConfig(**model_cfg, attn_implementation="eager")
    torch.manual_seed(0)
    layer = LlamaDecoderLayer(cfg, 0).to(torch.bfloat16)
    rot = LlamaRotaryEmbedding(cfg)

    def make(s):
        x = torch.randn(1, s, cfg.hidden_size, dtype=torch.bfloat16, requires_grad=True)
        pe = rot(x, torch.arange(s)[None])
        mask = torch.full((s, s), torch.finfo(torch.bfloat16).min, dtype=torch.bfloat16).triu(1)[None, None]
        return x, pe, mask

    def step(x, pe, mask):
        y = layer(x, attention_mask=mask, position_embeddings=pe)
        y.backward(torch.ones_like(y))

    step(*make(min(seq, 512)))  # warm-up (allocator, thread pool) on a short sequence
    args = make(seq)
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step(*args)
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return dict(value=seq / (dt * layers), unit="tokens/s", cores=threads, kind="reference",
                sample=(f"transformers eager LlamaDecoderLayer (Llama-3-8B dims) fwd+bwd, batch 1 x seq {seq}, bf16, "
                        f"mean of {iters} iterations ({', '.join(f'{t:.1f}' for t in times)} s) = {dt:.2f} s/layer x "
                        f"{layers} layers extrapolated; embedding/lm_head/loss excluded"))


def cpu_baseline_bert(batch, seq, sample_batch=1, threads=None, iters=1):
    """Reference eager path on the host cores for BASELINE config 2: the reference's own BertForMaskedLM (bert-base-uncased
    architecture, random init, bf16, train mode, eager attention) fwd+bwd on a `sample_batch` x seq slice of the workload
    (every sequence is independent: tokens/s does not depend on the batch beyond cache effects).  The sample is ONE sequence:
    the reference's bf16 path ran 33 tokens/s on the GPU box's 256 host cores in round 3 (profiles/r03b_bench_bert_cpu.json:
    123 s per 8 x 512 step), so one sequence is the ~15 s budget."""
    import torch
    from transformers import BertConfig, BertForMaskedLM

    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    cfg = BertConfig(attn_implementation="eager")
    model = BertForMaskedLM(cfg).to(torch.bfloat16).train()
    ids = torch.randint(1000, cfg.vocab_size, (sample_batch, seq))
    labels = ids.clone()
    labels[torch.rand(sample_batch, seq) < 0.85] = -100

    def step():
        model(input_ids=ids, labels=labels).loss.backward()
        model.zero_grad(set_to_none=True)

    ids_full, labels_full = ids, labels
    ids, labels = ids_full[:, :32], labels_full[:, :32]
    step()  # warm-up (allocator, thread pool) on a short sequence
    ids, labels = ids_full, labels_full
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return dict(value=sample_batch * seq / dt, unit="tokens/s", cores=threads, kind="reference",
                sample=(f"transformers eager BertForMaskedLM (bert-base-uncased dims, all 12 layers + MLM head + loss) fwd+bwd, "
                        f"batch {sample_batch} x seq {seq} of the workload's {batch} x {seq}, bf16, train mode, mean of "
                        f"{iters} iterations ({', '.join(f'{t:.2f}' for t in times)} s)"))


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def ddp_verify(net, model, fwd, dev, world):
    """Reduced gradients of one step with the dW GEMMs writing into DDP's bucket views (transformers_amd/ddp.py) against one step
    with torch's own copies into the buckets, same batch, same weights: norm-relative error and bit identity, worst parameter,
    worst rank.  The same code on every rank (its collective is symmetric); any exception becomes `error` (and counts as "not
    verified": the caller then times the run on torch's copies).  Leaves the gradients None."""
    import torch
    import torch.distributed as dist

    from transformers_amd import ddp as tamd_ddp

    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    params = [p for p in model.parameters() if p.requires_grad]
    flag = torch.zeros(3, device=dev, dtype=torch.float64)  # max rel err, any bit differs, any rank failed
    out = {}
    try:
        def one_step():
            fwd(net).loss.backward()

        before = dict(tamd_ddp.STATS)
        model.zero_grad(set_to_none=True)
        one_step()
        sync()
        zc_layers = tamd_ddp.STATS["zero_copy_layers"] - before["zero_copy_layers"]
        # (a low-memory copy: the gradients are 16 GB for the 8B model, kept once)
        kept = [p.grad.detach().clone() for p in params]
        model.zero_grad(set_to_none=True)
        was = tamd_ddp.set_enabled(False)
        try:
            one_step()
            sync()
        finally:
            tamd_ddp.set_enabled(was)
        for p, g0 in zip(params, kept):
            g1 = p.grad.detach()
            den = g1.double().norm()
            err = (g0.double() - g1.double()).norm() / den.clamp_min(1e-30)
            flag[0] = torch.maximum(flag[0], torch.nan_to_num(err, nan=float("inf")))
            flag[1] = torch.maximum(flag[1], (g0 != g1).any().double())
        del kept
        out.update(parameters=len(params), zero_copy_layers_in_checked_step=zc_layers, ranks=world,
                   what="reduced gradients of one step with the dW GEMMs writing into the bucket views vs one step with "
                        "torch's copies into the buckets, same batch; worst parameter, worst rank")
    except Exception as e:
        flag[2] = 1.0
        out["error"] = repr(e)
    try:
        model.zero_grad(set_to_none=True)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        ok = flag[2].item() == 0.0
        # verified: no rank failed and the two hand-overs agree to reduction-order noise (a wrong view or a missed contribution
        # is an O(1) error; `bit_identical` is reported beside it -- it holds on gloo and on RCCL at world size 1)
        out.update(max_rel_err=flag[0].item(), bit_identical=bool(ok and flag[1].item() == 0.0),
                   verified=bool(ok and flag[0].item() <= 1e-3))
        if not ok and "error" not in out:
            out["error"] = "the check failed on another rank"
    except Exception as e:  # (the collective itself failed: nothing can be said -- not verified)
        out.update(bit_identical=False, verified=False, error=out.get("error", repr(e)))
    return out


def ddp_verify_or_fall_back(args, net, model, fwd, dev, world):
    """`ddp_verify`, and what follows from it: a hand-over that did not verify (on any rank: the verdict is all-reduced) is
    switched off on every rank -- `args.no_ddp_zero_copy` follows, so the steps zero the gradients in place -- and the caller
    times the run on torch's copies.  The first multi-rank run must not report a throughput measured on wrong gradients."""
    from transformers_amd import ddp as tamd_ddp

    v = ddp_verify(net, model, fwd, dev, world)
    if not v.get("verified", False):
        tamd_ddp.set_enabled(False)
        args.no_ddp_zero_copy = True
        v["disabled_zero_copy"] = True
    return v


def ddp_report(args, net, model, fwd, dev, world, ddp_ms):
    """What an N-GPU line says about the data-parallel step beyond its throughput.  Runs AFTER the timed region,
    the same code on every rank (its two collectives are symmetric); every leg is fenced so that a failure becomes an `error`
    entry instead of taking the measured number down.
      gradient_bytes / buckets_per_step -- what DDP all-reduces per step, in how many buckets;
      compute_only_ms       -- `--steps`-independent: 3 untimed steps under `net.no_sync()` (same model, same batch, no all-reduce);
      exposed_allreduce_ms  -- ms_per_step - compute_only_ms (max over ranks): the part of the collective the backward did not hide;
      busbw_GBps            -- 2 (N-1)/N x gradient_bytes / exposed_allreduce_ms: the ring all-reduce's bus bandwidth IF the exposed
                               time were the whole collective (a lower bound of the real one: most of it overlaps).
    (The zero-copy hand-over is verified BEFORE the timed region: `ddp_verify()`, `ddp_verify` in the line.)"""
    import torch
    import torch.distributed as dist

    from transformers_amd import ddp as tamd_ddp

    out = {}
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)  # (the world-2 gloo test runs this on the CPU model)
    params = [p for p in model.parameters() if p.requires_grad]
    gbytes = sum(p.numel() * p.element_size() for p in params)
    out["gradient_bytes"] = gbytes
    out["buckets_per_step"] = None
    if tamd_ddp.STATS.get("buckets_reduced"):
        # (every step that reduced buckets so far: warm-up, the timed ones, the two of the verification and the two warm steps
        # of its fall-back)
        extra = getattr(args, "_untimed_ddp_steps", 0)
        out["buckets_per_step"] = tamd_ddp.STATS["buckets_reduced"] / max(args.steps + args.warmup + extra, 1)

    def one_step(sync=True):
        ctx = net.no_sync() if not sync else torch.enable_grad()
        with ctx:
            o = fwd(net)
            o.loss.backward()

    if not args.no_ddp_breakdown:
        try:
            model.zero_grad(set_to_none=True)
            n = 3
            one_step(sync=False)  # (untimed: the first no_sync step re-points nothing, but warms the path)
            model.zero_grad(set_to_none=True)
            sync()
            t0 = time.perf_counter()
            for _ in range(n):
                one_step(sync=False)
                model.zero_grad(set_to_none=True)
            sync()
            cms = (time.perf_counter() - t0) / n * 1e3
            t = torch.tensor([cms], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cms = t.item()
            exposed = max(ddp_ms - cms, 0.0)
            out.update(compute_only_ms=cms, exposed_allreduce_ms=exposed,
                       busbw_GBps=(2.0 * (world - 1) / world * gbytes / (exposed * 1e-3) / 1e9) if (world > 1 and exposed > 0) else None)
        except Exception as e:
            out["breakdown_error"] = repr(e)
    return out


SECONDARY_LEGS = {"bert_base": ["--config", "bert-base", "--steps", "20", "--warmup", "5"],
                  "llava": ["--config", "llava", "--steps", "20", "--warmup", "5"],
                  "train_step": ["--config", "llama3-8b", "--train-step", "--steps", "3", "--warmup", "2"]}


def secondary_legs(timeout_s=300, legs=None):
    """BASELINE.json configs 2 and 5 and one end-to-end training step, each as a child process running this script (the
    parent has released its GPU memory): a crash, a hang (timeout) or an out-of-memory in a leg becomes an `error` entry."""
    legs = SECONDARY_LEGS if legs is None else legs
    out = {}
    for name, extra in legs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--gpus", "1", "--no-cpu-baseline",
                                "--no-secondary", *extra], capture_output=True, text=True, timeout=timeout_s)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc {r.returncode}: {(r.stderr or r.stdout)[-400:]}")
            d = json.loads(lines[-1])
            rf = d.get("roofline") or {}
            leg = {"workload": d["config"]["workload"], "metric": d["metric"], "ms_per_step": d["ms_per_step"],
                   "tokens_per_s": d["value"], "mfu_vs_2500TF": d["mfu_vs_2500TF"], "steps": d["steps"],
                   "warmup": d["warmup"], "fallback_calls": d["fallback_calls"], "loss": d["loss"],
                   "max_memory_gb": d["max_memory_gb"],
                   "roofline": {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "launches_per_step",
                                                        "gemm_share_of_step_time", "measured_on")} if rf else None}
            if "train_step" in d:
                leg.update(d["train_step"])
            out[name] = leg
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timeout after {timeout_s} s"}
        except Exception as e:
            out[name] = {"error": repr(e)}
        out[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
    return out


# ------------------------------------------------------------------------------------------------ workloads
def build_llama(c, dev, args):
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM

    import transformers_amd

    cfg = LlamaConfig(**c["model"], attn_implementation="eager")
    torch.manual_seed(0)  # identical weights on every rank
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        model = LlamaForCausalLM(cfg)
    torch.set_default_dtype(old)
    model.train()
    transformers_amd.accelerate(model, fused_lm_head_loss=args.fused_lm_head_loss)
    return model, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="llama3-8b", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture one step (forward, or forward + backward with graph-safe dropout seeds) in a HIP graph after "
                         "the warm-up and time replays -- for configurations whose host side (hundreds to thousands of "
                         "10-50 us launches) bounds the step")
    ap.add_argument("--fused-lm-head-loss", action="store_true")
    ap.add_argument("--gemm-timer", choices=("auto", "on", "off"), default="auto",
                    help="HIP-event pair around every GEMM launch of the timed region (the `roofline` object).  auto = on for "
                         "the Llama configurations, off for bert-base / llava: their steps are hundreds to thousands of "
                         "10-50 us launches and two event records per GEMM add host time to a host-bound step")
    ap.add_argument("--force-ddp", action="store_true",
                    help="wrap in DDP over RCCL even with one rank (exercises init / bucket all-reduce / destroy)")
    ap.add_argument("--ddp-grads", choices=("none", "zero", "keep"), default=None,
                    help="between steps: none = zero_grad(set_to_none=True) (Trainer's default; ours without DDP), zero = "
                         "zero in place (gradients stay views of the DDP buckets; ours under DDP), keep = no zero_grad")
    ap.add_argument("--bucket-mb", type=int, default=int(os.environ.get("TAMD_DDP_BUCKET_MB", "256")))
    ap.add_argument("--no-ddp-zero-copy", action="store_true",
                    help="under DDP: leave the gradient hand-over to torch (copy into the bucket views) instead of writing the "
                         "weight-gradient GEMMs straight into them (transformers_amd/ddp.py)")
    ap.add_argument("--verify-ddp", action="store_true",
                    help="under DDP: two extra untimed steps on one fixed batch BEFORE the timed region -- the reduced gradients "
                         "of a step with the zero-copy hand-over against a step with torch's own copies into the buckets, "
                         "norm-relative per parameter, worst over parameters and ranks -> `ddp_verify` in the line.  Always on "
                         "with more than one rank; not bit-identical = the run is timed on torch's copies and says so")
    ap.add_argument("--train-step", action="store_true",
                    help="Llama configurations: time forward + backward + gradient-norm clip (1.0) + AdamW instead of fwd+bwd")
    ap.add_argument("--optimizer", choices=("tamd", "torch"), default="tamd",
                    help="--train-step: tamd = TamdAdamW(max_grad_norm=1.0) (one norm pass, clip folded into one multi-tensor "
                         "AdamW launch); torch = torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW(fused=True)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="single-GPU llama3-8b headline run: skip the `secondary` legs (bert-base, llava, train step)")
    ap.add_argument("--no-ddp-breakdown", action="store_true",
                    help="under DDP: skip the untimed no_sync() steps after the timed region that price the exposed all-reduce "
                         "(`ddp.exposed_allreduce_ms`, `ddp.busbw_GBps`)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ddp = world > 1 or args.force_ddp
    if ddp:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not launched:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" on ROCm is RCCL

    import transformers_amd

    c = CONFIGS[args.config]
    batch, seq, kind = c["batch"], c["seq"], c["kind"]
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)  # different synthetic data per rank
    backward = True
    if kind == "llama":
        model, cfg = build_llama(c, dev, args)
        ids = torch.randint(0, cfg.vocab_size, (batch, seq), device=dev, generator=g)
        fwd = lambda net: net(input_ids=ids, labels=ids, use_cache=False)  # noqa: E731
        flops_step = batch * seq * llama_flops_per_token(c["model"], seq)
        workload = (f"{args.config}: LlamaForCausalLM fwd+bwd incl. lm_head + causal-LM loss"
                    f"{' (fused lm_head+loss, no logits tensor)' if args.fused_lm_head_loss else ''}, "
                    f"random-init bf16 weights, per-GPU batch {batch} x seq {seq}")
        metric = f"fwd+bwd tokens/sec (whole job), {'Llama-3-8B' if args.config == 'llama3-8b' else args.config} seq={seq}"
    elif kind == "bert":
        from transformers import BertConfig, BertForMaskedLM

        cfg = BertConfig(**c["model"], attn_implementation="eager")
        torch.manual_seed(0)
        model = BertForMaskedLM(cfg).to(torch.bfloat16).to(dev).train()
        transformers_amd.accelerate(model)
        ids = torch.randint(1000, cfg.vocab_size, (batch, seq), device=dev, generator=g)
        labels = ids.clone()
        labels[torch.rand(batch, seq, device=dev, generator=g) < 0.85] = -100  # MLM: 15 % of the positions are scored
        fwd = lambda net: net(input_ids=ids, labels=labels)  # noqa: E731
        flops_step = batch * seq * bert_flops_per_token(seq)
        workload = (f"bert-base-uncased BertForMaskedLM fwd+bwd, random-init bf16, batch {batch} x seq {seq}, train mode "
                    "(hidden/attention dropout 0.1 as shipped), attention_mask=None")
        metric = "fwd+bwd tokens/sec (whole job), bert-base-uncased seq=512"
    else:  # llava: forward only
        from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration

        backward = False
        vcfg = CLIPVisionConfig(**c["model"]["vision"])
        tcfg = LlamaConfig(**c["model"]["text"])
        cfg = LlavaConfig(vision_config=vcfg, text_config=tcfg, image_token_id=32000, vision_feature_layer=-2,
                          vision_feature_select_strategy="default", attn_implementation="eager")
        torch.manual_seed(0)
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch.bfloat16)
        with torch.device(dev):
            model = LlavaForConditionalGeneration(cfg)
        torch.set_default_dtype(old)
        model.eval()
        transformers_amd.accelerate(model)
        n_img = (vcfg.image_size // vcfg.patch_size) ** 2
        txt = torch.randint(0, 31000, (1, seq - n_img), device=dev, generator=g)
        ids = torch.cat([txt[:, :5], torch.full((1, n_img), 32000, device=dev), txt[:, 5:]], dim=1)
        px = torch.randn(1, 3, vcfg.image_size, vcfg.image_size, device=dev, generator=g).bfloat16()
        fwd = lambda net: net(input_ids=ids, pixel_values=px, use_cache=False)  # noqa: E731
        flops_step = llava_flops(c)
        workload = (f"LLaVA-1.5-7B-shaped LlavaForConditionalGeneration forward (no grad): CLIP ViT-L/14-336 tower "
                    f"(hidden_states[-2]) + projector + Llama-2-7B-shaped LLM, one 336x336 image ({n_img} positions) + "
                    f"{seq - n_img} text tokens, random-init bf16")
        metric = "forward tokens/sec (whole job), LLaVA-1.5-7B 336px image + 512 text tokens"

    timer = GemmTimer()
    use_timer = args.gemm_timer == "on" or (args.gemm_timer == "auto" and kind == "llama")
    if use_timer:
        timer.install()
    net = model
    zero_copy_error = None
    if ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP

        net = DDP(model, device_ids=[local_rank], bucket_cap_mb=args.bucket_mb, gradient_as_bucket_view=True,
                  broadcast_buffers=False, find_unused_parameters=False, static_graph=True)
        if not args.no_ddp_zero_copy:
            # the layer ops write dW straight into DDP's bucket views and hand DDP aliases of them; the bucket is averaged
            # in place by RCCL (ReduceOp.AVG): no copy into the buckets, no scaling pass (transformers_amd/ddp.py)
            from transformers_amd import ddp as tamd_ddp

            tamd_ddp.reset()
            try:
                tamd_ddp.enable_zero_copy(net)
            except Exception as e:  # (the measurement goes on with torch's copies; the line says so)
                args.no_ddp_zero_copy = True
                zero_copy_error = repr(e)

    opt = None
    opt_events = []
    if args.train_step:
        if kind != "llama" or args.hip_graph:
            raise SystemExit("--train-step: Llama configurations, no --hip-graph")
        if args.optimizer == "tamd":
            opt = transformers_amd.TamdAdamW(model.parameters(), lr=1e-5, weight_decay=0.01, max_grad_norm=1.0)
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.01, fused=True)
        metric = metric.replace("fwd+bwd tokens/sec", "training-step (fwd+bwd+clip+AdamW) tokens/sec")
        workload = workload.replace("fwd+bwd incl.", "fwd+bwd + grad-norm clip 1.0 + AdamW (bf16 moments) incl.")

    def optimizer_step():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if args.optimizer == "torch":
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        e1.record()
        opt_events.append((e0, e1))

    def step():
        if not backward:
            with torch.no_grad():
                return fwd(net).logits[0, -1, 0].float()
        out = fwd(net)
        out.loss.backward()
        if opt is not None:
            optimizer_step()
        # Between steps.  Without DDP: what Trainer does (optimizer.zero_grad(), set_to_none=True).  Under DDP with
        # gradient_as_bucket_view the gradients are views of the all-reduce buckets: dropping them makes the next backward
        # hand DDP fresh tensors that it copies into the buckets and re-points, so they are zeroed in place instead --
        # measured on MI355X at world size 1 (profiles/r03c_ddp_grads_ab.jsonl): 1299.8 ms per step dropped, 1288.9 zeroed
        # in place, 1285.8 never zeroed (non-DDP step on the same box: 1273.4).  `--ddp-grads none|zero|keep` overrides.
        # (with the zero-copy hand-over the gradients must be None when the backward starts: the Trainer's default)
        mode = args.ddp_grads or ("zero" if (ddp and args.no_ddp_zero_copy) else "none")
        if mode == "none":
            model.zero_grad(set_to_none=True)
        elif mode == "zero":
            model.zero_grad(set_to_none=False)
        return out.loss.detach()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    ddp_preverify = None
    if ddp and backward and not args.no_ddp_zero_copy and (world > 1 or args.verify_ddp):
        # BEFORE anything is timed: the reduced gradients of one step with the zero-copy hand-over against one step with torch's
        # own copies into the buckets, same batch.  Not bit-identical (or the check itself fails): the hand-over is switched off
        # on every rank -- the verdict is all-reduced -- and the run is timed on torch's copies; the line says so.
        ddp_preverify = ddp_verify_or_fall_back(args, net, model, fwd, dev, world)
        args._untimed_ddp_steps = 2
        if ddp_preverify.get("disabled_zero_copy"):
            for _ in range(2):  # (warm the ordinary hand-over: its first step re-points the gradients at the buckets)
                loss = step()
            args._untimed_ddp_steps = 4
        barrier()
    run = step
    if args.hip_graph:
        # one step as ONE HIP graph.  Forward-only configurations: the host side is what bounds them.  Training steps
        # (round 4): dropout seeds are device words drawn by torch's graph-safe RNG kernel while capturing
        # (transformers_amd.ops.dropout_seeds), so every replay draws fresh masks; gradients are left to autograd's first-write
        # path (zero_grad(set_to_none=True) inside the captured step), i.e. a replay overwrites them.
        if world > 1:
            raise SystemExit("--hip-graph: single-GPU runs (DDP's bucket hooks are host code)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # (capture wants the allocator warmed on a side stream)
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()

        def run():
            graph.replay()
            return static_loss

        barrier()
    transformers_amd.fallback_calls(reset=True)
    timer.enabled = use_timer and rank == 0 and not args.hip_graph  # (per-GEMM events cannot be read out of a replayed graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    roofline_steps = args.steps
    if not use_timer and rank == 0 and not args.hip_graph and args.gemm_timer != "off":
        # bert-base / llava: hundreds to thousands of 10-100 us launches per step, where two event records per GEMM would
        # add host time to the timed region -- so the `roofline` object of these configurations comes from two extra
        # UNTIMED steps with the log on, run after the timed region (same kernels, same shapes)
        roofline_steps = 2
        from transformers_amd import graph_stack

        was = graph_stack.set_enabled(False)  # (launches replayed from a captured decoder stack cannot be timed one by one)
        timer.enabled = True
        for _ in range(roofline_steps):
            run()
        torch.cuda.synchronize()
        timer.enabled = False
        graph_stack.set_enabled(was)
    dt_min = dt_max = dt
    if world > 1:
        t = torch.tensor([dt, -dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dt_min = t[0].item(), -t[1].item()
        dt_max = dt
    ddp_extra = ddp_report(args, net, model, fwd, dev, world, dt / args.steps * 1e3) if (ddp and backward) else None
    tokens = batch * seq * world * args.steps
    value = tokens / dt
    if rank == 0:
        gs = timer.summary()
        roofline = None
        if gs:
            ach = gs["flops"] / (gs["ms"] * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel="tamd::gemm_fl_kernel / gemm_sm_kernel (csrc/gemm.hip; forward, dX and dW "
                                                 "layouts, all epilogues, split-K reductions included; avg over the launches "
                                                 "of a step)",
                            achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS,
                            traffic=gemm_traffic() if args.config == "llama3-8b" else None,
                            launches_per_step=gs["launches"] // roofline_steps,
                            avg_launch_ms=gs["ms"] / gs["launches"],
                            avg_launch_tflop=gs["flops"] / gs["launches"] / 1e12,
                            avg_launch_algorithmic_bytes=gs["bytes"] / gs["launches"],
                            gemm_share_of_step_time=(gs["ms"] / roofline_steps) / (dt / args.steps * 1e3),
                            fused_ways_out=fused_ways_out(gs, roofline_steps),
                            measured_on=("the timed steps" if use_timer else
                                         f"{roofline_steps} extra untimed steps after the timed region (event records off "
                                         "inside it)"))
        line = {
            "metric": metric,
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "model": args.config, "global_batch": batch * world, "seq_len": seq,
                       "parallelism": f"dp{world}" + (" (DDP over RCCL)" if ddp else "")
                       + (" (HIP graph replay)" if args.hip_graph else "")},
            "tokens_per_sec_per_gpu": value / world,
            "model_tflops_per_gpu": flops_step * args.steps / dt / 1e12,
            "mfu_vs_2500TF": flops_step * args.steps / dt / (PEAK_BF16_TFLOPS * 1e12),
            "loss": float(loss),
            "max_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
            # GPU tensors served by a reference module's own forward (ATen / vendor kernels) inside the timed region
            "fallback_calls": sum(transformers_amd.fallback_calls().values()),
            "fallbacks": transformers_amd.fallback_calls(),
            # the form of the SiLU*up backward the library MEASURED to be faster on this box, per shape (the dX GEMM's way out or
            # GEMM + kernel: bit-identical; csrc/torch_binding.cpp swiglu_bwd_fused)
            "swiglu_bwd": _swiglu_bwd_choices(),
            # forward-only decoder stacks replayed as one HIP graph (transformers_amd/graph_stack.py), whole run
            "stack_graph_replays": sum(m.__dict__["_tamd_stack"][0].replays for m in model.modules()
                                       if m.__dict__.get("_tamd_stack", (None, 1))[1] == 0),
            "roofline": roofline,
        }
        if ddp:
            from transformers_amd import ddp as tamd_ddp

            line["ddp_zero_copy"] = dict(tamd_ddp.STATS, enabled=not args.no_ddp_zero_copy)
            if zero_copy_error:
                line["ddp_zero_copy"]["error"] = zero_copy_error
            line["ddp"] = dict(ms_per_step_min_rank=dt_min / args.steps * 1e3, ms_per_step_max_rank=dt_max / args.steps * 1e3,
                               bucket_mb=args.bucket_mb, **(ddp_extra or {}))
            if ddp_preverify is not None:
                line["ddp_verify"] = ddp_preverify
                line["ddp_zero_copy"]["disabled_by_verify"] = bool(ddp_preverify.get("disabled_zero_copy", False))
        if opt is not None:
            torch.cuda.synchronize()
            oms = [a.elapsed_time(b) for a, b in opt_events[-args.steps:]]
            line["train_step"] = dict(optimizer=("TamdAdamW(max_grad_norm=1.0): mt_sumsq + mt_norm_finish + one mt_adamw launch per "
                                                 "dtype, clip coefficient in device memory" if args.optimizer == "tamd" else
                                                 "torch.nn.utils.clip_grad_norm_(1.0) + torch.optim.AdamW(fused=True)"),
                                      optimizer_ms=sum(oms) / max(len(oms), 1),
                                      optimizer_hbm_bytes=(1 + 7) * sum(p.numel() * p.element_size() for p in model.parameters()),
                                      grad_norm=(float(opt.grad_norm) if getattr(opt, "grad_norm", None) is not None else None))
            line["train_step"]["optimizer_TBps"] = line["train_step"]["optimizer_hbm_bytes"] / (line["train_step"]["optimizer_ms"] * 1e-3) / 1e12
        if kind == "llama" and world == 1 and not args.hip_graph and not args.train_step:
            # the north star's own number: one decoder layer's forward as a fraction of the MFMA peak (outside the timed region)
            try:
                fb = dict(transformers_amd.fallback_calls())
                line["layer_forward"] = layer_forward_probe(model, dev, batch, seq)
                line["layer_forward"]["fallback_calls"] = sum(transformers_amd.fallback_calls().values()) - sum(fb.values())
            except Exception as e:  # diagnostics must never take the headline number down
                line["layer_forward"] = {"error": repr(e)}
        if roofline is not None and world == 1 and args.config == "llama3-8b":
            try:
                roofline["clock_probe"] = clock_probe(dev)
                if "ms" in line.get("layer_forward", {}):
                    line["layer_forward"]["clock_GHz"] = roofline["clock_probe"]["clock_GHz"]
            except Exception as e:  # diagnostics only
                roofline["clock_probe"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and args.config in ("llama3-8b", "bert-base"):
            try:
                line["cpu_baseline"] = (cpu_baseline(c["model"], seq, c["model"]["num_hidden_layers"])
                                        if kind == "llama" else cpu_baseline_bert(batch, seq))
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                line["cpu_baseline"] = {"error": repr(e)}
        if (world == 1 and args.config == "llama3-8b" and not args.no_secondary and not args.train_step and not args.hip_graph
                and not ddp):
            # BASELINE configs 2 and 5 + the end-to-end training step, each in its own process (this one frees its memory first)
            try:
                import gc

                del net, model, fwd, ids, loss
                run = step = None
                gc.collect()
                torch.cuda.empty_cache()
                line["secondary"] = secondary_legs()
            except Exception as e:
                line["secondary"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if ddp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
