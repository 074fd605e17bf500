#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): fwd+bwd tokens/s of Llama-3-8B at seq 4096 on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward + backward of `LlamaForCausalLM` (random-init Llama-3-8B architecture, bf16 weights,
synthetic token ids, labels = ids, loss included) through the unchanged reference model classes with
`transformers_amd.accelerate(model)` applied; per-GPU batch 8 x 4096 tokens (weak scaling: global batch 8N).
With N > 1 the model is wrapped in torch DDP (what Trainer/accelerate do, src/transformers/trainer.py:1609-1635)
and gradients are all-reduced over RCCL/xGMI, overlapped with the backward.

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
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

CONFIGS = {
    # SURVEY.md §8 "L3": public Llama-3-8B config.json values
    "llama3-8b": dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                      num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, max_position_embeddings=8192,
                      rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, batch=8, seq=4096),
    # debugging only (never reported as the headline metric)
    "llama-tiny": dict(vocab_size=4096, hidden_size=1024, intermediate_size=2816, num_hidden_layers=4,
                       num_attention_heads=8, num_key_value_heads=2, rms_norm_eps=1e-5, max_position_embeddings=4096,
                       rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, batch=4, seq=2048),
}

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def flops_per_token(c) -> float:
    """Algorithmic fwd+bwd FLOPs per token: 3 x forward; causal attention counted at half."""
    h, i, l, v = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"], c["vocab_size"]
    d = h // c["num_attention_heads"]
    nq, nkv, s = c["num_attention_heads"] * d, c["num_key_value_heads"] * d, c["seq"]
    per_layer = 2 * h * (nq + 2 * nkv) + 2 * nq * h + 3 * 2 * h * i + 4 * nq * s / 2
    return 3.0 * (l * per_layer + 2 * h * v)


def gemm_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); written by
    tools/prof_traffic.py into profiles/.  None when no profile of the current kernel is committed."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_gemm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return {"unit": "bytes/launch", "hbm_bytes": t["hbm_bytes_per_launch"],
                "fetch_bytes": t["fetch_bytes_per_launch"], "write_bytes": t["write_bytes_per_launch"],
                "source": "profiles/r01_gemm_traffic.json"}
    except (OSError, KeyError, ValueError):
        return None


class GemmTimer:
    """HIP-event timing of every MFMA-GEMM launch inside the timed region (stream = torch's current stream,
    which is the stream the C-ABI launches on)."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        from transformers_amd import ops

        inner = ops.raw_gemm
        timer = self

        def timed_gemm(a, b, *, a_km=False, b_kn=False, **kw):
            if not timer.enabled:
                return inner(a, b, a_km=a_km, b_kn=b_kn, **kw)
            (k, m) = a.shape if a_km else (a.shape[1], a.shape[0])
            n = b.shape[1] if b_kn else b.shape[0]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = inner(a, b, a_km=a_km, b_kn=b_kn, **kw)
            e.record()
            timer.records.append((2.0 * m * n * k, s, e, 2.0 * (m * k + n * k + m * n)))
            return out

        ops.raw_gemm = timed_gemm
        # modules captured `ops.raw_gemm` by attribute lookup at call time (ops.raw_gemm(...)), so this is enough

    def summary(self):
        if not self.records:
            return None
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return dict(launches=len(self.records), flops=fl, ms=ms, bytes=sum(r[3] for r in self.records))


def cpu_baseline(cfg_dict, threads=None):
    """Reference eager path on the host cores: ONE decoder layer at (1, seq, hidden), bf16, fwd+bwd."""
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    threads = threads or os.cpu_count() or 1
    torch.set_num_threads(threads)
    c = {k: v for k, v in cfg_dict.items() if k not in ("batch", "seq")}
    cfg = LlamaConfig(**c, attn_implementation="eager")
    seq = cfg_dict["seq"]
    torch.manual_seed(0)
    layer = LlamaDecoderLayer(cfg, 0).to(torch.bfloat16)
    rot = LlamaRotaryEmbedding(cfg)
    x = torch.randn(1, seq, cfg.hidden_size, dtype=torch.bfloat16, requires_grad=True)
    pos = torch.arange(seq)[None]
    pe = rot(x, pos)
    mask = torch.full((seq, seq), torch.finfo(torch.bfloat16).min, dtype=torch.bfloat16).triu(1)[None, None]

    def step():
        y = layer(x, attention_mask=mask, position_embeddings=pe)
        y.backward(torch.ones_like(y))

    step()  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    layers = cfg_dict["num_hidden_layers"]
    return dict(value=seq / (dt * layers), unit="tokens/s", cores=threads, kind="reference",
                sample=(f"transformers eager LlamaDecoderLayer (Llama-3-8B dims) fwd+bwd, batch 1 x seq {seq}, bf16, "
                        f"{dt:.2f} s/layer x {layers} layers extrapolated; embedding/lm_head/loss excluded"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="llama3-8b", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=int(os.environ.get("TAMD_DDP_BUCKET_MB", "256")))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" on ROCm is RCCL

    import transformers_amd
    from transformers import LlamaConfig, LlamaForCausalLM

    c = CONFIGS[args.config]
    batch, seq = c["batch"], c["seq"]
    cfg = LlamaConfig(**{k: v for k, v in c.items() if k not in ("batch", "seq")}, attn_implementation="eager")
    torch.manual_seed(0)  # identical weights on every rank
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        model = LlamaForCausalLM(cfg)
    torch.set_default_dtype(old)
    model.train()
    transformers_amd.accelerate(model)
    timer = GemmTimer()
    timer.install()
    net = model
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP

        net = DDP(model, device_ids=[local_rank], bucket_cap_mb=args.bucket_mb, gradient_as_bucket_view=True,
                  broadcast_buffers=False, find_unused_parameters=False, static_graph=True)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)  # different synthetic data per rank
    ids = torch.randint(0, cfg.vocab_size, (batch, seq), device=dev, generator=g)

    def step():
        out = net(input_ids=ids, labels=ids, use_cache=False)
        out.loss.backward()
        model.zero_grad(set_to_none=True)
        return out.loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    timer.enabled = rank == 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    tokens = batch * seq * world * args.steps
    value = tokens / dt
    if rank == 0:
        fpt = flops_per_token(c)
        gs = timer.summary()
        roofline = None
        if gs:
            ach = gs["flops"] / (gs["ms"] * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel="tamd::gemm_fl_kernel (csrc/gemm.hip; forward, dX and dW layouts, "
                                                 "all epilogues; avg over the launches of a step)",
                            achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS,
                            traffic=gemm_traffic(), launches_per_step=gs["launches"] // args.steps,
                            avg_launch_ms=gs["ms"] / gs["launches"],
                            avg_launch_tflop=gs["flops"] / gs["launches"] / 1e12,
                            avg_launch_algorithmic_bytes=gs["bytes"] / gs["launches"],
                            gemm_share_of_step_time=gs["ms"] * 1e-3 / dt)
        line = {
            "metric": "fwd+bwd tokens/sec (whole job), Llama-3-8B seq=4096",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: LlamaForCausalLM fwd+bwd incl. lm_head + causal-LM loss, "
                                   f"random-init bf16 weights, per-GPU batch {batch} x seq {seq}",
                       "model": args.config, "global_batch": batch * world, "seq_len": seq,
                       "parallelism": f"dp{world}"},
            "tokens_per_sec_per_gpu": value / world,
            "model_tflops_per_gpu": value / world * fpt / 1e12,
            "mfu_vs_2500TF": value / world * fpt / (PEAK_BF16_TFLOPS * 1e12),
            "loss": float(loss),
            "max_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(c)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
