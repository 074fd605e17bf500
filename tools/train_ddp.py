#!/usr/bin/env python
"""torchrun + the reference's UNCHANGED `Trainer` on the accelerated model: the data-parallel entry of SURVEY.md
section 8e (one process per GPU, DDP gradient all-reduce over RCCL/xGMI).  Mirrors the reference's own launch test
(tests/trainer/distributed/test_trainer_distributed_ddp.py:57-97: `torchrun --nproc_per_node=N --nnodes=1
--master_port=P script.py args`), with the DDP knobs section 8e asks for (trainer.py:720-737 reads them from
TrainingArguments): ddp_find_unused_parameters=False, ddp_bucket_cap_mb, ddp_broadcast_buffers=False.

    torchrun --nproc_per_node=8 --nnodes=1 --master_addr 127.0.0.1 --master_port 29511 tools/train_ddp.py \
        --config llama3-8b --per_device_train_batch_size 8 --seq 4096 --max_steps 10 --output_dir /tmp/out
    python tools/train_ddp.py --nproc 2 ...      # self-launching form of the same command

Synthetic token ids (no network: no datasets, no checkpoints); random-init weights of the named architecture.
Rank 0 writes <output_dir>/train_ddp.json: losses per step, tokens/s, world size, whether every rank ended with
bit-identical weights (a checksum all-gathered over RCCL)."""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="llama-tiny", choices=["llama-tiny", "llama3-8b"])
    ap.add_argument("--per_device_train_batch_size", type=int, default=2)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--max_steps", type=int, default=4)
    ap.add_argument("--output_dir", default="/tmp/tamd_train_ddp")
    ap.add_argument("--ddp_bucket_cap_mb", type=int, default=256)
    ap.add_argument("--optim", default="tamd_adamw", choices=["tamd_adamw", "adamw_torch_fused"])
    ap.add_argument("--nproc", type=int, default=0, help="self-launch under torchrun with this many ranks")
    ap.add_argument("--emu", action="store_true",
                    help="CPU dry run of this script (gloo + the CPU execution model of the kernels in tests/hipemu)")
    args = ap.parse_args()
    if args.nproc and "RANK" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        argv = [a for i, a in enumerate(sys.argv[1:]) if a != "--nproc" and sys.argv[i] != "--nproc"]
        cmd = [sys.executable, "-m", "torch.distributed.run", f"--nproc_per_node={args.nproc}", "--nnodes=1",
               "--master_addr", "127.0.0.1", f"--master_port={port}", str(Path(__file__).resolve()), *argv]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        raise SystemExit(subprocess.call(cmd, env=env))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    from transformers import LlamaConfig, LlamaForCausalLM, Trainer, TrainerCallback, TrainingArguments

    import transformers_amd
    from bench import CONFIGS

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = "cpu" if args.emu else "cuda"
    if args.emu:
        sys.path.insert(0, str(ROOT / "tests" / "hipemu"))
        from emu_backend import get_emu
        from transformers_amd import ops

        ops._set_backend(get_emu())
        torch.cuda.synchronize = lambda *a, **k: None
    else:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    cfg = LlamaConfig(**CONFIGS[args.config]["model"], attn_implementation="eager")
    torch.manual_seed(0)  # identical initial weights on every rank
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(device):
        model = LlamaForCausalLM(cfg)
    torch.set_default_dtype(old)
    transformers_amd.accelerate(model)

    class Synthetic(torch.utils.data.Dataset):
        def __init__(self, n):
            g = torch.Generator().manual_seed(1234)
            self.x = torch.randint(0, cfg.vocab_size, (n, args.seq), generator=g)

        def __len__(self):
            return len(self.x)

        def __getitem__(self, i):
            return {"input_ids": self.x[i], "labels": self.x[i]}

    class Log(TrainerCallback):
        def __init__(self):
            self.losses, self.t0 = [], None

        def on_step_begin(self, a, state, control, **kw):
            if state.global_step == 1:  # first step = warm-up (allocator, DDP bucket rebuild)
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()

        def on_log(self, a, state, control, logs=None, **kw):
            if logs and "loss" in logs:
                self.losses.append(float(logs["loss"]))

    targs = TrainingArguments(
        output_dir=args.output_dir, max_steps=args.max_steps, per_device_train_batch_size=args.per_device_train_batch_size,
        learning_rate=1e-4, lr_scheduler_type="constant", bf16=False, report_to=[], use_cpu=args.emu, save_strategy="no", logging_steps=1,
        disable_tqdm=True, dataloader_pin_memory=False, ddp_find_unused_parameters=False,
        ddp_bucket_cap_mb=args.ddp_bucket_cap_mb, ddp_broadcast_buffers=False,
        optim="adamw_torch_fused" if args.optim == "adamw_torch_fused" else "adamw_torch")
    optimizers = (None, None)
    if args.optim == "tamd_adamw":
        optimizers = (transformers_amd.TamdAdamW(model.parameters(), lr=1e-4, weight_decay=0.0), None)
    log = Log()
    n_samples = args.max_steps * args.per_device_train_batch_size * world
    trainer = Trainer(model=model, args=targs, train_dataset=Synthetic(n_samples), optimizers=optimizers, callbacks=[log])
    out = trainer.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - (log.t0 or time.perf_counter())
    # every replica must hold the same bits after training: all-gather a checksum
    chk = torch.zeros((), dtype=torch.float64, device=device)
    for p in model.parameters():
        chk += p.detach().double().sum()
    same = True
    if world > 1:
        import torch.distributed as dist

        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        same = all(torch.equal(allc[0], c) for c in allc)
    if rank == 0:
        timed_steps = max(args.max_steps - 1, 1)
        res = dict(config=args.config, world_size=world, steps=out.global_step, losses=log.losses,
                   train_loss=out.training_loss, replicas_identical=bool(same),
                   tokens_per_s=timed_steps * args.per_device_train_batch_size * args.seq * world / max(dt, 1e-9),
                   ddp=type(trainer.model_wrapped).__name__, optimizer=type(getattr(trainer.optimizer, "optimizer", trainer.optimizer)).__name__,
                   attn_implementation=model.config._attn_implementation,
                   # the unchanged Trainer never asked for bucket views or a hook: transformers_amd.accelerate() arranged both
                   # (ddp.install_trainer_dropin); zero_copy_layers counts layer backwards that wrote into the buckets
                   gradient_as_bucket_view=bool(getattr(trainer.model_wrapped, "gradient_as_bucket_view", False)),
                   ddp_zero_copy=dict(transformers_amd.ddp.STATS))
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)
        (Path(args.output_dir) / "train_ddp.json").write_text(json.dumps(res))
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
