"""N>1 path on CPU: world_size-2 `gloo` DDP over the accelerated model (kernels on the CPU execution model).

Checks what the 8-GPU RCCL run relies on: (1) every parameter of the swapped modules (incl. the fused-QKV /
gate|up views and the custom autograd nodes) takes part in DDP's bucketed all-reduce, (2) the reduced gradient
equals the mean of the per-rank gradients computed without DDP, (3) replicas stay bit-identical after an SGD step."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build_model():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, head_dim=64, max_position_embeddings=128,
                      rms_norm_eps=1e-5, attn_implementation="eager")
    return LlamaForCausalLM(cfg).bfloat16().train(), cfg


def _worker(rank, world, port, out_dir):
    for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hipemu"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import transformers_amd
    from emu_backend import emu_backend
    from torch.nn.parallel import DistributedDataParallel as DDP

    with emu_backend():
        model, cfg = _build_model()
        transformers_amd.accelerate(model)
        ddp = DDP(model, bucket_cap_mb=1, gradient_as_bucket_view=True, broadcast_buffers=False,
                  find_unused_parameters=False, static_graph=True)
        g = torch.Generator().manual_seed(100 + rank)
        ids = torch.randint(0, cfg.vocab_size, (2, 48), generator=g)
        out = ddp(input_ids=ids, labels=ids, use_cache=False)
        out.loss.backward()
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
        assert all(p.grad is not None for p in model.parameters())
        with torch.no_grad():
            for p in model.parameters():
                p.add_(p.grad, alpha=-0.1)
        weights = {n: p.detach().float().clone() for n, p in model.named_parameters()}
    torch.save({"grads": grads, "weights": weights, "loss": out.loss.item()}, f"{out_dir}/rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_ddp_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    # (3) replicas identical
    for n in r[0]["weights"]:
        assert torch.equal(r[0]["weights"][n], r[1]["weights"][n]), n
        assert torch.equal(r[0]["grads"][n], r[1]["grads"][n]), n
    # (2) reduced gradient == mean of single-process gradients on the two data shards
    for p in (ROOT, ROOT / "tests", ROOT / "tests" / "hipemu"):
        sys.path.insert(0, str(p))
    import transformers_amd
    from emu_backend import emu_backend

    per_rank = []
    with emu_backend():
        for rank in range(world):
            model, cfg = _build_model()
            transformers_amd.accelerate(model)
            g = torch.Generator().manual_seed(100 + rank)
            ids = torch.randint(0, cfg.vocab_size, (2, 48), generator=g)
            model(input_ids=ids, labels=ids, use_cache=False).loss.backward()
            per_rank.append({n: p.grad.detach().float() for n, p in model.named_parameters()})
    for n, gd in r[0]["grads"].items():
        want = (per_rank[0][n] + per_rank[1][n]) / 2
        err = (gd - want).norm() / want.norm().clamp_min(1e-12)
        assert err < 1e-2, (n, err.item())  # bf16 bucket arithmetic
