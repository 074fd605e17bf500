"""Forward + backward of a Llama-3-8B-shaped decoder (a few layers of the real width, real vocabulary) on rows of 4096 tokens
holding ONE sequence, packed documents (position_ids restarting: the reference's padding-free batches, masking_utils.py:728-757,
data/data_collator.py DataCollatorWithFlattening) or a sliding window -- §8 f3 end to end: the mask reaches the kernels as the
two bound planes, whole key tiles outside a document / window are skipped.  One JSON line per layout.

    python tools/packed_step_bench.py [--layers 4] [--batch 8] [--seq 4096] [--steps 5]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import transformers_amd  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--seq", type=int, default=4096)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=args.layers,
                  num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=8192,
                  rope_theta=500000.0, rms_norm_eps=1e-5, attn_implementation="eager")
model = transformers_amd.accelerate(LlamaForCausalLM(cfg).to(torch.bfloat16).to(dev)).train()
b, s = args.batch, args.seq
ids = torch.randint(0, cfg.vocab_size, (b, s), device=dev)


def positions(lens):
    assert sum(lens) == s
    return torch.cat([torch.arange(n, device=dev) for n in lens])[None].expand(b, -1).contiguous()


layouts = [("one sequence per row", None)]
for n in (2048, 1024, 512):
    if s % n == 0 and n < s:
        layouts.append((f"packed documents of {n}", positions([n] * (s // n))))
layouts.append(("packed documents of 3000 + 700 + 396" if s == 4096 else "packed, ragged",
                positions([3000, 700, 396] if s == 4096 else [s - s // 3, s // 3])))


def step(pos):
    kw = {} if pos is None else {"position_ids": pos}
    out = model(input_ids=ids, labels=ids, use_cache=False, **kw)
    out.loss.backward()
    model.zero_grad(set_to_none=True)
    return out.loss


base = None
for name, pos in layouts:
    transformers_amd.fallback_calls(reset=True)
    for _ in range(2):
        loss = step(pos)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(args.steps):
        loss = step(pos)
    en.record()
    torch.cuda.synchronize()
    ms = st.elapsed_time(en) / args.steps
    rec = {"layout": name, "layers": args.layers, "batch": b, "seq": s, "ms_per_step": round(ms, 3),
           "tokens_per_s": round(b * s / ms * 1e3), "loss": round(float(loss.detach()), 4), "fallback_calls": sum(transformers_amd.fallback_calls().values())}
    if base is None:
        base = ms
    else:
        rec["vs_one_sequence"] = round(ms / base, 4)
    print(json.dumps(rec), flush=True)
