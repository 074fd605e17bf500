"""Generate tests/golden/*.npz by running the REFERENCE's own modules (run in the build container only).

The reference tree (/root/reference) cannot be imported here (its `tokenizers` pin, SURVEY.md header table), so
the generator imports the pip-installed transformers 5.15.0 after verifying that every hot-path source file it
exercises is byte-identical to the file under /root/reference/src/transformers (modulo the documented
non-semantic diffs for gpt2/clip listed in SURVEY.md).  Vectors are tiny (fp32 arrays of dtype-rounded values,
compressed npz) and committed together with this script; the GPU box never needs /root/reference.

    python oracle/make_golden.py          # writes tests/golden/*.npz
"""
from __future__ import annotations

import filecmp
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"
REF_SRC = Path("/root/reference/src/transformers")

IDENTICAL = ["models/llama/modeling_llama.py", "models/bert/modeling_bert.py", "models/llava/modeling_llava.py",
             "masking_utils.py", "activations.py", "loss/loss_utils.py", "pytorch_utils.py"]


def check_reference_identity():
    import transformers

    inst = Path(transformers.__file__).parent
    if not REF_SRC.exists():
        print("WARNING: /root/reference not present; cannot re-verify source identity")
        return
    for rel in IDENTICAL:
        if not filecmp.cmp(inst / rel, REF_SRC / rel, shallow=False):
            raise SystemExit(f"installed transformers {rel} differs from the reference tree")
    print(f"transformers {transformers.__version__}: {len(IDENTICAL)} hot-path files byte-identical to {REF_SRC}")


def f32(t):
    return t.detach().float().cpu().numpy()


def save(name, **arrays):
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    print(f"  {name}.npz: " + ", ".join(f"{k}{list(v.shape)}" for k, v in arrays.items()))


def main():
    check_reference_identity()
    from transformers import BertConfig, GPT2Config, LlamaConfig
    from transformers.activations import ACT2FN
    from transformers.loss.loss_utils import ForCausalLMLoss
    from transformers.models.bert import modeling_bert as mb
    from transformers.models.llama import modeling_llama as ml

    bf = torch.bfloat16
    torch.manual_seed(20260921)

    # ---- 1. LlamaRMSNorm (bf16 and fp32)
    for dt, tag in ((bf, "bf16"), (torch.float32, "f32")):
        m = ml.LlamaRMSNorm(256, eps=1e-5).to(dt)
        m.weight.data = (torch.rand(256) + 0.5).to(dt)
        x = (torch.randn(5, 256) * 3).to(dt)
        save(f"rmsnorm_{tag}", x=f32(x), w=f32(m.weight), y=f32(m(x)), eps=np.float32(1e-5))

    # ---- 2. rotary: cos/sin generation + application
    cfg = LlamaConfig(hidden_size=256, num_attention_heads=4, num_key_value_heads=2, head_dim=64,
                      rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, max_position_embeddings=128)
    rot = ml.LlamaRotaryEmbedding(cfg)
    q = torch.randn(2, 4, 24, 64).to(bf)
    k = torch.randn(2, 2, 24, 64).to(bf)
    pos = torch.arange(24)[None]
    cos, sin = rot(q, pos)
    qe, ke = ml.apply_rotary_pos_emb(q, k, cos, sin)
    save("rope_bf16", q=f32(q), k=f32(k), cos=f32(cos), sin=f32(sin), q_out=f32(qe), k_out=f32(ke),
         theta=np.float32(500000.0))

    # ---- 3. activations (known-answer inputs incl. the reference test's [-1, 0, 1, 2, 3] grid)
    x = torch.cat([torch.tensor([-10.0, -1.0, 0.0, 0.1, 1.0, 2.0, 3.0, 10.0]), torch.randn(56) * 3])
    save("activations_f32", x=f32(x), **{name: f32(ACT2FN[name](x)) for name in ("gelu", "gelu_new", "quick_gelu", "silu")})

    # ---- 4. LlamaMLP
    cfg_mlp = LlamaConfig(hidden_size=64, intermediate_size=128)
    mlp = ml.LlamaMLP(cfg_mlp).to(bf)
    x = torch.randn(3, 7, 64).to(bf)
    save("llama_mlp_bf16", x=f32(x), wg=f32(mlp.gate_proj.weight), wu=f32(mlp.up_proj.weight),
         wd=f32(mlp.down_proj.weight), y=f32(mlp(x)))

    # ---- 5. eager attention, Llama flavour (GQA, causal additive mask, fp32 softmax), bf16 and fp32
    class M:  # what eager_attention_forward reads from the module
        num_key_value_groups = 2
        training = False

    for dt, tag in ((bf, "bf16"), (torch.float32, "f32")):
        q = torch.randn(2, 4, 40, 64).to(dt)
        k = torch.randn(2, 2, 40, 64).to(dt)
        v = torch.randn(2, 2, 40, 64).to(dt)
        kv = torch.ones(2, 40, dtype=torch.bool)
        kv[1, 33:] = False
        allow = torch.tril(torch.ones(40, 40, dtype=torch.bool))[None, None] & kv[:, None, None, :]
        mask = torch.where(allow, torch.tensor(0.0, dtype=dt), torch.finfo(dt).min)
        o, w = ml.eager_attention_forward(M(), q, k, v, mask, scaling=64 ** -0.5)
        save(f"llama_attention_{tag}", q=f32(q), k=f32(k), v=f32(v), key_valid=kv.numpy(), out=f32(o))

    # ---- 6. BERT eager attention (softmax in dtype, bidirectional, padding)
    q = torch.randn(2, 2, 30, 64).to(bf)
    k = torch.randn(2, 2, 30, 64).to(bf)
    v = torch.randn(2, 2, 30, 64).to(bf)
    kv = torch.ones(2, 30, dtype=torch.bool)
    kv[0, 25:] = False
    mask = torch.where(kv[:, None, None, :], torch.tensor(0.0, dtype=bf), torch.finfo(bf).min).expand(2, 1, 30, 30)
    o, _ = mb.eager_attention_forward(M(), q, k, v, mask, scaling=64 ** -0.5)
    save("bert_attention_bf16", q=f32(q), k=f32(k), v=f32(v), key_valid=kv.numpy(), out=f32(o))

    # ---- 7. LlamaDecoderLayer forward + input/weight gradients (fp32: exact target; bf16: rounding-faithful)
    lcfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5,
                       rope_parameters={"rope_type": "default", "rope_theta": 500000.0},
                       max_position_embeddings=64, attn_implementation="eager")
    torch.manual_seed(7)
    model = ml.LlamaForCausalLM(lcfg).eval()
    ids = torch.randint(0, 128, (2, 20))
    labels = ids.clone()
    labels[0, :4] = -100
    am = torch.ones(2, 20, dtype=torch.long)
    am[1, 17:] = 0
    sd = {k: f32(v) for k, v in model.state_dict().items()}
    for dt, tag in ((torch.float32, "f32"), (bf, "bf16")):
        m = ml.LlamaForCausalLM(lcfg).to(dt)
        m.load_state_dict(model.state_dict())
        m.train()
        out = m(input_ids=ids, labels=labels, attention_mask=am, use_cache=False, output_hidden_states=True)
        out.loss.backward()
        grads = {"grad." + k: f32(p.grad) for k, p in m.named_parameters()
                 if k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
                          "model.layers.0.input_layernorm.weight", "model.embed_tokens.weight", "lm_head.weight")}
        save(f"llama_model_{tag}", ids=ids.numpy(), labels=labels.numpy(), attention_mask=am.numpy(),
             logits=f32(out.logits), loss=np.float32(out.loss.item()),
             hidden_1=f32(out.hidden_states[1]), **grads,
             **({"sd." + k: v for k, v in sd.items()} if tag == "f32" else {}))

    # ---- 8. ForCausalLMLoss on bf16 logits (the reference upcasts; ignore_index; num_items_in_batch)
    logits = (torch.randn(3, 9, 50) * 2).to(bf)
    lab = torch.randint(0, 50, (3, 9))
    lab[1, 2:5] = -100
    save("causal_lm_loss", logits=f32(logits), labels=lab.numpy(),
         loss_mean=np.float32(ForCausalLMLoss(logits, lab, 50).item()),
         loss_items=np.float32(ForCausalLMLoss(logits, lab, 50, num_items_in_batch=torch.tensor(11)).item()))

    # ---- 9. BERT: embeddings + one encoder layer (post-LN, biases, gelu), eval mode
    bcfg = BertConfig(vocab_size=100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2,
                      intermediate_size=256, max_position_embeddings=40, attn_implementation="eager")
    torch.manual_seed(9)
    bert = mb.BertModel(bcfg).to(bf).eval()
    ids = torch.randint(1, 100, (2, 24))
    am = torch.ones(2, 24, dtype=torch.long)
    am[0, 20:] = 0
    with torch.no_grad():
        emb = bert.embeddings(input_ids=ids)
        out = bert(input_ids=ids, attention_mask=am)
    bsd = {"sd." + k: f32(v) for k, v in bert.state_dict().items() if "pooler" not in k}
    save("bert_layer_bf16", ids=ids.numpy(), attention_mask=am.numpy(), embeddings=f32(emb),
         last_hidden_state=f32(out.last_hidden_state), **bsd)

    # ---- 10. GPT-2 (config 1 plumbing): tiny random-init, CPU fp32 eager logits
    from transformers import AutoModelForCausalLM

    torch.manual_seed(10)
    gcfg = GPT2Config(n_layer=2, n_embd=64, n_head=2, vocab_size=200, n_positions=32)
    gpt = AutoModelForCausalLM.from_config(gcfg, attn_implementation="eager").eval()
    ids = torch.randint(0, 200, (1, 16))
    with torch.no_grad():
        lg = gpt(ids).logits
    save("gpt2_tiny_f32", ids=ids.numpy(), logits=f32(lg), seed=np.int64(10))

    # ---- 11. LLaVA placeholder merge (integer positions, bit-exact)
    emb = torch.randn(1, 12, 16)
    feats = torch.randn(1, 5, 16)
    ids = torch.tensor([[1, 7, 99, 99, 99, 3, 99, 99, 4, 5, 6, 2]])
    mask = (ids == 99).unsqueeze(-1).expand_as(emb)
    merged = emb.masked_scatter(mask, feats)  # models/llava/modeling_llava.py:244-248
    save("llava_merge", ids=ids.numpy(), embeds=f32(emb), feats=f32(feats), merged=f32(merged))

    # ---- 11b. packed sequences: the reference's own index finder + mask functions on restarting position ids
    from transformers import masking_utils as mu

    pos = torch.tensor([[0, 1, 2, 3, 0, 1, 0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]])
    pids = mu.find_packed_sequence_indices(pos)
    fn = mu.and_masks(mu.causal_mask_function, mu.packed_sequence_mask_function(pids))
    bi, qi, ki = torch.meshgrid(torch.arange(2), torch.arange(12), torch.arange(12), indexing="ij")
    dense = fn(bi, torch.zeros_like(bi), qi, ki)
    save("packed_mask", position_ids=pos.numpy(), seq_ids=pids.numpy(), mask=dense.numpy())

    # ---- 12. torch.optim.AdamW (what Trainer builds by default, trainer.py:1783-1799): 3 steps, fp32, single-tensor
    torch.manual_seed(12)
    p0 = torch.randn(64, 48)
    w = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([w], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, foreach=False, fused=False)
    grads, params = [], []
    for _ in range(3):
        gr = torch.randn(64, 48) * 0.5
        w.grad = gr.clone()
        opt.step()
        grads.append(f32(gr))
        params.append(f32(w.data.clone()))
    st = opt.state[w]
    save("adamw_f32", p0=f32(p0), grads=np.stack(grads), params=np.stack(params), exp_avg=f32(st["exp_avg"]),
         exp_avg_sq=f32(st["exp_avg_sq"]), hyper=np.array([3e-3, 0.9, 0.95, 1e-8, 0.1], dtype=np.float64))
    make_adamw_clip()


def make_adamw_clip():
    """13. What an optimizer step of `Trainer` does to the gradients and the parameters (trainer.py:2538-2548
    `accelerator.clip_grad_norm_(model.parameters(), args.max_grad_norm)` = torch.nn.utils.clip_grad_norm_, then
    `optimizer.step()`, trainer.py:1783-1799): 3 steps over three fp32 tensors, max_grad_norm 1.0 -- the first two steps
    clip (norm > 1), the third does not (its gradients are tiny).  `python oracle/make_golden.py adamw_clip` writes only
    this file."""
    torch.manual_seed(13)
    shapes = [(40, 24), (33,), (7, 5, 3)]
    ws = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    p0 = [f32(w.data.clone()).ravel() for w in ws]
    opt = torch.optim.AdamW(ws, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1, foreach=False, fused=False)
    grads, params, norms = [], [], []
    for t in range(3):
        gs = [torch.randn(s) * (0.5 if t < 2 else 1e-3) for s in shapes]
        for w, g in zip(ws, gs):
            w.grad = g.clone()
        norms.append(float(torch.nn.utils.clip_grad_norm_(ws, 1.0, foreach=False)))
        opt.step()
        grads.append(np.concatenate([f32(g).ravel() for g in gs]))
        params.append(np.concatenate([f32(w.data).ravel() for w in ws]))
    save("adamw_clip_f32", p0=np.concatenate(p0), sizes=np.array([int(np.prod(s)) for s in shapes], dtype=np.int64),
         grads=np.stack(grads), params=np.stack(params), norms=np.array(norms, dtype=np.float64),
         exp_avg=np.concatenate([f32(opt.state[w]["exp_avg"]).ravel() for w in ws]),
         exp_avg_sq=np.concatenate([f32(opt.state[w]["exp_avg_sq"]).ravel() for w in ws]),
         hyper=np.array([3e-3, 0.9, 0.95, 1e-8, 0.1, 1.0], dtype=np.float64))


def make_window_chunk_mask():
    """11c (round 6).  The reference's causal sliding-window and chunked mask functions (masking_utils.py:92-113, 134-138,
    161-165), alone and AND-ed with each other and with a packed-sequence mask, evaluated densely by the reference itself.
    `python oracle/make_golden.py window_chunk_mask` writes only this fixture."""
    check_reference_identity()
    from transformers import masking_utils as mu

    s, b = 23, 2
    left = torch.tensor([0, 3])
    pos = torch.tensor([list(range(9)) + list(range(14)), list(range(16)) + list(range(7))])
    pids = mu.find_packed_sequence_indices(pos)
    cases = {"window5": mu.sliding_window_causal_mask_function(5),
             "window1": mu.sliding_window_causal_mask_function(1),
             "chunk6_left": mu.chunked_causal_mask_function(6, left),
             "window7_chunk10_packed": mu.and_masks(mu.sliding_window_causal_mask_function(7), mu.chunked_overlay(10, left),
                                                    mu.packed_sequence_mask_function(pids))}
    bi, qi, ki = torch.meshgrid(torch.arange(b), torch.arange(s), torch.arange(s), indexing="ij")
    out = {k: fn(bi, torch.zeros_like(bi), qi, ki).numpy() for k, fn in cases.items()}
    save("window_chunk_mask", left_padding=left.numpy(), seq_ids=pids.numpy(), **out)
    return 0


if __name__ == "__main__":
    if sys.argv[1:] == ["adamw_clip"]:
        sys.exit(make_adamw_clip())
    if sys.argv[1:] == ["window_chunk_mask"]:
        sys.exit(make_window_chunk_mask())
    sys.exit(main())
