"""Pin the oracle (oracle/oracle.py, a numpy restatement) against golden vectors produced by the reference's
own modules (oracle/make_golden.py -> tests/golden/*.npz), then check the kernels against the same vectors.

Tolerances: where the oracle reproduces every rounding point of the reference (RMSNorm, rotary, SwiGLU,
embedding, placeholder merge) the comparison is bit-exact or within one storage-dtype ulp on a tiny fraction of
elements (fp32 summation order inside torch's reductions is not specified); fp32 paths 1e-5; model level 1e-2
(bf16 accumulation of independent roundings over 2 layers)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import oracle as orc  # noqa: E402

from conftest import rel_err  # noqa: E402
from transformers_amd import ops  # noqa: E402

G = ROOT / "tests" / "golden"


def load(name):
    return dict(np.load(G / f"{name}.npz"))


def nrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def frac_diff(a, b):
    return float(np.mean(np.asarray(a) != np.asarray(b)))


# ------------------------------------------------------------------ oracle vs golden (CPU)
def test_oracle_rmsnorm():
    for tag in ("bf16", "f32"):
        g = load(f"rmsnorm_{tag}")
        y = orc.rmsnorm(g["x"], g["w"], float(g["eps"]), tag)
        if tag == "bf16":
            assert frac_diff(y, g["y"]) < 5e-3 and nrel(y, g["y"]) < 1e-3
        else:
            assert nrel(y, g["y"]) < 1e-6


def test_oracle_rope_bit_exact():
    g = load("rope_bf16")
    cos, sin = orc.rope_cos_sin(np.arange(24)[None], 64, float(g["theta"]), "bf16")
    assert frac_diff(cos, g["cos"]) < 2e-3 and frac_diff(sin, g["sin"]) < 2e-3  # libm cos/sin last-ulp cases
    assert np.array_equal(orc.apply_rope(g["q"], g["cos"], g["sin"], "bf16"), g["q_out"])
    assert np.array_equal(orc.apply_rope(g["k"], g["cos"], g["sin"], "bf16"), g["k_out"])


def test_oracle_activations_known_answers():
    g = load("activations_f32")
    for name in ("gelu", "gelu_new", "quick_gelu", "silu"):
        assert np.allclose(orc.ACTS[name](g["x"]), g[name], rtol=2e-5, atol=1e-6), name  # torch evaluates erf/tanh/exp in fp32


def test_oracle_mlp_and_attention():
    g = load("llama_mlp_bf16")
    y = orc.llama_mlp(g["x"], g["wg"], g["wu"], g["wd"], "bf16")
    assert nrel(y, g["y"]) < 4e-3
    for tag in ("bf16", "f32"):
        g = load(f"llama_attention_{tag}")
        mask = orc.attention_mask_bool(2, 40, 40, True, g["key_valid"])
        o = orc.eager_attention(g["q"], g["k"], g["v"], 64 ** -0.5, mask, tag)
        assert nrel(o, g["out"]) < (4e-3 if tag == "bf16" else 2e-6)
    g = load("bert_attention_bf16")
    mask = orc.attention_mask_bool(2, 30, 30, False, g["key_valid"])
    o = orc.eager_attention(g["q"], g["k"], g["v"], 64 ** -0.5, mask, "bf16", softmax_in_fp32=False)
    assert nrel(o, g["out"]) < 4e-3


def test_oracle_loss_and_merge():
    g = load("causal_lm_loss")
    assert abs(orc.causal_lm_loss(g["logits"], g["labels"]) - g["loss_mean"]) < 1e-5
    assert abs(orc.causal_lm_loss(g["logits"], g["labels"], num_items_in_batch=11) - g["loss_items"]) < 1e-5
    g = load("llava_merge")
    assert np.array_equal(orc.llava_merge(g["ids"], g["embeds"], g["feats"], 99), g["merged"])


LCFG = dict(num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5,
            rope_theta=500000.0)


def test_oracle_llama_model():
    g32 = load("llama_model_f32")
    sd = {k[3:]: v for k, v in g32.items() if k.startswith("sd.")}
    kv = g32["attention_mask"].astype(bool)
    logits = orc.llama_model_logits(g32["ids"], sd, LCFG, "f32", kv)
    assert nrel(logits[kv], g32["logits"][kv]) < 1e-5
    assert abs(orc.causal_lm_loss(logits, g32["labels"]) - g32["loss"]) < 1e-5
    gb = load("llama_model_bf16")
    sdb = {k: orc.rnd_bf16(v) for k, v in sd.items()}
    lb = orc.llama_model_logits(gb["ids"], sdb, LCFG, "bf16", kv)
    assert nrel(lb[kv], gb["logits"][kv]) < 1e-2


def test_oracle_bert_layer():
    g = load("bert_layer_bf16")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    ids = g["ids"]
    b, s = ids.shape
    emb = orc.bert_embeddings(ids, np.zeros_like(ids), np.broadcast_to(np.arange(s), (b, s)),
                              sd["embeddings.word_embeddings.weight"], sd["embeddings.token_type_embeddings.weight"],
                              sd["embeddings.position_embeddings.weight"], sd["embeddings.LayerNorm.weight"],
                              sd["embeddings.LayerNorm.bias"], 1e-12, "bf16")
    assert nrel(emb, g["embeddings"]) < 3e-3
    p = {k[len("encoder.layer.0."):]: v for k, v in sd.items() if k.startswith("encoder.layer.0.")}
    kv = g["attention_mask"].astype(bool)
    out = orc.bert_layer(g["embeddings"], p, 2, 1e-12, kv, "bf16")
    assert nrel(out[kv], g["last_hidden_state"][kv]) < 6e-3


# ------------------------------------------------------------------ kernels vs golden (emu on CPU, hip on GPU)
def T(a, env, dtype=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(env.device)


def test_kernel_rmsnorm_rope_golden(env):
    g = load("rmsnorm_bf16")
    y = ops.rmsnorm(T(g["x"], env), T(g["w"], env), float(g["eps"]))
    assert frac_diff(y.float().cpu().numpy(), g["y"]) < 5e-3
    g = load("rope_bf16")
    # kernel layout: [B,S,H,D] rows of a fused projection buffer; golden is [B,H,S,D]
    q = T(g["q"], env).permute(0, 2, 1, 3).contiguous()
    k = T(g["k"], env).permute(0, 2, 1, 3).contiguous()
    qk = torch.cat([q.view(2, 24, -1), k.view(2, 24, -1)], -1).contiguous()
    ops.raw_rope_(qk.view(48, -1), T(g["cos"], env), T(g["sin"], env), 24, 6, 64)
    got_q = qk[..., :256].view(2, 24, 4, 64).permute(0, 2, 1, 3).float().cpu().numpy()
    got_k = qk[..., 256:].view(2, 24, 2, 64).permute(0, 2, 1, 3).float().cpu().numpy()
    assert np.array_equal(got_q, g["q_out"]) and np.array_equal(got_k, g["k_out"])  # bit-exact


def test_kernel_attention_golden(env):
    for name, causal in (("llama_attention_bf16", True), ("bert_attention_bf16", False)):
        g = load(name)
        q, k, v = (T(g[n], env).transpose(1, 2) for n in ("q", "k", "v"))
        kv = torch.from_numpy(g["key_valid"]).to(env.device)
        o = ops.attention(q, k, v, 64 ** -0.5, causal, kv)
        valid = g["key_valid"]  # padded query rows are don't-care (tests/test_modeling_common.py:470-500)
        got, want = o.float().cpu().numpy(), g["out"]
        if not causal:
            assert nrel(got[valid], want[valid]) < 8e-3
        else:
            assert nrel(got, want) < 8e-3
    g = load("llama_attention_f32")  # exact target: the kernel's fp32-accumulate path vs the fp32 reference
    q, k, v = (T(g[n], env).transpose(1, 2) for n in ("q", "k", "v"))
    o = ops.attention(q, k, v, 64 ** -0.5, True, torch.from_numpy(g["key_valid"]).to(env.device))
    mask = orc.attention_mask_bool(2, 40, 40, True, g["key_valid"])
    exact = orc.exact_attention(orc.rnd_bf16(g["q"]), orc.rnd_bf16(g["k"]), orc.rnd_bf16(g["v"]), 64 ** -0.5, mask)
    assert nrel(o.float().cpu().numpy(), exact) < 4e-3


def test_kernel_llama_model_golden(env):
    """Whole tiny model through AutoModel-level classes + accelerate(), against the reference's own outputs."""
    import transformers_amd
    from transformers import LlamaConfig, LlamaForCausalLM

    g32, gb = load("llama_model_f32"), load("llama_model_bf16")
    cfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                      num_attention_heads=2, num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5,
                      rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, max_position_embeddings=64,
                      attn_implementation="eager")
    m = LlamaForCausalLM(cfg)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g32.items() if k.startswith("sd.")})
    m = m.bfloat16().to(env.device).train()
    transformers_amd.accelerate(m)
    ids = torch.from_numpy(g32["ids"]).to(env.device)
    labels = torch.from_numpy(g32["labels"]).to(env.device)
    am = torch.from_numpy(g32["attention_mask"]).to(env.device)
    out = m(input_ids=ids, labels=labels, attention_mask=am, use_cache=False)
    out.loss.backward()
    kv = g32["attention_mask"].astype(bool)
    got = out.logits.detach().float().cpu().numpy()
    e_fast, e_ref = nrel(got[kv], g32["logits"][kv]), nrel(gb["logits"][kv], g32["logits"][kv])
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    assert abs(out.loss.item() - float(g32["loss"])) < 3e-3 * float(g32["loss"])
    # bit-exact integer paths: embedding rows selected, label shift / ignore handling (loss over the same tokens)
    emb = m.model.embed_tokens(ids)
    assert torch.equal(emb, m.model.embed_tokens.weight[ids])
    for k in [k for k in gb if k.startswith("grad.")]:
        p = dict(m.named_parameters())[k[5:]]
        ef, er = nrel(p.grad.float().cpu().numpy(), g32[k]), nrel(gb[k], g32[k])
        assert ef <= 1.25 * er + 2e-3, (k, ef, er)


def test_gpt2_golden_cpu_plumbing():
    from transformers import AutoModelForCausalLM, GPT2Config

    import transformers_amd  # noqa: F401  (imported: must not disturb the CPU eager path)

    g = load("gpt2_tiny_f32")
    torch.manual_seed(int(g["seed"]))
    cfg = GPT2Config(n_layer=2, n_embd=64, n_head=2, vocab_size=200, n_positions=32)
    m = AutoModelForCausalLM.from_config(cfg, attn_implementation="eager").eval()
    with torch.no_grad():
        lg = m(torch.from_numpy(g["ids"])).logits
    assert np.allclose(lg.numpy(), g["logits"], atol=1e-5)


def test_oracle_adamw_golden():
    """torch.optim.AdamW (single-tensor, fp32) for three steps vs the oracle's restatement of the update rule."""
    g = load("adamw_f32")
    lr, b1, b2, eps, wd = (float(x) for x in g["hyper"])
    p, m, v = g["p0"], np.zeros_like(g["p0"]), np.zeros_like(g["p0"])
    for t in range(3):
        p, m, v = orc.adamw_step(p, g["grads"][t], m, v, lr, b1, b2, eps, wd, t + 1)
        assert nrel(p, g["params"][t]) < 5e-7, t
    assert nrel(m, g["exp_avg"]) < 5e-7 and nrel(v, g["exp_avg_sq"]) < 5e-7  # a few fp32 ulps (lerp form)


def test_kernel_adamw_golden(env):
    """The fused AdamW kernel against the same golden run (fp32 storage: no rounding beyond fp32 arithmetic)."""
    import transformers_amd

    g = load("adamw_f32")
    lr, b1, b2, eps, wd = (float(x) for x in g["hyper"])
    w = torch.nn.Parameter(torch.from_numpy(g["p0"]).to(env.device))
    opt = transformers_amd.TamdAdamW([w], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    for t in range(3):
        w.grad = torch.from_numpy(g["grads"][t]).to(env.device)
        opt.step()
        assert nrel(w.detach().cpu().numpy(), g["params"][t]) < 5e-7, t
    st = opt.state[w]
    assert nrel(st["exp_avg"].cpu().numpy(), g["exp_avg"]) < 5e-7
    assert nrel(st["exp_avg_sq"].cpu().numpy(), g["exp_avg_sq"]) < 5e-7


def _split(flat, sizes):
    out, o = [], 0
    for k in sizes:
        out.append(flat[o:o + int(k)])
        o += int(k)
    return out


def test_oracle_adamw_clip_golden():
    """torch.nn.utils.clip_grad_norm_(max_norm=1) + torch.optim.AdamW over three tensors for three steps (two clipped, one
    not) vs the oracle's restatement: the clip coefficient scales the gradient, then the AdamW rule."""
    g = load("adamw_clip_f32")
    lr, b1, b2, eps, wd, max_norm = (float(x) for x in g["hyper"])
    sizes = g["sizes"]
    p, m, v = g["p0"], np.zeros_like(g["p0"]), np.zeros_like(g["p0"])
    clipped = []
    for t in range(3):
        norm, coef = orc.clip_grad_norm(_split(g["grads"][t], sizes), max_norm)
        assert abs(float(norm) - g["norms"][t]) < 2e-6 * g["norms"][t], t
        clipped.append(float(coef) < 1.0)
        p, m, v = orc.adamw_step(p, g["grads"][t] * coef, m, v, lr, b1, b2, eps, wd, t + 1)
        assert nrel(p, g["params"][t]) < 5e-7, t
    assert clipped == [True, True, False]
    assert nrel(m, g["exp_avg"]) < 5e-7 and nrel(v, g["exp_avg_sq"]) < 5e-7


@pytest.mark.parametrize("route", ["fused", "clip_then_step"])
def test_kernel_adamw_clip_golden(env, route):
    """The same golden run through the kernels: `fused` = TamdAdamW(max_grad_norm=1) (norm + coefficient in device memory,
    applied inside the multi-tensor AdamW launch), `clip_then_step` = transformers_amd.optim.clip_grad_norm_ (scales the
    gradients in place, as the reference does) followed by the plain step."""
    import transformers_amd
    from transformers_amd import optim

    g = load("adamw_clip_f32")
    lr, b1, b2, eps, wd, max_norm = (float(x) for x in g["hyper"])
    sizes = g["sizes"]
    ws = [torch.nn.Parameter(torch.from_numpy(x.copy()).to(env.device)) for x in _split(g["p0"], sizes)]
    opt = transformers_amd.TamdAdamW(ws, lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd,
                                     max_grad_norm=max_norm if route == "fused" else None)
    for t in range(3):
        for w, gr in zip(ws, _split(g["grads"][t], sizes)):
            w.grad = torch.from_numpy(gr.copy()).to(env.device)
        if route == "fused":
            opt.step()
            norm = opt.grad_norm
            for w, gr in zip(ws, _split(g["grads"][t], sizes)):  # the clipped gradient is never written
                assert np.array_equal(w.grad.cpu().numpy(), gr)
        else:
            norm = optim.clip_grad_norm_(ws, max_norm)
            opt.step()
        assert abs(float(norm) - g["norms"][t]) < 2e-6 * g["norms"][t], t
        got = np.concatenate([w.detach().cpu().numpy().ravel() for w in ws])
        assert nrel(got, g["params"][t]) < 5e-7, t
    assert len(opt._tables) == 1  # three tensors, ONE launch
    m = np.concatenate([opt.state[w]["exp_avg"].cpu().numpy().ravel() for w in ws])
    v = np.concatenate([opt.state[w]["exp_avg_sq"].cpu().numpy().ravel() for w in ws])
    assert nrel(m, g["exp_avg"]) < 5e-7 and nrel(v, g["exp_avg_sq"]) < 5e-7


def test_oracle_packed_mask_golden():
    """The reference's packed-sequence index finder and its and_masks(causal, packed) mask on restarting position ids,
    against the oracle's restatement; and the two bound arrays the kernels take describe exactly that mask."""
    g = load("packed_mask")
    ids = orc.packed_sequence_ids(g["position_ids"])
    assert np.array_equal(ids, g["seq_ids"])
    mask = orc.packed_attention_mask_bool(ids)
    assert np.array_equal(mask[:, 0], g["mask"])
    lo, hi = orc.packed_bounds(ids)
    s = ids.shape[1]
    qi, ki = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    for b in range(ids.shape[0]):
        from_q = (lo[b][:, None] <= ki) & (ki <= qi)              # q_start[q] <= k <= q
        from_k = (ki <= qi) & (qi <= hi[b][None, :])              # k <= q <= k_end[k]
        assert np.array_equal(from_q, g["mask"][b]) and np.array_equal(from_k, g["mask"][b])
    # the device-side construction used by the product path gives the same arrays
    assert np.array_equal(ops.packed_q_start(torch.from_numpy(ids)).numpy(), np.stack((lo, hi)))


def test_oracle_window_and_chunk_mask_golden():
    """The reference's causal sliding-window / chunked mask functions (alone, and AND-ed with a packed-sequence mask) evaluated
    densely by the reference itself (oracle/make_golden.py window_chunk_mask), against the oracle's restatement of the two
    formulas; every one of them is exactly `first[q] <= k <= q`, and the device-side constructions of the product path
    (ops.sliding_window_q_start / chunked_q_start / packed_q_start / intersect_q_start) give those two arrays."""
    g = load("window_chunk_mask")
    left, ids = g["left_padding"], g["seq_ids"]
    b, s = ids.shape
    w5 = np.broadcast_to(orc.sliding_window_mask_bool(s, 5), (b, s, s))
    assert np.array_equal(w5, g["window5"])
    assert np.array_equal(np.broadcast_to(orc.sliding_window_mask_bool(s, 1), (b, s, s)), g["window1"])
    assert np.array_equal(g["window1"][0], np.eye(s, dtype=bool))
    c6 = orc.chunked_mask_bool(s, 6, left)
    assert np.array_equal(c6, g["chunk6_left"])
    combo = (np.broadcast_to(orc.sliding_window_mask_bool(s, 7), (b, s, s)) & orc.chunked_mask_bool(s, 10, left)
             & orc.packed_attention_mask_bool(ids)[:, 0])
    assert np.array_equal(combo, g["window7_chunk10_packed"])
    dev = torch.device("cpu")
    lt, it = torch.from_numpy(left), torch.from_numpy(ids)
    assert np.array_equal(ops.sliding_window_q_start(b, s, 5, dev).numpy(), orc.mask_bounds(g["window5"]))
    assert np.array_equal(ops.sliding_window_q_start(b, s, 1, dev).numpy(), orc.mask_bounds(g["window1"]))
    assert np.array_equal(ops.chunked_q_start(b, s, 6, lt, dev).numpy(), orc.mask_bounds(g["chunk6_left"]))
    both = ops.intersect_q_start(ops.intersect_q_start(ops.sliding_window_q_start(b, s, 7, dev), ops.chunked_q_start(b, s, 10, lt, dev)),
                                 ops.packed_q_start(it))
    assert np.array_equal(both.numpy(), orc.mask_bounds(g["window7_chunk10_packed"]))


def test_kernel_packed_and_dropout_attention_vs_oracle(env):
    """Flash kernels with q_start (packed rows) and with in-kernel dropout against the oracle's exact restatements."""
    rng = np.random.default_rng(5)
    b, s, hq, hkv, d = (2, 320, 4, 2, 64) if env.big else (2, 96, 2, 1, 64)
    pos = np.stack([np.concatenate([np.arange(n) for n in lens]) for lens in
                    (([200, 120], [64, 1, 255]) if env.big else ([60, 36], [32, 1, 63]))])
    ids = orc.packed_sequence_ids(pos)
    q, k, v = (torch.from_numpy(rng.standard_normal((b, s, h, d)).astype(np.float32)).bfloat16() for h in (hq, hkv, hkv))
    scale = d ** -0.5
    dev = env.device
    qs = ops.packed_q_start(torch.from_numpy(ids).to(dev))
    o, _ = ops.raw_attn_fwd(q.to(dev), k.to(dev), v.to(dev), scale, True, q_start=qs)
    tr = lambda t: t.float().numpy().transpose(0, 2, 1, 3)  # [B,S,H,D] -> [B,H,S,D]
    want = orc.exact_attention(tr(q), tr(k), tr(v), scale, orc.packed_attention_mask_bool(ids))
    assert nrel(o.float().cpu().numpy(), want) < 5e-3
    # the same planes carry a sliding window and chunks (masking_utils.py:92-113): window AND chunks AND packing at once
    window, chunk, left = (70, 96, np.array([0, 5])) if env.big else (20, 28, np.array([0, 5]))
    dense = (orc.sliding_window_mask_bool(s, window)[None] & orc.chunked_mask_bool(s, chunk, left)
             & orc.packed_attention_mask_bool(ids)[:, 0])
    planes = ops.intersect_q_start(ops.intersect_q_start(ops.sliding_window_q_start(b, s, window, dev),
                                                         ops.chunked_q_start(b, s, chunk, torch.from_numpy(left), dev)), qs)
    assert np.array_equal(planes.cpu().numpy(), orc.mask_bounds(dense))
    o, _ = ops.raw_attn_fwd(q.to(dev), k.to(dev), v.to(dev), scale, True, q_start=planes)
    want = orc.exact_attention(tr(q), tr(k), tr(v), scale, dense[:, None])
    assert nrel(o.float().cpu().numpy(), want) < 5e-3
    # dropout: the keep mask is the exported hash; the oracle applies it after the softmax
    p, seed = 0.25, 0x5EED1234ABCD
    keep = ops.dropout_keep_mask(seed, b, hq, s, s, p).numpy()
    o, _ = ops.raw_attn_fwd(q.to(dev), k.to(dev), v.to(dev), scale, True, dropout_p=p, seed=seed)
    want = orc.dropout_attention_exact(tr(q), tr(k), tr(v), scale, orc.attention_mask_bool(b, s, s, True), keep, p)
    assert nrel(o.float().cpu().numpy(), want) < 6e-3
