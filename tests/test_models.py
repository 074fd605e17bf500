"""Model-level parity through the unchanged reference classes (AutoModel / LlamaForCausalLM + accelerate()).

Gate (SURVEY.md §8c "parity noise floor"): in bf16, our logits' error against the reference's fp32 eager path
must not exceed ~1.1x the error of the reference's own bf16 eager path against the same fp32 run; the
reference's backend-parity rule (tests/test_modeling_common.py:199-231: bf16 atol/rtol 1e-2) is the regression
gate for hidden states; losses agree to 2e-3 relative; integer paths are exact."""
import copy

import numpy as np
import pytest
import torch

import transformers_amd
from conftest import record, rel_err
from transformers import LlamaConfig, LlamaForCausalLM


def tiny_llama(big):
    if big:
        return LlamaConfig(vocab_size=4096, hidden_size=1024, intermediate_size=2816, num_hidden_layers=3,
                           num_attention_heads=8, num_key_value_heads=2, head_dim=128, max_position_embeddings=2048,
                           rms_norm_eps=1e-5, attn_implementation="eager")
    return LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2, head_dim=64, max_position_embeddings=512,
                       rms_norm_eps=1e-5, attn_implementation="eager")


@pytest.mark.parametrize("padding", [False, True, "left", "ragged"])
def test_llama_forward_backward_parity(env, padding):
    torch.manual_seed(0)
    cfg = tiny_llama(env.big)
    ref = LlamaForCausalLM(cfg).bfloat16().train()
    ref32 = copy.deepcopy(ref).float()
    fast = copy.deepcopy(ref).to(env.device)
    b, s = (4, 1024) if env.big else (2, 96)
    if padding == "ragged":  # token count not a multiple of 8 (dynamic padding in a DataLoader): tails everywhere
        b, s = (3, 333) if env.big else (3, 37)
    ids = torch.randint(0, cfg.vocab_size, (b, s))
    labels = ids.clone()
    labels[0, :10] = -100
    am = None
    if padding == "left":
        # the reference's backend-parity test pads on both sides (tests/test_modeling_common.py:158, 361-365): with
        # left padding the padded query rows see no key at all (eager gives them uniform attention, the kernel 0);
        # like the reference test (:498-503) only the valid rows are compared -- and no gradient may be poisoned
        am = torch.ones(b, s, dtype=torch.long)
        am[1, :16] = 0
        am[b - 1, :2] = 0
        labels[am == 0] = -100
        # (the last padded position predicts the first real token: keep that garbage-in term out of the loss too)
        labels[1, 16] = -100
        labels[b - 1, 2] = -100
    elif padding:
        am = torch.ones(b, s, dtype=torch.long)
        am[1, s - 16:] = 0
        labels[1, s - 16:] = -100
    o_ref = ref(input_ids=ids, labels=labels, attention_mask=am, use_cache=False)
    o_ref.loss.backward()
    o32 = ref32(input_ids=ids, labels=labels, attention_mask=am, use_cache=False)
    o32.loss.backward()
    transformers_amd.accelerate(fast)
    assert fast.config._attn_implementation == "tamd"
    assert type(fast.model.layers[0]).__name__ == "TamdLlamaDecoderLayer"
    dev = env.device
    transformers_amd.fallback_calls(reset=True)
    o = fast(input_ids=ids.to(dev), labels=labels.to(dev), attention_mask=None if am is None else am.to(dev),
             use_cache=False)
    o.loss.backward()
    # the headline architecture must run on the kernels: a regression that silently routes a layer to the reference
    # module's own forward would still produce matching numbers
    assert transformers_amd.fallback_calls() == {}, transformers_amd.fallback_calls()
    valid = torch.ones(b, s, dtype=torch.bool) if am is None else am.bool()
    e_fast = rel_err(o.logits[valid.to(dev)], o32.logits[valid])
    e_ref = rel_err(o_ref.logits[valid], o32.logits[valid])
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    assert abs(o.loss.item() - o32.loss.item()) <= 2e-3 * abs(o32.loss.item()) + 2e-3
    g32 = dict(ref32.named_parameters())
    gref = dict(ref.named_parameters())
    for n, p in fast.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        ef, er = rel_err(p.grad, g32[n].grad), rel_err(gref[n].grad, g32[n].grad)
        assert ef <= 1.25 * er + 2e-3, (n, ef, er)
        record("llama_model", f"{padding}:{n}", ef, er)
    record("llama_model", f"{padding}:logits", e_fast, e_ref)


def test_llama_module_level_path_matches_fused_layer(env):
    """The module-by-module replacements (hooks installed -> no whole-layer fusion) and the fused layer agree."""
    torch.manual_seed(1)
    cfg = tiny_llama(False)
    m = LlamaForCausalLM(cfg).bfloat16().eval().to(env.device)
    transformers_amd.accelerate(m)
    ids = torch.randint(0, cfg.vocab_size, (2, 64)).to(env.device)
    with torch.no_grad():
        a = m(input_ids=ids, use_cache=False).logits
        hs = m(input_ids=ids, use_cache=False, output_hidden_states=True)
        handles = [l.mlp.register_forward_hook(lambda *_: None) for l in m.model.layers]
        b = m(input_ids=ids, use_cache=False).logits
        for h in handles:
            h.remove()
    assert len(hs.hidden_states) == cfg.num_hidden_layers + 1
    assert rel_err(b, a) < 1e-2
    # (not the same bits: the fused layer's rotary kernel hands the attention pre-scaled queries -- one rounding -- where the
    # module-level path lets the attention kernels scale and re-round q; a random-init toy model's logits are near-ties)
    assert (a.argmax(-1) == b.argmax(-1)).float().mean() > 0.93


def test_generate_with_cache_uses_reference_modules(env):
    """A forward with a KV cache (round 4): our layer around the reference's own cache object -- fused q|k|v product, rotary
    kernel, `Cache.update`, the registered attention function, residual adds in the o_proj / down_proj epilogues -- for the
    prefill and for single-token decode steps (M = batch rows: the weight-streaming kernel of csrc/gemv.hip); nothing falls
    back.  Under autograd the cached forward is the reference module's (counted as `kv_cache`)."""
    torch.manual_seed(2)
    cfg = tiny_llama(False)
    ref = LlamaForCausalLM(cfg).bfloat16().eval()
    fast = copy.deepcopy(ref).to(env.device)
    transformers_amd.accelerate(fast)
    ids = torch.randint(0, cfg.vocab_size, (2, 12))
    transformers_amd.fallback_calls(reset=True)
    with torch.no_grad():
        want = ref(input_ids=ids[:, :10], use_cache=True)
        got = fast(input_ids=ids[:, :10].to(env.device), use_cache=True)
        assert rel_err(got.logits, want.logits) < 0.014
        assert got.past_key_values is not None
        for t in (10, 11):  # two decode steps on the caches the prefills returned
            want = ref(input_ids=ids[:, t:t + 1], past_key_values=want.past_key_values, use_cache=True)
            got = fast(input_ids=ids[:, t:t + 1].to(env.device), past_key_values=got.past_key_values, use_cache=True)
            assert got.logits.shape == (2, 1, cfg.vocab_size) and rel_err(got.logits, want.logits) < 0.02, t
    assert transformers_amd.fallback_calls() == {}
    fast.train()  # (autograd on: the cached forward is the reference module's -- and the caller is told once how to avoid it)
    import warnings

    from transformers_amd.models import common

    common._WARNED.clear()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        got = fast(input_ids=ids[:, :10].to(env.device), use_cache=True)
        fast(input_ids=ids[:, :10].to(env.device), use_cache=True)
    assert any("kv_cache" in k for k in transformers_amd.fallback_calls()), transformers_amd.fallback_calls()
    told = [w for w in seen if "use_cache=False" in str(w.message)]
    assert 1 <= len(told) <= 2, [str(w.message) for w in seen]  # (once per module class: layer, and the attention inside it)


@pytest.mark.parametrize("padding_side", [None, "left"])
def test_static_cache_prefill_and_decode_match_sdpa(env, padding_side):
    """with a pre-allocated KV cache (`cache_implementation="static"`) K/V are [B,H,max_cache_len,D]
    while the 2-D mask is [B, tokens so far]: `tamd_mask` must pad the mask to the cache length and the attention
    function must only look at the slots in use.  Prefill and two teacher-forced decode steps against the reference's
    sdpa backend with the same StaticCache class, an independent config object per model; then `generate`."""
    from transformers import StaticCache

    torch.manual_seed(21)
    cfg = tiny_llama(False)
    ref = LlamaForCausalLM(cfg).bfloat16().eval()
    ref.set_attn_implementation("sdpa")
    fast = LlamaForCausalLM(copy.deepcopy(cfg)).bfloat16().eval()
    fast.load_state_dict(ref.state_dict())
    fast = transformers_amd.accelerate(fast.to(env.device))
    assert fast.config is not ref.config and ref.config._attn_implementation == "sdpa"
    dev = env.device
    b, p, steps, max_len = 2, 11, 2, 32
    ids = torch.randint(1, cfg.vocab_size, (b, p + steps))
    am = torch.ones(b, p + steps, dtype=torch.long)
    if padding_side == "left":
        am[1, :4] = 0
    valid = am.bool()

    def run(model, device):
        cache = StaticCache(config=model.config, max_cache_len=max_len)
        outs = []
        with torch.no_grad():
            o = model(input_ids=ids[:, :p].to(device), attention_mask=am[:, :p].to(device), past_key_values=cache,
                      use_cache=True)
            outs.append(o.logits.float().cpu())
            for t in range(steps):
                o = model(input_ids=ids[:, p + t: p + t + 1].to(device), attention_mask=am[:, : p + t + 1].to(device),
                          past_key_values=cache, use_cache=True)
                outs.append(o.logits.float().cpu())
        return torch.cat(outs, dim=1)

    want, got = run(ref, "cpu"), run(fast, dev)
    assert got.shape == want.shape == (b, p + steps, cfg.vocab_size)
    assert rel_err(got[valid], want[valid]) < 0.0099
    # the static path agrees with the dynamic-cache path of the same model (same kernels, sliced K/V)
    with torch.no_grad():
        dyn = fast(input_ids=ids.to(dev), attention_mask=am.to(dev), use_cache=False).logits.float().cpu()
    assert rel_err(got[valid], dyn[valid]) < 1e-2
    gen_kw = dict(max_new_tokens=4, do_sample=False, cache_implementation="static", pad_token_id=0)
    # (generate would wrap a static-cache forward in torch.compile: accelerate() flips the reference's own switch -- the layers
    # are opaque ops, tracing them only breaks the graph per layer; a HIP graph is this package's answer, not a tracer)
    assert fast.generation_config.disable_compile is True and not ref.generation_config.disable_compile
    g_ref = ref.generate(ids[:, :p], attention_mask=am[:, :p], **gen_kw)
    g_fast = fast.generate(ids[:, :p].to(dev), attention_mask=am[:, :p].to(dev), **gen_kw)
    assert g_fast.shape == g_ref.shape
    # random-init logits are nearly flat: bf16 rounding may flip an argmax, after which the continuations differ
    assert (g_fast.cpu()[:, : p + 1] == g_ref[:, : p + 1]).float().mean() > 0.9


def test_autocast_with_fp32_master_weights(env):
    """Trainer(bf16=True) keeps fp32 parameters and runs under autocast; q/k arrive in fp32 after the
    rotary (cos/sin are fp32).  The registered attention function casts to the autocast dtype instead of raising;
    accelerate() on an fp32 model leaves the attention backend alone and says so."""
    import warnings

    torch.manual_seed(22)
    cfg = tiny_llama(False)
    m32 = LlamaForCausalLM(cfg).to(env.device)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        transformers_amd.accelerate(m32)
    assert any("fp32 model" in str(x.message) for x in w)
    assert m32.config._attn_implementation == "eager"
    from transformers_amd.attention import tamd_attention_forward

    q = torch.randn(1, 4, 24, 64, device=env.device)
    k = torch.randn(1, 2, 24, 64, device=env.device)
    mod = m32.model.layers[0].self_attn
    with pytest.raises(transformers_amd.ops.TamdError):
        tamd_attention_forward(mod, q, k, k, None)
    if env.name == "hip":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o, _ = tamd_attention_forward(mod, q, k, k, None, scaling=0.125)
        assert o.dtype == torch.bfloat16 and o.shape == (1, 24, 4, 64)
        want = torch.nn.functional.scaled_dot_product_attention(q, k.repeat_interleave(2, 1), k.repeat_interleave(2, 1),
                                                                is_causal=True, scale=0.125).transpose(1, 2)
        assert rel_err(o, want) < 0.0058


def test_state_dict_keys_and_fused_views_roundtrip(env):
    cfg = tiny_llama(False)
    ref = LlamaForCausalLM(cfg).bfloat16()
    fast = copy.deepcopy(ref).to(env.device)
    keys = list(fast.state_dict().keys())
    transformers_amd.accelerate(fast)
    ids = torch.randint(0, cfg.vocab_size, (1, 32)).to(env.device)
    fast(input_ids=ids, use_cache=False)  # builds the fused QKV / gate|up buffers
    assert list(fast.state_dict().keys()) == keys
    sd = {k: v.detach().cpu().clone() for k, v in fast.state_dict().items()}
    for k, v in ref.state_dict().items():
        assert torch.equal(sd[k], v), k
    # writes through the parameters reach the fused buffers (optimizer / load_state_dict semantics)
    att = fast.model.layers[0].self_attn
    with torch.no_grad():
        att.k_proj.weight.mul_(0).add_(1.0)
    fw = att._fused().weight()
    hq = cfg.num_attention_heads * cfg.head_dim
    assert (fw[hq: hq + att.k_proj.weight.shape[0]] == 1).all()
    assert fast.generation_config.disable_compile is True
    transformers_amd.revert(fast)
    assert type(fast.model.layers[0]).__name__ == "LlamaDecoderLayer" and not fast.generation_config.disable_compile


def _load_tamd(path, env):
    from transformers import AutoModelForCausalLM

    m = AutoModelForCausalLM.from_pretrained(path, dtype=torch.bfloat16, attn_implementation="tamd").to(env.device)
    if not env.big:  # CPU model: safetensors hands out memory-mapped (unaligned) storage; a GPU copy is always aligned
        for prm in m.parameters():
            prm.data = prm.data.clone()
    return transformers_amd.accelerate(m).eval()


def test_save_pretrained_from_pretrained_roundtrip(env, tmp_path):
    """checkpoints written from an accelerated model (parameters are views of fused QKV /
    gate|up buffers) are ordinary per-parameter safetensors that the reference loads back, and a model loaded with
    `from_pretrained(..., attn_implementation="tamd")` and accelerated gives bit-identical logits."""
    torch.manual_seed(6)
    cfg = tiny_llama(False)
    LlamaForCausalLM(cfg).bfloat16().save_pretrained(tmp_path / "a")  # a reference checkpoint
    fast = _load_tamd(tmp_path / "a", env)
    assert fast.config._attn_implementation == "tamd"
    ids = torch.randint(0, cfg.vocab_size, (2, 48)).to(env.device)
    with torch.no_grad():
        want = fast(input_ids=ids, use_cache=False).logits
    fast.save_pretrained(tmp_path / "b")  # after a forward: the fused buffers exist and own the storage
    back = _load_tamd(tmp_path / "b", env)
    for (k, x), (_, y) in zip(fast.state_dict().items(), back.state_dict().items()):
        assert torch.equal(x.cpu(), y.cpu()), k
    with torch.no_grad():
        got = back(input_ids=ids, use_cache=False).logits
    assert torch.equal(got, want)


def test_idle_output_recorder_hooks_do_not_disable_the_fused_layer(env):
    """The reference leaves its output-capturing hooks on the attention modules after the first call that asks for
    hidden states (utils/output_capturing.py:100-119); they are idle afterwards and must not push every later call
    onto the module-by-module path.  A user hook still does."""
    cfg = tiny_llama(False)
    m = transformers_amd.accelerate(LlamaForCausalLM(cfg).bfloat16().to(env.device)).eval()
    ids = torch.randint(0, cfg.vocab_size, (1, 32)).to(env.device)
    probe = torch.zeros(1, 1, cfg.hidden_size, dtype=torch.bfloat16, device=env.device)
    layer = m.model.layers[0]
    with torch.no_grad():
        plain = m(input_ids=ids, use_cache=False).logits
        hs = m(input_ids=ids, use_cache=False, output_hidden_states=True)
        assert len(hs.hidden_states) == cfg.num_hidden_layers + 1
        assert len(layer.self_attn._forward_hooks) >= 1          # the recorder stayed behind ...
        assert layer._fused_ok(probe, None)                      # ... and is recognised as idle
        assert torch.equal(m(input_ids=ids, use_cache=False).logits, plain)
        h = layer.mlp.register_forward_hook(lambda *_: None)
        assert not layer._fused_ok(probe, None)
        h.remove()


def test_gpt2_cpu_path_untouched():
    """BASELINE config 1: gpt2 eager forward on CPU through AutoModelForCausalLM must run unchanged with the
    package imported (and even after accelerate(): CPU tensors take the reference forward)."""
    from transformers import AutoModelForCausalLM, GPT2Config

    torch.manual_seed(0)
    cfg = GPT2Config(n_layer=2, n_embd=64, n_head=2, vocab_size=300, n_positions=64)
    m = AutoModelForCausalLM.from_config(cfg, attn_implementation="eager").eval()
    ids = torch.randint(0, 300, (1, 16))
    with torch.no_grad():
        want = m(ids).logits
        m2 = copy.deepcopy(m)
        transformers_amd.accelerate(m2, attn_implementation=False)
        got = m2(ids).logits
    assert torch.equal(want, got)


def _grad_parity(fast, ref, ref32, skip=()):
    g32, gref = dict(ref32.named_parameters()), dict(ref.named_parameters())
    for n, p in fast.named_parameters():
        if any(s in n for s in skip) or g32[n].grad is None:
            continue
        assert p.grad is not None, n
        ef, er = rel_err(p.grad, g32[n].grad), rel_err(gref[n].grad, g32[n].grad)
        record("grad_parity:" + type(fast).__name__, n, ef, er)
        # noise-floor gate: ours <= 1.5x the reference's own bf16 error + 2.5e-3 (the additive term
        # covers parameters whose reference error is ~0: biases summed in fp32 by both).  Measured worst cases on MI355X:
        # BERT layer-0 query.weight (its gradient passes the bf16 dS of every layer's attention backward) 0.0124 against a
        # reference error of 0.0080 -- the gate is 0.0145 there, 1.17x the measurement (profiles/r03b); and query.bias, a
        # column sum of dq that nearly cancels (key.bias cancels exactly: softmax shift invariance), so its RELATIVE error
        # amplifies whatever dq carries: 0.0170 against 0.0094 (profiles/r03c) -- those get 2.2x + 3e-3 = 0.0236 (1.39x).
        # Round 4's measurements (profiles/r04_parity.json, MI355X): worst weight bert.encoder.layer.1...key.weight 0.0162
        # against 0.0092 -> gate 0.0163 = 1.006x the measurement; worst query.bias 0.0197 against 0.0095 -> gate 0.0239 =
        # 1.21x: both already inside the "<= 1.25x the worst measurement" VERDICT r4 asks for, so the constants stay
        k, c = (2.2, 3e-3) if n.endswith("query.bias") else (1.5, 2.5e-3)
        assert ef <= k * er + c, (n, ef, er)


@pytest.mark.parametrize("padding_side", ["right", "left", "right-vocab30522"])
def test_bert_masked_lm_parity(env, padding_side):
    """BASELINE config 2 architecture (encoder LayerNorm/GeLU path) at test scale, dropout 0 (parity mode); padding on
    either side as in the reference's backend-parity test (tests/test_modeling_common.py:158, 361-365).  The third case is
    bert-base's REAL masked-LM head (modeling_bert.py:466-497, 939-982): the 30522-row tied decoder over 30528 padded rows +
    the loss, against the reference's fp32 `BertForMaskedLM` on the host -- MI355X only (the CPU execution model would take
    hours over a 768 x 30522 head)."""
    from transformers import BertConfig, BertForMaskedLM

    torch.manual_seed(3)
    big = env.big
    real_vocab = padding_side.endswith("vocab30522")
    if real_vocab and not big:
        pytest.skip("the real 30522-row vocabulary runs on the GPU only")
    padding_side = padding_side.split("-")[0]
    # vocab % 8 == 2 like bert-base's 30522: the tied decoder runs on zero-padded rows (fused_params.PaddedRows)
    cfg = BertConfig(vocab_size=30522 if real_vocab else (1002 if big else 202), hidden_size=768 if big else 128,
                     num_hidden_layers=2, num_attention_heads=12 if big else 2,
                     intermediate_size=3072 if big else 256, max_position_embeddings=512 if big else 64,
                     attn_implementation="eager", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref = BertForMaskedLM(cfg).bfloat16().train()
    ref32 = copy.deepcopy(ref).float()
    fast = copy.deepcopy(ref).to(env.device)
    b, s = (8, 512) if big else (2, 40)
    ids = torch.randint(1, cfg.vocab_size, (b, s))
    am = torch.ones(b, s, dtype=torch.long)
    if padding_side == "right":
        am[0, s - 7:] = 0
    else:
        am[0, :7] = 0
        am[b - 1, :2] = 0
    labels = ids.clone()
    labels[:, ::3] = -100
    labels[am == 0] = -100
    o_ref = ref(input_ids=ids, attention_mask=am, labels=labels)
    o_ref.loss.backward()
    o32 = ref32(input_ids=ids, attention_mask=am, labels=labels)
    o32.loss.backward()
    transformers_amd.accelerate(fast)
    assert type(fast.bert.embeddings).__name__ == "TamdBertEmbeddings"
    dev = env.device
    transformers_amd.fallback_calls(reset=True)
    o = fast(input_ids=ids.to(dev), attention_mask=am.to(dev), labels=labels.to(dev))
    o.loss.backward()
    assert transformers_amd.fallback_calls() == {}, transformers_amd.fallback_calls()  # nothing served by ATen modules
    assert o.logits.shape == o32.logits.shape and o.logits.stride(-2) % 64 == 0  # the [.., V] view of padded rows
    e_fast, e_ref = abs(o.loss.item() - o32.loss.item()), abs(o_ref.loss.item() - o32.loss.item())
    tag = padding_side + (":vocab30522" if real_vocab else "")
    record("bert_model", f"{tag}:loss_abs", e_fast, e_ref)
    assert e_fast <= 1.1 * e_ref + 2e-3 * abs(o32.loss.item()), (e_fast, e_ref)
    v = am.bool()
    e_fast, e_ref = rel_err(o.logits[v.to(dev)], o32.logits[v]), rel_err(o_ref.logits[v], o32.logits[v])
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    record("bert_model", f"{tag}:logits", e_fast, e_ref)
    # key.bias has an exactly-zero gradient (softmax shift invariance): relative error is meaningless there
    _grad_parity(fast, ref, ref32, skip=("key.bias",))


def test_training_with_default_dropout_runs_on_kernels(env):
    """Stock configs train with dropout > 0 (BERT 0.1/0.1; Llama attention_dropout is user-set): the attention
    keep mask is drawn inside the kernels from a seed taken from torch's RNG, so `torch.manual_seed` repeats a
    step bit for bit, a different seed changes it, and the loss stays close to the dropout-free value."""
    from transformers import BertConfig, BertForMaskedLM, LlamaConfig, LlamaForCausalLM

    dev = env.device
    torch.manual_seed(13)
    cfg = BertConfig(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=256, max_position_embeddings=64, attn_implementation="eager")
    assert cfg.attention_probs_dropout_prob == 0.1 and cfg.hidden_dropout_prob == 0.1
    bert = transformers_amd.accelerate(BertForMaskedLM(cfg).bfloat16().to(dev)).train()
    ids = torch.randint(1, 200, (2, 40), device=dev)

    def step(model, seed, **kw):
        torch.manual_seed(seed)
        model.zero_grad()
        out = model(input_ids=ids, labels=ids, **kw)
        out.loss.backward()
        return out.loss.item(), [p.grad.clone() for p in model.parameters() if p.grad is not None]

    l1, g1 = step(bert, 100)
    l2, g2 = step(bert, 100)
    l3, g3 = step(bert, 101)
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert any(not torch.equal(a, b) for a, b in zip(g1, g3))
    l_eval = bert.eval()(input_ids=ids, labels=ids).loss.item()
    assert abs(l1 - l_eval) < 0.25 * abs(l_eval)

    lcfg = LlamaConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, head_dim=64, attention_dropout=0.2,
                       max_position_embeddings=64, attn_implementation="eager")
    llama = transformers_amd.accelerate(LlamaForCausalLM(lcfg).bfloat16().to(dev)).train()
    ids = torch.randint(0, 128, (2, 33), device=dev)
    l1, g1 = step(llama, 7, use_cache=False)
    l2, g2 = step(llama, 7, use_cache=False)
    l3, g3 = step(llama, 8, use_cache=False)
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert any(not torch.equal(a, b) for a, b in zip(g1, g3))  # attention dropout is the only randomness here
    assert all(torch.isfinite(g).all() for g in g1)


def test_packed_sequences_match_reference_and_separate_runs(env):
    """Padding-free batches (position_ids restarting inside a row, no attention_mask): the reference turns them into a
    block-diagonal causal mask (masking_utils.py:728-757, 973-974); `tamd_mask` turns the same information into
    q_start for the kernels.  Checked against the reference's fp32 eager run and against running each sequence alone;
    other mask overlays are refused, not ignored."""
    torch.manual_seed(14)
    cfg = tiny_llama(env.big)
    ref = LlamaForCausalLM(cfg).bfloat16().train()
    ref32 = copy.deepcopy(ref).float()
    fast = transformers_amd.accelerate(copy.deepcopy(ref).to(env.device))
    lens = [300, 77, 135] if env.big else [40, 9, 31]
    s = sum(lens)
    ids = torch.randint(0, cfg.vocab_size, (1, s))
    pos = torch.cat([torch.arange(n) for n in lens])[None]
    labels = ids.clone()
    dev = env.device
    o32 = ref32(input_ids=ids, position_ids=pos, labels=labels, use_cache=False)
    o32.loss.backward()
    oref = ref(input_ids=ids, position_ids=pos, labels=labels, use_cache=False)
    o = fast(input_ids=ids.to(dev), position_ids=pos.to(dev), labels=labels.to(dev), use_cache=False)
    o.loss.backward()
    e_fast, e_ref = rel_err(o.logits, o32.logits), rel_err(oref.logits, o32.logits)
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    g32 = dict(ref32.named_parameters())
    for n, p in fast.named_parameters():
        if "layers.0.self_attn.k_proj" in n or "layers.1.mlp.down_proj" in n:
            assert rel_err(p.grad, g32[n].grad) < 0.026, n
    # every sequence alone (its own forward, positions from 0) reproduces its slice of the packed logits
    fast.eval()
    st = 0
    with torch.no_grad():
        packed = fast(input_ids=ids.to(dev), position_ids=pos.to(dev), use_cache=False).logits
        for n in lens:
            alone = fast(input_ids=ids[:, st:st + n].to(dev), use_cache=False).logits
            assert rel_err(packed[:, st:st + n], alone) < 1e-2
            st += n
        # without position_ids the same tokens are ONE sequence: different logits after the first boundary
        single = fast(input_ids=ids.to(dev), use_cache=False).logits
        assert rel_err(single[:, lens[0]:], packed[:, lens[0]:]) > 5e-2
        # the reference's varlen kwargs (FlashAttentionKwargs: cu_seq_lens_q/k, max_length_q/k) describe the same packing:
        # next to the restarting position_ids they change nothing (bit for bit) ...
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)
        kw = dict(cu_seq_lens_q=cu, cu_seq_lens_k=cu, max_length_q=max(lens), max_length_k=max(lens))
        both = fast(input_ids=ids.to(dev), position_ids=pos.to(dev), use_cache=False, **kw).logits
        assert torch.equal(both, packed)
        # ... and alone (positions 0 .. s-1 run on: the mask factory sees ONE sequence) they still make the attention
        # block-diagonal: every sequence's slice follows its stand-alone run (rotary angles differ by a per-sequence
        # offset, which attention scores do not see), unlike the run without them
        only = fast(input_ids=ids.to(dev), use_cache=False, **kw).logits
        st = lens[0]
        for n in lens[1:]:
            alone = fast(input_ids=ids[:, st:st + n].to(dev), use_cache=False).logits
            assert rel_err(only[:, st:st + n], alone) < 3e-2 < rel_err(single[:, st:st + n], alone)
            st += n
    # an overlay the kernels do not implement is refused loudly
    from transformers.masking_utils import and_masks, causal_mask_function, sliding_window_bidirectional_overlay
    from transformers_amd.attention import tamd_mask
    from transformers_amd.ops import TamdError

    with pytest.raises(TamdError):
        tamd_mask(1, 8, 8, mask_function=and_masks(causal_mask_function, sliding_window_bidirectional_overlay(4)))


@pytest.mark.parametrize("padded", [False, True])
def test_sliding_window_model_through_the_attention_registry(env, padded):
    """Boundary B1 alone on a model family this package has no modules for: a Mistral-shaped decoder with a sliding window
    shorter than the sequence (masking_utils.py:1102-1239 `create_sliding_window_causal_mask` -> the registered mask
    factory -> the kernels' bound planes).  Forward and backward against the reference's own fp32 / bf16 eager runs --
    and the window is really applied: the run with the window differs from the run without it."""
    from transformers import MistralConfig, MistralForCausalLM

    torch.manual_seed(21)
    window = 96 if env.big else 20
    cfg = MistralConfig(vocab_size=512, hidden_size=512 if env.big else 128, intermediate_size=1024 if env.big else 256,
                        num_hidden_layers=2, num_attention_heads=8 if env.big else 2, num_key_value_heads=2 if env.big else 1,
                        head_dim=64, max_position_embeddings=1024, sliding_window=window, attn_implementation="eager")
    ref = MistralForCausalLM(cfg).bfloat16().train()
    ref32 = copy.deepcopy(ref).float()
    fast = transformers_amd.accelerate(copy.deepcopy(ref).to(env.device))
    assert fast.config._attn_implementation == "tamd"
    s = 400 if env.big else 75
    ids = torch.randint(0, cfg.vocab_size, (2, s))
    am = None
    if padded:
        am = torch.ones(2, s, dtype=torch.long)
        am[1, s - (50 if env.big else 13):] = 0
    labels = ids.clone() if am is None else ids.masked_fill(am == 0, -100)
    dev = env.device
    kw = {} if am is None else {"attention_mask": am}
    o32 = ref32(input_ids=ids, labels=labels, use_cache=False, **kw)
    o32.loss.backward()
    oref = ref(input_ids=ids, labels=labels, use_cache=False, **kw)
    kwd = {} if am is None else {"attention_mask": am.to(dev)}
    o = fast(input_ids=ids.to(dev), labels=labels.to(dev), use_cache=False, **kwd)
    o.loss.backward()
    keep = slice(None) if am is None else am.bool()
    e_fast, e_ref = rel_err(o.logits[keep], o32.logits[keep]), rel_err(oref.logits[keep], o32.logits[keep])
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    assert abs(o.loss.item() - o32.loss.item()) < 2e-3 * abs(o32.loss.item()) + 2e-3
    g32 = dict(ref32.named_parameters())
    for n, p in fast.named_parameters():
        if "layers.0.self_attn.k_proj" in n or "layers.0.self_attn.q_proj" in n or "layers.1.self_attn.v_proj" in n:
            assert rel_err(p.grad, g32[n].grad) < 0.03, n
    ref32.config.sliding_window = None
    with torch.no_grad():
        full = ref32(input_ids=ids, use_cache=False, **kw).logits
    assert rel_err(full[0, window:], o32.logits[0, window:]) > 10 * e_fast  # the window matters at this length


def test_bert_post_ln_block_with_hidden_dropout_matches_reference(env):
    """Train mode, hidden_dropout_prob = 0.1 (the shipped value): dense -> dropout -> +residual -> LayerNorm
    (modeling_bert.py:289-293, :347-351) through GEMM+bias and the dropout+add+LayerNorm kernel.  The mask is the
    kernels' counter-based hash (not torch's Philox stream), so the comparison applies the SAME mask -- rebuilt on the
    host from the seed -- inside a restatement of the reference block; eval mode equals the reference bit for bit in
    structure (no dropout)."""
    from transformers import BertConfig
    from transformers.models.bert import modeling_bert as mb

    from transformers_amd import ops
    from transformers_amd.models.bert import TamdBertOutput, TamdBertSelfOutput

    torch.manual_seed(15)
    cfg = BertConfig(hidden_size=128, intermediate_size=256, num_attention_heads=2)
    dev = env.device
    for ref_cls, fast_cls, width in ((mb.BertSelfOutput, TamdBertSelfOutput, 128), (mb.BertOutput, TamdBertOutput, 256)):
        ref = ref_cls(cfg).bfloat16().train()
        fast = copy.deepcopy(ref).to(dev)
        fast.__class__ = fast_cls
        from transformers_amd.models.common import REPLACEMENTS
        for m in fast.modules():
            if type(m) in REPLACEMENTS:
                m.__class__ = REPLACEMENTS[type(m)]
        h = torch.randn(2, 24, width).bfloat16()
        res = torch.randn(2, 24, 128).bfloat16()
        hr, rr = h.clone().requires_grad_(True), res.clone().requires_grad_(True)
        hf, rf = h.clone().to(dev).requires_grad_(True), res.clone().to(dev).requires_grad_(True)
        torch.manual_seed(99)
        seed = ops.dropout_seed()  # what the fast module will draw next from the CPU generator
        torch.manual_seed(99)
        yf = fast(hf, rf)
        keep = ops.hidden_dropout_keep_mask(seed, 48, 128, cfg.hidden_dropout_prob).view(2, 24, 128)
        assert 0.8 < keep.float().mean().item() < 0.98
        d = ref.dense(hr)
        yr = ref.LayerNorm((d * keep / (1 - cfg.hidden_dropout_prob)).to(d.dtype) + rr)
        assert rel_err(yf, yr) < 0.0002
        g = torch.randn_like(yr)
        yr.backward(g)
        yf.backward(g.to(dev))
        assert rel_err(hf.grad, hr.grad) < 7.5e-3 and rel_err(rf.grad, rr.grad) < 7.5e-3
        assert rel_err(fast.dense.weight.grad, ref.dense.weight.grad) < 0.0073
        assert rel_err(fast.LayerNorm.weight.grad, ref.LayerNorm.weight.grad) < 0.0072
        fast.eval(), ref.eval()
        assert rel_err(fast(hf, rf), ref(hr, rr)) < 1e-2


def test_full_size_layer_properties(env):
    """At BASELINE.json's full size (Llama-3-8B decoder layer, batch 8 x seq 4096 on the GPU; a miniature on the CPU
    model) no fp32 reference fits the test budget, so parity is checked through size-independent properties:
    batch rows are independent (bit-exact), the layer is causal (bit-exact), and the backward is linear in dY (a
    power-of-two scale is exact in bf16)."""
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    from transformers_amd.patch import _tables

    big = env.big
    cfg = (LlamaConfig(vocab_size=128, hidden_size=4096, intermediate_size=14336, num_hidden_layers=1,
                       num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, max_position_embeddings=8192,
                       rope_parameters={"rope_type": "default", "rope_theta": 500000.0}, attn_implementation="eager")
           if big else tiny_llama(False))
    b, s = (8, 4096) if big else (3, 160)
    dev = env.device
    torch.manual_seed(17)
    with torch.device(dev):
        layer = LlamaDecoderLayer(cfg, 0).bfloat16()
        rot = LlamaRotaryEmbedding(cfg)
    transformers_amd.attention.register()
    cfg._attn_implementation = "tamd"
    for m in layer.modules():
        r = _tables().get(type(m))
        if r is not None:
            m.__class__ = r
    assert type(layer).__name__ == "TamdLlamaDecoderLayer"
    x = torch.randn(b, s, cfg.hidden_size, device=dev).bfloat16()
    pe = rot(x, torch.arange(s, device=dev)[None])
    with torch.no_grad():
        y = layer(x, position_embeddings=pe)
        assert torch.isfinite(y.float()).all()
        # (1) batch independence: a row alone gives the same bits (GEMM rows, attention heads and norms are per row)
        y1 = layer(x[1:2], position_embeddings=pe)
        assert torch.equal(y1[0], y[1])
        # (2) causality: changing the second half of the sequence leaves the first half untouched
        x2 = x.clone()
        x2[:, s // 2:] = torch.randn_like(x2[:, s // 2:])
        y2 = layer(x2, position_embeddings=pe)
        assert torch.equal(y2[:, : s // 2], y[:, : s // 2]) and not torch.equal(y2[:, s // 2:], y[:, s // 2:])
    # (3) the backward is linear in dY: scaling by 4 scales every gradient by exactly 4
    xg = x.clone().requires_grad_(True)
    dy = (torch.randn_like(x) * 0.25)
    layer(xg, position_embeddings=pe).backward(dy)
    g1 = [xg.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    xg.grad = None
    layer.zero_grad(set_to_none=True)
    layer(xg, position_embeddings=pe).backward(dy * 4)
    g4 = [xg.grad] + [p.grad for p in layer.parameters()]
    for a, c in zip(g1, g4):
        # exact unless an intermediate leaves the normal range of bf16/fp32 (flush-to-zero is not scale-invariant)
        assert ((a * 4) == c).float().mean().item() > 0.9999 and rel_err(a * 4, c) < 1e-4


FULL_SIZE_LAYERS = {
    # BASELINE.json configs 3/4: Llama-3-8B (GQA 32 / 8 heads of 128, I = 14336 = 56 tiles, rope theta 5e5) at seq 4096
    "llama3_8b": dict(kv_heads=8, inter=14336, theta=500000.0, seq=4096),
    # BASELINE.json config 5's language model: Llama-2-7B-shaped (MHA 32 x 128, I = 11008 = 43 tiles -- a ragged last tile
    # column in every MLP GEMM), at LLaVA's prompt length 576 image + 512 text positions = 1088 (4.25 query tiles)
    "llama2_7b": dict(kv_heads=32, inter=11008, theta=10000.0, seq=1088),
}


@pytest.mark.gpu
@pytest.mark.parametrize("which", sorted(FULL_SIZE_LAYERS))
def test_full_size_layer_matches_fp32_reference(which):
    """Parity at the BASELINE configurations' REAL layer dimensions (VERDICT r1 #5, r2 #1c).  One decoder layer at
    (B=1, S): forward output, dX and every dW of the HIP path against the reference's eager layer in fp32 on the host
    cores, next to the reference's own bf16 eager run of the same layer (the SURVEY section 8c noise-floor gate: ours
    <= 1.1x the reference's bf16 error)."""
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

    from transformers_amd.patch import _tables

    shape = FULL_SIZE_LAYERS[which]
    dev = torch.device("cuda:0")
    cfg = LlamaConfig(vocab_size=128, hidden_size=4096, intermediate_size=shape["inter"], num_hidden_layers=1,
                      num_attention_heads=32, num_key_value_heads=shape["kv_heads"], rms_norm_eps=1e-5,
                      max_position_embeddings=8192,
                      rope_parameters={"rope_type": "default", "rope_theta": shape["theta"]}, attn_implementation="eager")
    b, s = 1, shape["seq"]
    torch.manual_seed(23)
    ref = LlamaDecoderLayer(cfg, 0).bfloat16()
    for p in ref.parameters():  # LlamaDecoderLayer alone is not initialised by _init_weights: N(0, 0.02) like the model
        if p.dim() == 2:
            torch.nn.init.normal_(p, std=0.02)
    rot = LlamaRotaryEmbedding(cfg)
    x = torch.randn(b, s, cfg.hidden_size).bfloat16()
    dy = (torch.randn(b, s, cfg.hidden_size) * 0.25).bfloat16()
    pos = torch.arange(s)[None]

    def run_reference(layer, dtype):
        xr = x.detach().to(dtype).clone().requires_grad_(True)  # (a fresh leaf: .to() of the same dtype returns x itself)
        pe = rot(xr, pos)
        mask = torch.full((s, s), torch.finfo(dtype).min, dtype=dtype).triu(1)[None, None]
        y = layer(xr, attention_mask=mask, position_embeddings=pe)
        y.backward(dy.to(dtype))
        return y.detach(), xr.grad, {n: p.grad for n, p in layer.named_parameters()}

    ref32 = copy.deepcopy(ref).float()
    y32, dx32, dw32 = run_reference(ref32, torch.float32)
    yb, dxb, dwb = run_reference(ref, torch.bfloat16)
    fast = copy.deepcopy(ref).to(dev)
    fast.zero_grad(set_to_none=True)
    transformers_amd.attention.register()
    fcfg = copy.deepcopy(cfg)
    fcfg._attn_implementation = "tamd"
    for m in fast.modules():
        r = _tables().get(type(m))
        if r is not None:
            m.__class__ = r
        if hasattr(m, "config"):
            m.config = fcfg
    assert type(fast).__name__ == "TamdLlamaDecoderLayer"
    xf = x.detach().to(dev).requires_grad_(True)
    pe = rot.to(dev)(xf, pos.to(dev))
    assert fast._fused_ok(xf, None)
    yf = fast(xf, position_embeddings=pe)
    yf.backward(dy.to(dev))
    torch.cuda.synchronize()
    e_fast, e_ref = rel_err(yf, y32), rel_err(yb, y32)
    record(f"{which}_layer_full_size", "y", e_fast, e_ref)
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    e_fast, e_ref = rel_err(xf.grad, dx32), rel_err(dxb, dx32)
    record(f"{which}_layer_full_size", "dx", e_fast, e_ref)
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)
    for n, p in fast.named_parameters():
        e_fast, e_ref = rel_err(p.grad, dw32[n]), rel_err(dwb[n], dw32[n])
        record(f"{which}_layer_full_size", f"d {n}", e_fast, e_ref)
        assert e_fast <= 1.25 * e_ref + 2e-3, (n, e_fast, e_ref)


def test_fused_lm_head_loss(env):
    """lm_head GEMM + causal-LM loss chunk by chunk, no [tokens, vocab] tensor.  Same loss and
    gradients as the unfused accelerated model (ignored labels, num_items_in_batch, a ragged last chunk); opt-in,
    instance-level, undone by revert()."""
    from transformers_amd import ops

    torch.manual_seed(16)
    cfg = tiny_llama(env.big)
    base = LlamaForCausalLM(cfg).bfloat16().to(env.device).train()
    fused = transformers_amd.accelerate(copy.deepcopy(base), fused_lm_head_loss=True)
    plain = transformers_amd.accelerate(copy.deepcopy(base))
    assert type(fused) is LlamaForCausalLM and "forward" in fused.__dict__
    b, s = (4, 1000) if env.big else (2, 75)
    ids = torch.randint(0, cfg.vocab_size, (b, s)).to(env.device)
    labels = ids.clone()
    labels[0, :7] = -100
    for kw in ({}, {"num_items_in_batch": torch.tensor(b * s - 11)}):
        fused.zero_grad(set_to_none=True)
        plain.zero_grad(set_to_none=True)
        of = fused(input_ids=ids, labels=labels, use_cache=False, **kw)
        op = plain(input_ids=ids, labels=labels, use_cache=False, **kw)
        assert of.logits is None and op.logits is not None
        assert abs(of.loss.item() - op.loss.item()) <= 2e-3 * abs(op.loss.item())
        (of.loss * 3).backward()
        (op.loss * 3).backward()
        gp = dict(plain.named_parameters())
        for n, p in fused.named_parameters():
            if n in ("lm_head.weight", "model.norm.weight", "model.layers.0.self_attn.q_proj.weight",
                     "model.embed_tokens.weight"):
                assert rel_err(p.grad, gp[n].grad) < 1.5e-2, n
    # ... and against the REFERENCE (VERDICT r2 #1b): loss_utils.py:49-71 on the eager model's logits, fp32 on the host, next
    # to the reference's own bf16 eager run of the same model (SURVEY section 8c noise-floor gate: 1.1x on the loss path,
    # 1.25x on weight gradients)
    ref_bf = copy.deepcopy(base).cpu().train()
    ref_bf.config._attn_implementation = "eager"
    ref32 = copy.deepcopy(ref_bf).float()
    ids_c, labels_c = ids.cpu(), labels.cpu()
    o32 = ref32(input_ids=ids_c, labels=labels_c, use_cache=False)
    obf = ref_bf(input_ids=ids_c, labels=labels_c, use_cache=False)
    o32.loss.backward()
    obf.loss.backward()
    fused.zero_grad(set_to_none=True)
    of = fused(input_ids=ids, labels=labels, use_cache=False)
    of.loss.backward()
    e_fast, e_ref = abs(of.loss.item() - o32.loss.item()), abs(obf.loss.item() - o32.loss.item())
    record("fused_lm_head_loss_vs_reference", "loss_abs", e_fast, e_ref)
    assert e_fast <= 1.1 * e_ref + 2e-3 * abs(o32.loss.item()), (e_fast, e_ref)
    g32, gbf = dict(ref32.named_parameters()), dict(ref_bf.named_parameters())
    for n, p in fused.named_parameters():
        if n in ("lm_head.weight", "model.norm.weight", "model.embed_tokens.weight",
                 "model.layers.0.self_attn.q_proj.weight"):
            e_fast, e_ref = rel_err(p.grad, g32[n].grad), rel_err(gbf[n].grad, g32[n].grad)
            record("fused_lm_head_loss_vs_reference", f"d {n}", e_fast, e_ref)
            assert e_fast <= 1.25 * e_ref + 2e-3, (n, e_fast, e_ref)
    # the op alone, with chunks that do not divide the token count
    h = torch.randn(3, 50, cfg.hidden_size).bfloat16().to(env.device).requires_grad_(True)
    w = base.lm_head.weight.detach().clone().requires_grad_(True)
    lab = torch.randint(0, cfg.vocab_size, (3, 50)).to(env.device)
    loss = ops.fused_linear_cross_entropy(h, w, lab, chunk_tokens=64)
    hr, wr = h.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    ref = ops.causal_lm_loss(ops.linear(hr, wr), lab, cfg.vocab_size)
    assert abs(loss.item() - ref.item()) <= 2e-3 * abs(ref.item())
    loss.backward()
    ref.backward()
    assert rel_err(h.grad, hr.grad) < 0.0065 and rel_err(w.grad, wr.grad) < 0.0065
    # eval / no labels: the reference forward, logits present; revert removes the instance-level forward
    fused.eval()
    with torch.no_grad():
        assert fused(input_ids=ids, labels=labels, use_cache=False).logits is not None
    transformers_amd.revert(fused)
    assert "forward" not in fused.__dict__


def test_clip_vision_tower_hidden_states(env):
    """LLaVA's use of CLIP (models/llava/modeling_llava.py:154-166): output_hidden_states -> hidden_states[-2]."""
    from transformers import CLIPVisionConfig, CLIPVisionModel

    torch.manual_seed(4)
    big = env.big
    cfg = CLIPVisionConfig(hidden_size=1024 if big else 128, intermediate_size=4096 if big else 256,
                           num_hidden_layers=3 if big else 2, num_attention_heads=16 if big else 2,
                           image_size=336 if big else 56, patch_size=14, attn_implementation="eager")
    ref = CLIPVisionModel(cfg).bfloat16().eval()
    ref32 = copy.deepcopy(ref).float()
    fast = copy.deepcopy(ref).to(env.device)
    px = torch.randn(1, 3, cfg.image_size, cfg.image_size)
    with torch.no_grad():
        a = ref(pixel_values=px.bfloat16(), output_hidden_states=True)
        a32 = ref32(pixel_values=px, output_hidden_states=True)
        transformers_amd.accelerate(fast)
        transformers_amd.fallback_calls(reset=True)
        c = fast(pixel_values=px.bfloat16().to(env.device), output_hidden_states=True)
    assert len(c.hidden_states) == cfg.num_hidden_layers + 1
    assert c.hidden_states[-2].shape[1] == (cfg.image_size // 14) ** 2 + 1  # 577 tokens at 336 px: ragged tiles
    # the patch embedding ran as a GEMM over non-overlapping patches (modeling_clip.py:148-154, 209-218): the embeddings are
    # the reference's up to the summation order of 588 products (fp32 accumulation on both sides, one rounding)
    assert any(type(m).__name__ == "TamdCLIPVisionEmbeddings" for m in fast.modules())
    assert not any(k.startswith("TamdCLIPVisionEmbeddings") for k in transformers_amd.fallback_calls())
    assert rel_err(c.hidden_states[0], a32.hidden_states[0]) <= 1.1 * rel_err(a.hidden_states[0], a32.hidden_states[0]) + 1e-3
    e_fast = rel_err(c.hidden_states[-2], a32.hidden_states[-2])
    e_ref = rel_err(a.hidden_states[-2], a32.hidden_states[-2])
    assert e_fast <= 1.1 * e_ref + 1e-3, (e_fast, e_ref)


def test_clip_patch_embedding_weight_gradient(env):
    """Training a CLIP tower (VERDICT r4 missing 6): the patch embedding runs as a GEMM also when its weight wants a gradient
    (models/clip/modeling_clip.py:148-154, 209-218) -- dW through the GEMM's own backward -- and nothing falls back."""
    from transformers import CLIPVisionConfig, CLIPVisionModel

    torch.manual_seed(41)
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, image_size=56,
                           patch_size=14, attn_implementation="eager")
    ref = CLIPVisionModel(cfg).bfloat16().train()
    ref32 = copy.deepcopy(ref).float()
    fast = copy.deepcopy(ref).to(env.device)
    transformers_amd.accelerate(fast)
    transformers_amd.fallback_calls(reset=True)
    px = torch.randn(2, 3, 56, 56)
    g = torch.randn(2, 17, 128)
    outs = {}
    for name, m, x, gg in (("fast", fast, px.bfloat16().to(env.device), g.bfloat16().to(env.device)), ("ref", ref, px.bfloat16(), g.bfloat16()),
                           ("ref32", ref32, px, g)):
        emb = m.embeddings(x)
        emb.backward(gg)
        outs[name] = (emb.detach().float().cpu(), m.embeddings.patch_embedding.weight.grad.detach().float().cpu())
    assert not any(k.startswith("TamdCLIPVisionEmbeddings") for k in transformers_amd.fallback_calls()), transformers_amd.fallback_calls()
    assert outs["fast"][1].shape == outs["ref32"][1].shape == (128, 3, 14, 14)
    for i in (0, 1):  # the embeddings and dW: within 1.1x of the reference's own bf16 error against its fp32 run
        e_fast, e_ref = rel_err(outs["fast"][i], outs["ref32"][i]), rel_err(outs["ref"][i], outs["ref32"][i])
        assert e_fast <= 1.1 * e_ref + 1e-3, (i, e_fast, e_ref)


def test_gpt2_on_kernels(env):
    """GPT-2 blocks (Conv1D = k-major GEMM operand, gelu_new, pre-LN, tied lm_head) through the kernels."""
    from transformers import GPT2Config, GPT2LMHeadModel

    torch.manual_seed(5)
    cfg = GPT2Config(n_layer=2, n_embd=768 if env.big else 128, n_head=12 if env.big else 2, vocab_size=304,
                     n_positions=256, attn_implementation="eager", resid_pdrop=0, embd_pdrop=0, attn_pdrop=0)
    ref = GPT2LMHeadModel(cfg).bfloat16().train()
    ref32 = copy.deepcopy(ref).float()
    fast = copy.deepcopy(ref).to(env.device)
    ids = torch.randint(0, 304, (2, 200 if env.big else 48))
    o_ref = ref(ids, labels=ids)
    o_ref.loss.backward()
    o32 = ref32(ids, labels=ids)
    o32.loss.backward()
    transformers_amd.accelerate(fast)
    o = fast(ids.to(env.device), labels=ids.to(env.device))
    o.loss.backward()
    assert rel_err(o.logits, o32.logits) <= 1.1 * rel_err(o_ref.logits, o32.logits) + 1e-3
    _grad_parity(fast, ref, ref32)


def test_llava_forward_parity(env):
    """BASELINE config 5 architecture at test scale: CLIP vision tower (hidden_states[-2] through the output
    hooks) -> projector -> masked_scatter into the text embeddings -> Llama stack.  Forward only."""
    from transformers import CLIPVisionConfig, LlavaConfig, LlavaForConditionalGeneration

    torch.manual_seed(6)
    big = env.big
    vcfg = CLIPVisionConfig(hidden_size=1024 if big else 128, intermediate_size=4096 if big else 256,
                            num_hidden_layers=3 if big else 2, num_attention_heads=16 if big else 2,
                            image_size=336 if big else 56, patch_size=14)
    tcfg = LlamaConfig(vocab_size=1200, hidden_size=1024 if big else 128, intermediate_size=2816 if big else 256,
                       num_hidden_layers=2, num_attention_heads=8 if big else 2, num_key_value_heads=8 if big else 2,
                       head_dim=128 if big else 64, max_position_embeddings=2048, rms_norm_eps=1e-5)
    cfg = LlavaConfig(vision_config=vcfg, text_config=tcfg, image_token_id=1100, vision_feature_layer=-2,
                      vision_feature_select_strategy="default", attn_implementation="eager")
    ref = LlavaForConditionalGeneration(cfg).bfloat16().eval()
    ref32 = copy.deepcopy(ref).float()
    fast = copy.deepcopy(ref).to(env.device)
    n_img = (vcfg.image_size // 14) ** 2  # 576 at 336 px
    n_txt = 64 if big else 12
    ids = torch.randint(0, 1000, (1, n_txt))
    ids = torch.cat([ids[:, :3], torch.full((1, n_img), 1100), ids[:, 3:]], dim=1)
    px = torch.randn(1, 3, vcfg.image_size, vcfg.image_size)
    with torch.no_grad():
        a = ref(input_ids=ids, pixel_values=px.bfloat16(), use_cache=False).logits
        a32 = ref32(input_ids=ids, pixel_values=px, use_cache=False).logits
        transformers_amd.accelerate(fast)
        assert fast.config.text_config._attn_implementation == "tamd"
        assert fast.config.vision_config._attn_implementation == "tamd"
        dev = env.device
        c = fast(input_ids=ids.to(dev), pixel_values=px.bfloat16().to(dev), use_cache=False).logits
        # integer path: the placeholder positions receive exactly the projected image rows, in order
        emb = fast.get_input_embeddings()(ids.to(dev))
        feats = fast.get_image_features(pixel_values=px.bfloat16().to(dev), return_dict=True).pooler_output
        feats = torch.cat(list(feats), 0) if isinstance(feats, (list, tuple)) else feats
        mask = (ids == 1100).to(dev)
        merged = emb.masked_scatter(mask.unsqueeze(-1).expand_as(emb), feats.to(emb.dtype))
        assert torch.equal(merged[0][mask[0]], feats.reshape(-1, feats.shape[-1]).to(emb.dtype))
    e_fast, e_ref = rel_err(c, a32), rel_err(a, a32)
    assert e_fast <= 1.15 * e_ref + 1e-3, (e_fast, e_ref)
    # with a KV cache (what `generate` does): the multimodal prefill fills the cache, then teacher-forced decode steps -- the
    # language model's cached layer path (fused q|k|v, rotary kernel, Cache.update, M = batch products), nothing falls back
    nxt = torch.randint(0, 1000, (1, 2))
    with torch.no_grad():
        transformers_amd.fallback_calls(reset=True)
        r = ref32(input_ids=ids, pixel_values=px, use_cache=True)
        f = fast(input_ids=ids.to(dev), pixel_values=px.bfloat16().to(dev), use_cache=True)
        assert rel_err(f.logits, r.logits) <= 1.15 * e_ref + 1e-3
        for t in range(2):
            r = ref32(input_ids=nxt[:, t:t + 1], past_key_values=r.past_key_values, use_cache=True)
            f = fast(input_ids=nxt[:, t:t + 1].to(dev), past_key_values=f.past_key_values, use_cache=True)
            assert f.logits.shape == r.logits.shape and rel_err(f.logits, r.logits) <= 1.3 * e_ref + 2e-3, t
        assert transformers_amd.fallback_calls() == {}, transformers_amd.fallback_calls()


def test_bert_decoder_cross_attention_matches_reference(env):
    """a BERT decoder with cross-attention (bert2bert: `BertCrossAttention`, modeling_bert.py) reaches the
    registered attention function with sq != sk and a bidirectional mask over the ENCODER keys.  Output against the
    reference's fp32 eager run, next to its own bf16 run, for a decoder shorter and longer than the encoder."""
    from transformers import BertConfig, BertModel

    torch.manual_seed(31)
    cfg = BertConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=300,
                     max_position_embeddings=64, is_decoder=True, add_cross_attention=True,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")
    ref = BertModel(cfg).bfloat16().eval()
    ref32 = copy.deepcopy(ref).float()
    fast = transformers_amd.accelerate(copy.deepcopy(ref).to(env.device))
    dev = env.device
    for sq, sk in ((5, 12), (1, 12), (20, 12)):
        ids = torch.randint(0, 300, (2, sq))
        enc = torch.randn(2, sk, 128)
        enc_mask = torch.ones(2, sk, dtype=torch.long)
        enc_mask[1, 8:] = 0
        with torch.no_grad():
            a32 = ref32(input_ids=ids, encoder_hidden_states=enc, encoder_attention_mask=enc_mask).last_hidden_state
            a = ref(input_ids=ids, encoder_hidden_states=enc.bfloat16(), encoder_attention_mask=enc_mask).last_hidden_state
            c = fast(input_ids=ids.to(dev), encoder_hidden_states=enc.bfloat16().to(dev),
                     encoder_attention_mask=enc_mask.to(dev)).last_hidden_state
        e_fast, e_ref = rel_err(c, a32), rel_err(a, a32)
        assert e_fast <= 1.1 * e_ref + 1e-3, (sq, sk, e_fast, e_ref)


@pytest.mark.gpu
def test_decoder_stack_replays_as_one_hip_graph():
    """Forward-only calls of a whole Llama stack (transformers_amd/graph_stack.py): the second sighting of a shape is
    captured, later calls replay the graph -- the same bits as the eager path, for changing inputs; calls that want
    per-layer hidden states, gradients or a KV cache stay eager."""
    from transformers_amd import graph_stack

    torch.manual_seed(41)
    dev = torch.device("cuda:0")
    cfg = tiny_llama(True)
    model = transformers_amd.accelerate(LlamaForCausalLM(cfg).bfloat16().to(dev).eval())
    stack = model.model.layers[0].__dict__["_tamd_stack"][0]
    ids = [torch.randint(0, cfg.vocab_size, (2, 300), device=dev) for _ in range(4)]
    old = graph_stack.set_enabled(False)
    try:
        with torch.no_grad():
            want = [model(input_ids=i, use_cache=False).logits.clone() for i in ids]
        graph_stack.set_enabled(True)
        with torch.no_grad():
            got = [model(input_ids=i, use_cache=False).logits.clone() for i in ids]
            assert stack.replays == 3  # first call eager, second captures and replays, then replays
            for g, w in zip(got, want):
                assert torch.equal(g, w)
            # per-layer hidden states: eager, and each layer's own output
            before = stack.replays
            hs = model(input_ids=ids[0], use_cache=False, output_hidden_states=True).hidden_states
            assert stack.replays == before and len(hs) == cfg.num_hidden_layers + 1
            assert not torch.equal(hs[1], hs[2])
            # a padded batch is another signature (first sighting: eager), then its own graph
            am = torch.ones(2, 300, dtype=torch.long, device=dev)
            am[0, :17] = 0
            a = model(input_ids=ids[1], attention_mask=am, use_cache=False).logits.clone()
            b = model(input_ids=ids[1], attention_mask=am, use_cache=False).logits.clone()
            assert stack.replays == before + 1 and torch.equal(a, b)
        # training / gradients: eager
        before = stack.replays
        model.train()
        model(input_ids=ids[0], labels=ids[0], use_cache=False).loss.backward()
        assert stack.replays == before
    finally:
        graph_stack.set_enabled(old)


@pytest.mark.parametrize("p_hidden", [0.0, 0.1])
def test_bert_layer_op_matches_reference_layer(env, p_hidden):
    """torch.ops.tamd.bert_layer (TamdBertLayer: BertLayer.forward as one autograd node, modeling_bert.py:374-416) against
    the reference's own BertLayer in fp32, with a padding mask: output, input gradient and every parameter gradient.
    With hidden dropout the reference layer is restated with the kernels' keep masks rebuilt on the host from the seeds
    (counter-based hash, not torch's Philox stream), as in the post-LN block test above."""
    from transformers import BertConfig
    from transformers.models.bert import modeling_bert as mb

    from transformers_amd import ops
    from transformers_amd.patch import _tables

    torch.manual_seed(71)
    big = env.big
    hd, inter, heads = (768, 3072, 12) if big else (128, 256, 2)
    b, s = (4, 384) if big else (2, 40)
    cfg = BertConfig(hidden_size=hd, intermediate_size=inter, num_attention_heads=heads, hidden_dropout_prob=p_hidden,
                     attention_probs_dropout_prob=0.0, attn_implementation="eager")
    ref = mb.BertLayer(cfg).float().train()
    with torch.no_grad():
        for n, p in ref.named_parameters():  # non-trivial biases and LayerNorm gains
            if p.dim() == 1:
                p.copy_(torch.randn_like(p) * 0.1 + (1.0 if n.endswith("LayerNorm.weight") else 0.0))
    dev = env.device
    fast = copy.deepcopy(ref).bfloat16().to(dev)
    transformers_amd.attention.register()
    fcfg = copy.deepcopy(cfg)
    fcfg._attn_implementation = "tamd"
    for m in fast.modules():
        r = _tables().get(type(m))
        if r is not None:
            m.__class__ = r
        if hasattr(m, "config"):
            m.config = fcfg
    assert type(fast).__name__ == "TamdBertLayer"
    ref = copy.deepcopy(fast).cpu().float()  # the reference holds the SAME (bf16-rounded) weights, in fp32
    for m in ref.modules():
        inv = {v: k for k, v in _tables().items()}
        if type(m) in inv:
            m.__class__ = inv[type(m)]
        if hasattr(m, "config"):
            m.config = cfg
    ref.__dict__.pop("_tamd_stack", None)
    x = torch.randn(b, s, hd).bfloat16()
    valid = torch.ones(b, s, dtype=torch.bool)
    valid[0, s - 9:] = False
    add_mask = torch.zeros(b, 1, 1, s).masked_fill(~valid[:, None, None, :], torch.finfo(torch.float32).min)
    xr = x.float().requires_grad_(True)
    xf = x.to(dev).requires_grad_(True)
    torch.manual_seed(5)
    seeds = (ops.dropout_seed(), ops.dropout_seed()) if p_hidden > 0 else (0, 0)
    torch.manual_seed(5)
    transformers_amd.fallback_calls(reset=True)
    yf = fast(xf, attention_mask=valid.to(dev))
    assert transformers_amd.fallback_calls() == {}
    if p_hidden == 0.0:
        yr = ref(xr, attention_mask=add_mask)
    else:  # the reference layer with the kernels' masks: BertSelfOutput / BertOutput restated (modeling_bert.py:289-293, 347-351)
        keep1 = ops.hidden_dropout_keep_mask(seeds[0], b * s, hd, p_hidden).view(b, s, hd)
        keep2 = ops.hidden_dropout_keep_mask(seeds[1], b * s, hd, p_hidden).view(b, s, hd)
        att = ref.attention
        a, _ = att.self(xr, attention_mask=add_mask)
        h1 = att.output.LayerNorm(att.output.dense(a) * keep1 / (1 - p_hidden) + xr)
        yr = ref.output.LayerNorm(ref.output.dense(ref.intermediate(h1)) * keep2 / (1 - p_hidden) + h1)
    v = valid
    assert rel_err(yf[v.to(dev)], yr[v]) < 6e-3
    g = torch.randn(b, s, hd).bfloat16()
    g[~valid] = 0
    yr.backward(g.float())
    yf.backward(g.to(dev))
    assert rel_err(xf.grad[v.to(dev)], xr.grad[v]) < 1.2e-2
    gr = dict(ref.named_parameters())
    for n, p in fast.named_parameters():
        if "key.bias" in n:  # exactly zero (softmax shift invariance)
            continue
        assert p.grad is not None, n
        assert rel_err(p.grad, gr[n].grad) < 1.5e-2, n


@pytest.mark.gpu
def test_captured_training_step_draws_fresh_dropout_masks():
    """VERDICT r3 "graph-safe dropout" (modeling_bert.py:131, 289-293; modeling_llama.py:209): a BERT training step with the
    shipped dropout 0.1 / 0.1 captured in ONE HIP graph.  While the stream is capturing the dropout seeds are device words
    written by torch's own (graph-safe) RNG kernel, so every replay draws new masks -- a host-drawn seed would be baked into
    the graph and replay one mask for ever -- and reseeding the generator repeats a replay bit for bit."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from transformers import BertConfig, BertForMaskedLM
    from transformers_amd import ops

    old = ops._set_backend(None)
    try:
        dev = torch.device("cuda:0")
        torch.manual_seed(21)
        cfg = BertConfig(vocab_size=1000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                         max_position_embeddings=128, attn_implementation="eager")
        model = transformers_amd.accelerate(BertForMaskedLM(cfg).bfloat16().to(dev)).train()
        ids = torch.randint(1, 1000, (4, 128), device=dev)
        params = [p for p in model.parameters()]

        def step():
            out = model(input_ids=ids, labels=ids)
            out.loss.backward()
            return out.loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream (allocator, autograd)
            for _ in range(2):
                step()
                model.zero_grad(set_to_none=True)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = step()
        grads = [p.grad for p in params if p.grad is not None]

        def replay(seed):
            torch.manual_seed(seed)
            for g in grads:
                g.zero_()  # (the captured backward accumulates into the static .grad tensors)
            graph.replay()
            torch.cuda.synchronize()
            return static_loss.item(), [g.clone() for g in grads]

        l1, g1 = replay(5)
        l2, g2 = replay(5)
        l3, g3 = replay(6)
        assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2))      # same generator state: the same masks
        assert l1 != l3 and any(not torch.equal(a, b) for a, b in zip(g1, g3))  # another state: other masks
        torch.manual_seed(5)
        for g in grads:
            g.zero_()
        graph.replay()
        graph.replay()  # consecutive replays advance the generator: the second differs from the first
        torch.cuda.synchronize()
        assert static_loss.item() != l1
        assert all(torch.isfinite(g).all() for g in grads)
    finally:
        ops._set_backend(old)
