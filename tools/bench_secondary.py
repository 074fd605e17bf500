"""Secondary configurations of BASELINE.json (parity-test cases, not the bench line): model-level throughput of the
accelerated path next to the reference's own GPU paths (eager / sdpa) on the same MI355X, random-init weights.
  BB  bert-base-uncased MLM fwd+bwd, batch 32 x seq 512 (hidden/attention dropout 0.1 as shipped, train mode)
  LV  LLaVA-1.5-7B-shaped forward: CLIP ViT-L/14-336 tower (hidden_states[-2]) + Llama-2-7B-shaped LLM, 1088 positions
"""
import copy, json, sys, time
import torch
sys.path.insert(0, ".")
import transformers_amd
from transformers import BertConfig, BertForMaskedLM, CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM
dev = torch.device("cuda:0")

def timeit(fn, iters=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

# ---- BB
torch.manual_seed(0)
cfg = BertConfig()  # bert-base-uncased
b, s = 32, 512
ids = torch.randint(1000, 30000, (b, s), device=dev)
am = torch.ones(b, s, dtype=torch.long, device=dev); am[::4, 400:] = 0
labels = ids.clone(); labels[:, ::2] = -100
res = {"config": "bert-base-uncased MLM fwd+bwd, batch 32 x 512, bf16, dropout 0.1 (train mode)"}
for impl in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("eager", "sdpa", "tamd")):
    c = copy.deepcopy(cfg); c._attn_implementation = "eager" if impl == "tamd" else impl
    m = BertForMaskedLM(c).bfloat16().to(dev).train()
    if impl == "tamd": transformers_amd.accelerate(m)
    def step():
        m.zero_grad(set_to_none=True)
        m(input_ids=ids, attention_mask=am, labels=labels).loss.backward()
    t = timeit(step)
    res[impl] = {"ms": round(t * 1e3, 2), "tokens_per_s": round(b * s / t)}
    del m
print(json.dumps(res), flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "bb":
    sys.exit(0)
# ---- LV
torch.manual_seed(1)
vc = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336, patch_size=14)
lc = LlamaConfig(vocab_size=32064, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=32, max_position_embeddings=4096)
px = torch.randn(1, 3, 336, 336, device=dev).bfloat16()
emb = torch.randn(1, 1088, 4096, device=dev).bfloat16()
res = {"config": "LLaVA-1.5-7B shape forward: CLIP-L/336 tower (1 image, hidden_states[-2]) + Llama-2-7B LLM on 1088 positions, bf16"}
for impl in ("eager", "sdpa", "tamd"):
    v = copy.deepcopy(vc); l = copy.deepcopy(lc)
    v._attn_implementation = l._attn_implementation = "eager" if impl == "tamd" else impl
    tower = CLIPVisionModel(v).bfloat16().to(dev).eval()
    llm = LlamaForCausalLM(l).bfloat16().to(dev).eval()
    if impl == "tamd":
        transformers_amd.accelerate(tower); transformers_amd.accelerate(llm)
    with torch.no_grad():
        tt = timeit(lambda: tower(pixel_values=px, output_hidden_states=True).hidden_states[-2])
        tl = timeit(lambda: llm(inputs_embeds=emb, use_cache=False).logits)
    res[impl] = {"tower_ms": round(tt * 1e3, 2), "llm_ms": round(tl * 1e3, 2)}
    del tower, llm
print(json.dumps(res), flush=True)
