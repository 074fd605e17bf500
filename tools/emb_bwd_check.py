"""embedding backward on degenerate id distributions (one id for every token = BERT token_type; random ids)."""
import sys, json, torch
sys.path.insert(0, ".")
from transformers_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
dout = torch.randn(32, 512, 768, device=dev).bfloat16()
for name, ids, vocab in (("all tokens one id (token_type)", torch.zeros(32, 512, dtype=torch.long, device=dev), 2),
                         ("positions 0..511 x 32", torch.arange(512, device=dev).repeat(32, 1), 512),
                         ("random ids, vocab 30522", torch.randint(0, 30522, (32, 512), device=dev), 30522)):
    t = timeit(lambda: ops.raw_embedding_bwd(ids, dout, vocab))
    ref = torch.zeros(vocab, 768, dtype=torch.float32, device=dev).index_add_(0, ids.view(-1), dout.view(-1, 768).float())
    got = ops.raw_embedding_bwd(ids, dout, vocab)
    err = ((got.float() - ref).norm() / ref.norm()).item()
    print(json.dumps({"case": name, "ms": round(t, 3), "rel_err": round(err, 5)}), flush=True)
