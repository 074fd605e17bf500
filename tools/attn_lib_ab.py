"""A/B of the attention entry points of TWO builds of the C-ABI library in one process, interleaved on one GPU:
    python tools/attn_lib_ab.py [base.so] [new.so] [more.so ...]   (defaults: tools/ab/libtamd_base.so, transformers_amd/libtamd.so)
Both libraries are driven through ctypes (include/tamd.h: tamd_attn_fwd / tamd_attn_bwd) on the same tensors; per shape it
prints forward / backward time and TFLOP/s of each and the error of each against an fp32 eager restatement on a slice
(forward 4*B*H*Sq*Sk*D flops, x0.5 causal; backward 2.5x).  The base library is a build of an earlier commit kept as a
git-ignored .so (it travels with the tree to the GPU box): no second code arm in the sources."""
import ctypes
import json
import math
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from transformers_amd._cabi import AttnBwdParams, AttnParams, TamdLib, TAMD_BF16  # noqa: E402


# parameter blocks as an ABI-6 build lays them out (tamd_attn_params grew by q_prescaled in ABI 7, which moves every field
# of tamd_attn_bwd_params behind the embedded block)
class AttnParamsV6(ctypes.Structure):
    _fields_ = [f for f in AttnParams._fields_ if f[0] not in ("q_prescaled", "dropout_seed_dev")]


class AttnBwdParamsV6(ctypes.Structure):
    _fields_ = [("fwd", AttnParamsV6)] + [f for f in AttnBwdParams._fields_ if f[0] != "fwd"]


KEY_VALID = None  # AB_MASK=1: a [batch, seq_k] padding mask (the last 5 + 7 b keys of row b invalid): the HAS_MASK instantiations
DROPOUT = float(os.environ.get("AB_DROPOUT", "0"))  # attention dropout probability (timing only: the builds' masks differ)


def params(q, k, v, o, lse, scale, causal, p):
    p.q, p.k, p.v, p.o, p.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr()
    p.key_valid = KEY_VALID.data_ptr() if KEY_VALID is not None else None
    p.batch, p.seq_q, p.heads_q, p.head_dim = q.shape
    p.seq_k, p.heads_kv = k.shape[1], k.shape[2]
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        setattr(p, f"{name}_stride_b", t.stride(0))
        setattr(p, f"{name}_stride_s", t.stride(1))
        setattr(p, f"{name}_stride_h", t.stride(2))
    p.scale, p.causal, p.dtype, p.dropout_p, p.dropout_seed, p.q_start = scale, int(causal), TAMD_BF16, DROPOUT, 1234567, None
    return p


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def reference(q, k, v, do, scale, causal, nb=1, nh=2):
    """fp32 eager attention and its gradients on batch rows < nb and the first nh query heads' kv groups"""
    g = q.shape[2] // k.shape[2]
    hk = max(1, nh // g)
    hq = hk * g
    qr, kr, vr = (t[:nb, :, :h].detach().float().requires_grad_(True) for t, h in ((q, hq), (k, hk), (v, hk)))
    qf = qr.permute(0, 2, 1, 3)
    kf = kr.permute(0, 2, 1, 3).repeat_interleave(g, 1)
    vf = vr.permute(0, 2, 1, 3).repeat_interleave(g, 1)
    sc = qf @ kf.transpose(-1, -2) * scale
    if causal:
        sq, sk = sc.shape[-2:]
        sc = sc.masked_fill(~torch.tril(torch.ones(sq, sk, dtype=torch.bool, device=q.device), diagonal=sk - sq), float("-inf"))
    out = (torch.softmax(sc, -1) @ vf).permute(0, 2, 1, 3)
    out.backward(do[:nb, :, :hq].float())
    return out.detach(), qr.grad, kr.grad, vr.grad, nb, hq, hk


def main():
    base = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "tools" / "ab" / "libtamd_base.so"
    new = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "transformers_amd" / "libtamd.so"
    libs = {"base": TamdLib(base, accept_abi=(6, 7, 8, 9)), "new": TamdLib(new)}  # (the struct grew at its end: ABI 6 reads a prefix)
    for extra in sys.argv[3:]:  # further builds of the current ABI (tools/build_variant.py), tagged by their file name
        libs[Path(extra).stem.replace("libtamd_", "")] = TamdLib(Path(extra))
    if os.environ.get("AB_LIBS"):  # e.g. AB_LIBS=new under rocprofv3: kernel names of one build only
        libs = {k: v for k, v in libs.items() if k in os.environ["AB_LIBS"].split(",")}
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [("llama3-8b", 8, 4096, 32, 8, 128, True), ("llama3-8b-2x8192", 2, 8192, 32, 8, 128, True),
              ("bidir-128", 8, 4096, 32, 8, 128, False), ("bert-base", 32, 512, 12, 12, 64, False),
              ("clip-l", 16, 577, 16, 16, 64, False), ("llava-lm", 1, 1088, 32, 32, 128, True)]
    if os.environ.get("AB_SHAPES"):
        shapes = [x for x in shapes if x[0] in os.environ["AB_SHAPES"].split(",")]
    for name, b, s, hq, hkv, d, causal in shapes:
        torch.manual_seed(0)
        q = torch.randn(b, s, hq, d, device=dev).bfloat16()
        k = torch.randn(b, s, hkv, d, device=dev).bfloat16()
        v = torch.randn(b, s, hkv, d, device=dev).bfloat16()
        do = torch.randn(b, s, hq, d, device=dev).bfloat16()
        scale = 1 / math.sqrt(d)
        global KEY_VALID
        KEY_VALID = None
        if os.environ.get("AB_MASK"):
            KEY_VALID = torch.ones(b, s, dtype=torch.uint8, device=dev)
            for i in range(b):
                KEY_VALID[i, s - 5 - 7 * i:] = 0
        fl = 4.0 * b * hq * s * s * d * (0.5 if causal else 1.0)
        ref = reference(q, k, v, do, scale, causal) if s <= 4096 and DROPOUT == 0 and KEY_VALID is None and not os.environ.get("AB_NOREF") else None
        row = {"shape": name, "dropout_p": DROPOUT}
        runs = {}
        for tag, lib in libs.items():
            o = torch.empty_like(q)
            lse = torch.empty(b, hq, s, device=dev, dtype=torch.float32)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            delta = torch.empty(2, b, hq, s, device=dev, dtype=torch.float32)
            v6 = lib.tamd_abi_version() < 7
            bp = AttnBwdParamsV6() if v6 else AttnBwdParams()
            fp = params(q, k, v, o, lse, scale, causal, bp.fwd)
            for fn, st in ((lib.tamd_attn_fwd, type(fp)), (lib.tamd_attn_bwd, type(bp))):
                fn.argtypes = [ctypes.POINTER(st), ctypes.c_void_p]
            bp.dout, bp.dq, bp.dk, bp.dv, bp.delta = do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
            bp.rope_cos, bp.rope_sin, bp.rope_cos_batch = None, None, 1
            fwd = lambda lib=lib, fp=fp: lib.check(lib.tamd_attn_fwd(ctypes.byref(fp), stream), "fwd")  # noqa: E731
            bwd = lambda lib=lib, bp=bp: lib.check(lib.tamd_attn_bwd(ctypes.byref(bp), stream), "bwd")  # noqa: E731
            fwd()
            bwd()
            torch.cuda.synchronize()
            if ref is not None:
                out, gq, gk, gv, nb, nq, nk = ref
                row[f"{tag}_err"] = {"o": round(rel(o[:nb, :, :nq], out), 5), "dq": round(rel(dq[:nb, :, :nq], gq), 5),
                                     "dk": round(rel(dk[:nb, :, :nk], gk), 5),
                                     "dv": round(rel(dv[:nb, :, :nk], gv), 5)}
            runs[tag] = (fwd, bwd, (o, lse, dq, dk, dv, delta, fp, bp))
            if tag != "base" and "base" in runs:  # the same bits as the base build?
                bo, _, bdq, bdk, bdv = runs["base"][2][:5]
                row[f"{tag}_same_bits"] = {"o": bool(torch.equal(o, bo)), "dq": bool(torch.equal(dq, bdq)),
                                           "dk": bool(torch.equal(dk, bdk)), "dv": bool(torch.equal(dv, bdv))}
        tf = {t: [] for t in runs}
        tb = {t: [] for t in runs}
        for _ in range(3):  # interleaved rounds
            for tag, (fwd, bwd, _) in runs.items():
                tf[tag].append(timeit(fwd))
                tb[tag].append(timeit(bwd, iters=10))
        for tag in runs:
            f, bw = min(tf[tag]), min(tb[tag])
            row[tag] = {"fwd_ms": round(f, 4), "fwd_TF": round(fl / f / 1e9), "bwd_ms": round(bw, 4),
                        "bwd_TF": round(2.5 * fl / bw / 1e9), "fwd_all": [round(x, 4) for x in tf[tag]],
                        "bwd_all": [round(x, 4) for x in tb[tag]]}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
