"""Search over placements of the three-barrier GEMM loop (csrc/gemm.hip `kfine_*`): where the 32 fragment reads, 16 LDS-DMA pieces, 3
barriers and their waits of a 64-deep stage sit among its 128 MFMAs.  A table is DATA here: `gen` turns a parameter set into the two
`if constexpr` chains (force-included as TAMD_B3_K0_ACTIONS / TAMD_B3_K1_ACTIONS), checks the hazards the structure depends on, and
`build` compiles ONE instantiation (forward layout, plain epilogue) with a tiny launcher into tools/ab/b3/libb3_<id>.so (git-ignored; the
files travel to the GPU box with the tree); `run` times every variant against variant 0 (the vendor table = the product) on the
Llama-3-8B forward shapes, interleaved, and compares the outputs bit for bit.

    python tools/b3_search.py build [--seed S] [--count N]      # CPU: hipcc cross-compiles
    python tools/b3_search.py run > gpurun_out/<tag>_b3_search.jsonl   # GPU
"""
import argparse
import ctypes
import json
import random
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tools" / "ab" / "b3"
sys.path.insert(0, str(ROOT))

VENDOR = dict(
    xr=[1, 3, 5, 7, 9, 11, 13, 15], bar1=22, ap=[23, 26, 29, 32, 35, 53, 56, 59], yr=[25, 28, 31, 34, 37, 39, 41, 43], bar2=52,
    bp0=[62],                                   # B pieces issued in k-step 0 (behind barrier 2)
    bp1=[1, 22, 24, 26, 33, 37, 61], bar3=29,   # ... in k-step 1; barrier 3 (landed data)
    xr1=[30, 31, 32, 34, 35, 39, 40, 41], yr1=[42, 43, 46, 49, 51, 54, 57, 60])


def check(t):
    """The hazards of the structure (profiles/r06_hipblaslt_loop.md): returns None or the reason a table is invalid."""
    k0 = {}
    k1 = {}

    def put(d, pos, what):
        if not 1 <= pos <= 64:
            return f"{what} at {pos}"
        if pos in d:
            return f"{what} and {d[pos]} both behind MFMA {pos}"
        d[pos] = what
        return None

    for j, p in enumerate(t["xr"]):
        e = put(k0, p, f"xr{j}")
        if e:
            return e
    for j, p in enumerate(t["yr"]):
        e = put(k0, p, f"yr{j}")
        if e:
            return e
    for j, p in enumerate(t["ap"]):
        e = put(k0, p, f"ap{j}")
        if e:
            return e
    for j, p in enumerate(t["bp0"]):
        e = put(k0, p, f"bp{j}")
        if e:
            return e
    for name in ("bar1", "bar2"):
        for d in (0, 1):
            e = put(k0, t[name] - d, name + ("" if d == 0 else "_wait"))
            if e:
                return e
    for j, p in enumerate(t["bp1"]):
        e = put(k1, p, f"bp{len(t['bp0']) + j}")
        if e:
            return e
    for d in (0, 1):
        e = put(k1, t["bar3"] - d, "bar3" + ("" if d == 0 else "_wait"))
        if e:
            return e
    for j, p in enumerate(t["xr1"]):
        e = put(k1, p, f"xr1_{j}")
        if e:
            return e
    for j, p in enumerate(t["yr1"]):
        e = put(k1, p, f"yr1_{j}")
        if e:
            return e
    if len(t["bp0"]) + len(t["bp1"]) != 8 or len(t["ap"]) != 8:
        return "piece count"
    if sorted(t["ap"]) != t["ap"] or sorted(t["bp0"]) != t["bp0"] or sorted(t["bp1"]) != t["bp1"]:
        return "pieces out of order (the operand base steps behind piece 7)"
    if max(t["xr"]) >= t["bar1"] - 1:
        return "an A fragment of k-step 1 is read behind the wait of barrier 1"
    if min(t["ap"]) <= t["bar1"]:
        return "an A piece in front of barrier 1 (write after read on A_s)"
    if max(t["yr"]) >= t["bar2"] - 1 or t["bar2"] <= t["bar1"] + 1:
        return "a B fragment of k-step 1 is read behind the wait of barrier 2"
    if t["bp0"] and min(t["bp0"]) <= t["bar2"]:
        return "a B piece in front of barrier 2 (write after read on B_s)"
    if min(t["xr1"] + t["yr1"]) <= t["bar3"]:
        return "a fragment of stage s+1 is read in front of barrier 3"
    # fragment j of stage s+1 overwrites register buffer 0 while k-step 1 runs on buffer 1: no constraint; the stage opens with lgkmcnt(0)
    return None


def gen(t):
    err = check(t)
    if err:
        raise ValueError(err)
    k0, k1 = [], []
    for j, p in enumerate(t["xr"]):
        k0.append((p, f"fx[1][{j}] = frag_a4(sa, 1, {j});"))
    k0.append((t["bar1"] - 1, "wait_lgkmcnt0();"))
    k0.append((t["bar1"], "raw_barrier();"))
    for j, p in enumerate(t["ap"]):
        k0.append((p, f"issue({j}, sa);"))
    for j, p in enumerate(t["yr"]):
        k0.append((p, f"fw[1][{j}] = frag_b4(sb, 1, {j});"))
    k0.append((t["bar2"] - 1, "wait_lgkmcnt0();"))
    k0.append((t["bar2"], "raw_barrier();"))
    nb = 0
    for p in t["bp0"]:
        k0.append((p, f"issue({8 + nb}, sb);"))
        nb += 1
    before = 0
    for p in t["bp1"]:
        k1.append((p, f"issue({8 + nb}, sb);"))
        nb += 1
        if p < t["bar3"] - 1:
            before += 1
    inflight = 8 + len(t["bp0"]) + before  # this stage's pieces issued when the landed-data wait comes: everything older has landed
    k1.append((t["bar3"] - 1, f"wait_vmcnt<{inflight}>();"))
    k1.append((t["bar3"], "raw_barrier();"))
    for j, p in enumerate(t["xr1"]):
        k1.append((p, f"fx[0][{j}] = frag_a4(na, 0, {j});"))
    for j, p in enumerate(t["yr1"]):
        k1.append((p, f"fw[0][{j}] = frag_b4(nbs, 0, {j});"))

    def chain(acts):
        return " \\\n".join(f"  if constexpr (i == {p}) {{ {code} }}" for p, code in sorted(acts))

    return ("#define TAMD_B3_TABLE 1\n#define TAMD_B3_K0_ACTIONS \\\n" + chain(k0) + "\n#define TAMD_B3_K1_ACTIONS \\\n" + chain(k1) + "\n")


def mutate(t, rng):
    """One random, structure-preserving change of a table; returns a valid table or None."""
    t = {k: (list(v) if isinstance(v, list) else v) for k, v in t.items()}
    kind = rng.choice(["shift_ap", "shift_bar3", "shift_bp1", "shift_reads1", "shift_bar12", "shift_yr", "move_one", "bp_split"])
    d = rng.choice([-4, -3, -2, -1, 1, 2, 3, 4])
    if kind == "shift_ap":
        lo = rng.choice([0, 5])
        for j in range(lo, 8 if lo else 5):
            t["ap"][j] += d
    elif kind == "shift_bar3":
        t["bar3"] += d
        t["xr1"] = [p + d for p in t["xr1"]]
    elif kind == "shift_bp1":
        j = rng.randrange(len(t["bp1"]))
        t["bp1"][j] += d
    elif kind == "shift_reads1":
        t["yr1"] = [p + d for p in t["yr1"]]
    elif kind == "shift_bar12":
        which = rng.choice(["bar1", "bar2"])
        t[which] += d
    elif kind == "shift_yr":
        t["yr"] = [p + d for p in t["yr"]]
    elif kind == "move_one":
        key = rng.choice(["xr", "yr", "ap", "xr1", "yr1"])
        j = rng.randrange(8)
        t[key][j] += d
    elif kind == "bp_split":
        if t["bp1"] and rng.random() < 0.5:  # one more B piece in k-step 0
            t["bp0"] = t["bp0"] + [min(64, (t["bp0"][-1] if t["bp0"] else t["bar2"]) + 2)]
            t["bp1"] = t["bp1"][1:] if len(t["bp1"]) > 1 else t["bp1"]
            if len(t["bp0"]) + len(t["bp1"]) != 8:
                return None
        else:
            return None
    return t if check(t) is None else None


SRC = """#define TAMD_GEMM_KERNELS_ONLY 1
#include "{gemm}"
using namespace tamd;
extern "C" int b3_run(const void* A, const void* B, void* C, long long M, long long N, long long K, void* stream) {{
  GemmArgs g{{}};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
  g.tiles_m = (int)((M + 255) / 256); g.tiles_n = (int)((N + 255) / 256);
  g.splits = 1; g.col_scale = 1.f;
  hipLaunchKernelGGL((gemm_fl_kernel<bf16_t, false, false, TAMD_EPI_NONE, TAMD_ACT_NONE, 1024>), dim3(g.tiles_m * g.tiles_n), dim3(256),
                     (size_t)kXSmem, (hipStream_t)stream, g);
  return (int)hipGetLastError();
}}
"""


def build_one(idx, table):
    from transformers_amd import build as tb

    d = OUT / f"v{idx}"
    d.mkdir(parents=True, exist_ok=True)
    (d / "table.h").write_text(gen(table))
    (d / "table.json").write_text(json.dumps(table))
    (d / "k.hip").write_text(SRC.format(gemm=str(tb.CSRC / "gemm.hip")))
    lib = OUT / f"libb3_{idx}.so"
    cmd = [tb._hipcc(), *tb.FLAGS, "-include", str(d / "table.h"), "-shared", str(d / "k.hip"), "-o", str(lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return idx, r.stderr[-2000:]
    return idx, None


def cmd_build(args):
    rng = random.Random(args.seed)
    tables = [VENDOR]
    seen = {json.dumps(VENDOR, sort_keys=True)}
    bases = [VENDOR] + [json.loads(b) for b in args.base]
    tries = 0
    while len(tables) < args.count and tries < 100000:
        tries += 1
        t = rng.choice(bases)
        for _ in range(rng.choice([1, 1, 2, 3])):
            t2 = mutate(t, rng)
            if t2 is None:
                break
            t = t2
        else:
            key = json.dumps(t, sort_keys=True)
            if key not in seen:
                seen.add(key)
                tables.append(t)
    for b in bases[1:]:
        if json.dumps(b, sort_keys=True) not in seen:
            tables.append(b)
    if OUT.exists():
        for f in OUT.glob("libb3_*.so"):
            f.unlink()
    with ThreadPoolExecutor(args.jobs) as ex:
        for idx, err in ex.map(lambda it: build_one(*it), enumerate(tables)):
            print(idx, "ok" if err is None else "FAILED: " + err, flush=True)


def cmd_run(args):
    import torch

    dev = torch.device("cuda:0")
    libs = sorted(OUT.glob("libb3_*.so"), key=lambda p: int(p.stem.split("_")[1]))
    shapes = {"gate_up": (32768, 28672, 4096), "o_proj": (32768, 4096, 4096), "down": (32768, 4096, 14336), "qkv": (32768, 6144, 4096)}
    fns = {}
    for p in libs:
        dll = ctypes.CDLL(str(p))
        dll.b3_run.restype = ctypes.c_int
        dll.b3_run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_longlong] * 3 + [ctypes.c_void_p]
        fns[int(p.stem.split("_")[1])] = dll.b3_run
    res = {i: {} for i in fns}
    for name in args.shapes.split(","):
        m, n, k = shapes[name]
        torch.manual_seed(0)
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        st = torch.cuda.current_stream().cuda_stream
        ref = None
        for i, fn in fns.items():
            out.zero_()
            assert fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), m, n, k, st) == 0
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            res[i].setdefault("same_bits", True)
            res[i]["same_bits"] = res[i]["same_bits"] and bool(torch.equal(out, ref))
        for rnd in range(args.rounds):
            for i, fn in fns.items():
                for _ in range(2):
                    fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), m, n, k, st)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), m, n, k, st)
                e1.record()
                torch.cuda.synchronize()
                res[i].setdefault(name, []).append(round(2.0 * m * n * k / (e0.elapsed_time(e1) / args.iters) / 1e9))
        del x, w, out
    base = res[0]
    for i in sorted(res):
        r = res[i]
        med = {s: sorted(r[s])[len(r[s]) // 2] for s in args.shapes.split(",")}
        rel = {s: round(med[s] / sorted(base[s])[len(base[s]) // 2] - 1, 4) for s in med}
        table = json.loads((OUT / f"v{i}" / "table.json").read_text())
        print(json.dumps({"variant": i, "same_bits": r["same_bits"], "median_TF": med, "vs_vendor": rel,
                          "mean_vs_vendor": round(sum(rel.values()) / len(rel), 4), "table": table}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build")
    b.add_argument("--seed", type=int, default=0)
    b.add_argument("--count", type=int, default=32)
    b.add_argument("--jobs", type=int, default=8)
    b.add_argument("--base", action="append", default=[], help="JSON of a table to mutate from (besides the vendor's); repeatable")
    r = sub.add_parser("run")
    r.add_argument("--rounds", type=int, default=3)
    r.add_argument("--iters", type=int, default=8)
    r.add_argument("--shapes", default="gate_up,o_proj,down")
    a = ap.parse_args()
    {"build": cmd_build, "run": cmd_run}[a.cmd](a)
