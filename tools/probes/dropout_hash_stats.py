"""Statistics of the attention-dropout block mix (csrc/dropout.h attn_block_words, round 4: 24-bit multiplies) against the round-3
mix: avalanche, keep rate, row / column variance, neighbour / head / seed correlations, chi-square of the 16-bit fields, and the
effect of mixing the seed with splitmix64 first (without it, seeds that differ in the high word give correlated masks).
    python tools/probes/dropout_hash_stats.py        (numpy only, ~1 minute)"""
import numpy as np
M32 = np.uint64(0xFFFFFFFF)
def u32(x): return (x & M32).astype(np.uint64)
def mul24(a, b): return u32((a & np.uint64(0xFFFFFF)) * (np.uint64(b) & np.uint64(0xFFFFFF)))
def h_new(x, y, K=(0x9E3779, 0x85EBCB, 0xC2B2AF, 0x27D4EB, 0x165667B1, 0xD3A265, 0x7F4A7D)):
    x = u32(x); y = u32(y)
    x = u32(mul24(x, K[0]) + mul24(x >> np.uint64(8), K[1]) + y)
    x ^= x >> np.uint64(15)
    x = u32(mul24(x, K[2]) + mul24(x >> np.uint64(8), K[3]))
    x ^= x >> np.uint64(13)
    w0 = x
    t = x ^ np.uint64(K[4])
    w1 = u32(mul24(t, K[5]) + mul24(t >> np.uint64(8), K[6]))
    w1 ^= w1 >> np.uint64(16)
    return w0, w1
def h_old(x, y):
    x = u32(x); y = u32(y)
    x = u32(x * np.uint64(0x9E3779B1)); x ^= x >> np.uint64(15)
    x = u32(x + y); x = u32(x * np.uint64(0xC2B2AE3D)); x ^= x >> np.uint64(13)
    x = u32(x * np.uint64(0x27D4EB2F)); x ^= x >> np.uint64(16)
    w0 = x
    t = u32((x ^ np.uint64(0x85EBCA77)) * np.uint64(0x9E3779B1)); t ^= t >> np.uint64(15); t = u32(t * np.uint64(0xC2B2AE3D)); t ^= t >> np.uint64(16)
    return w0, t
def avalanche(h, n=200000, seed=1):
    rng = np.random.default_rng(seed)
    # realistic inputs: sequential block indices xor a random seed
    x = (np.arange(n, dtype=np.uint64) ^ np.uint64(0x1234ABCD)); y = np.full(n, 0x0F1E2D3C, dtype=np.uint64)
    w0, w1 = h(x, y)
    worst = 0; res = []
    for bit in range(32):
        a0, a1 = h(x ^ np.uint64(1 << bit), y)
        d = np.concatenate([((w0 ^ a0)[:, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1), ((w1 ^ a1)[:, None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)], 1)
        pr = d.mean(0)
        res.append((bit, pr.min(), pr.max()))
    return res
def stats(h, p=0.1, S=512, bh=8, seed=0x1234567890ABC):
    thr = int(p * 65536)
    csq = csk = S // 2
    b = np.arange(bh, dtype=np.uint64)[:, None, None]; qb = np.arange(csq, dtype=np.uint64)[None, :, None]; kb = np.arange(csk, dtype=np.uint64)[None, None, :]
    idx = (b * np.uint64(csq) + qb) * np.uint64(csk) + kb
    w0, w1 = h((idx & M32) ^ np.uint64(seed & 0xFFFFFFFF), (idx >> np.uint64(32)) ^ np.uint64(seed >> 32))
    f = np.stack([w0 & np.uint64(0xFFFF), w1 & np.uint64(0xFFFF), w0 >> np.uint64(16), w1 >> np.uint64(16)], -1)  # (qeven,keven),(qeven,kodd),(qodd,keven),(qodd,kodd)
    keep = (f >= thr)
    full = np.zeros((bh, S, S), bool)
    full[:, 0::2, 0::2] = keep[..., 0]; full[:, 0::2, 1::2] = keep[..., 1]; full[:, 1::2, 0::2] = keep[..., 2]; full[:, 1::2, 1::2] = keep[..., 3]
    m = full.astype(np.float64)
    out = {"mean": m.mean(), "row_var_ratio": m.sum(-1).var() / (S * p * (1 - p)), "col_var_ratio": m.sum(-2).var() / (S * p * (1 - p))}
    c = m - m.mean()
    for name, (dq, dk) in {"k+1": (0, 1), "q+1": (1, 0), "k+2": (0, 2), "q+2": (2, 0), "diag": (1, 1), "k+64": (0, 64), "q+32": (32, 0)}.items():
        out["corr " + name] = (c[:, :S - dq, :S - dk] * c[:, dq:, dk:]).mean() / c.var()
    # across heads and across seeds
    out["corr bh+1"] = (c[:-1] * c[1:]).mean() / c.var()
    return out, full
for name, h in (("old", h_old), ("new", h_new)):
    av = avalanche(h)
    print(name, "avalanche min/max flip prob over input bits:", min(a[1] for a in av), max(a[2] for a in av))
    print("   worst per-bit:", [(a[0], round(a[1], 3), round(a[2], 3)) for a in av if a[1] < 0.4 or a[2] > 0.6][:10])
    st, full = stats(h)
    print("  ", {k: round(v, 4) for k, v in st.items()})
    st2, full2 = stats(h, seed=0x1234567890ABD)
    c1 = full.astype(float) - full.mean(); c2 = full2.astype(float) - full2.mean()
    print("   corr seed+1:", (c1 * c2).mean() / c1.var())

print("---- more tests of the new hash")
def fields(h, seed, S=1024, bh=4):
    csq = csk = S // 2
    b = np.arange(bh, dtype=np.uint64)[:, None, None]; qb = np.arange(csq, dtype=np.uint64)[None, :, None]; kb = np.arange(csk, dtype=np.uint64)[None, None, :]
    idx = (b * np.uint64(csq) + qb) * np.uint64(csk) + kb
    return h((idx & M32) ^ np.uint64(seed & 0xFFFFFFFF), (idx >> np.uint64(32)) ^ np.uint64(seed >> 32))
for name, h in (("old", h_old), ("new", h_new)):
    base = 0x2B3C4D5E6F708192 & ((1 << 62) - 1)
    w0, w1 = fields(h, base)
    f = np.concatenate([(w0 & np.uint64(0xFFFF)).ravel(), (w0 >> np.uint64(16)).ravel(), (w1 & np.uint64(0xFFFF)).ravel(), (w1 >> np.uint64(16)).ravel()])
    hist = np.bincount((f >> np.uint64(8)).astype(np.int64), minlength=256)
    exp = f.size / 256
    chi = ((hist - exp) ** 2 / exp).sum()
    print(name, "chi2(255 dof) of the top byte of the fields:", round(chi, 1), " mean keep at p=0.1:", (f >= 6553).mean())
    k0 = (f >= 6553).astype(float); c0 = k0 - k0.mean()
    for label, s2 in (("lo+1", base + 1), ("lo^0x100", base ^ 0x100), ("hi+1", base + (1 << 32)), ("hi^bit20", base ^ (1 << 52)), ("random", 0x1122334455667788 & ((1 << 62) - 1))):
        a0, a1 = fields(h, s2)
        g = np.concatenate([(a0 & np.uint64(0xFFFF)).ravel(), (a0 >> np.uint64(16)).ravel(), (a1 & np.uint64(0xFFFF)).ravel(), (a1 >> np.uint64(16)).ravel()])
        k1 = (g >= 6553).astype(float); c1 = k1 - k1.mean()
        print("   corr with seed", label, round((c0 * c1).mean() / c0.var(), 5), " (sigma %.5f)" % (1 / np.sqrt(c0.size)))
    # the four fields of one block against each other
    fs = [(w0 & np.uint64(0xFFFF)), (w0 >> np.uint64(16)), (w1 & np.uint64(0xFFFF)), (w1 >> np.uint64(16))]
    ks = [((x >= 6553).astype(float) - 0.9).ravel() for x in fs]
    print("   within-block corr:", [round((ks[i] * ks[j]).mean() / 0.09, 5) for i in range(4) for j in range(i + 1, 4)])

print("---- with the seed pre-mixed by splitmix64 on the host")
def splitmix(seed):
    z = seed & ((1 << 64) - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
    return z ^ (z >> 31)
def fields2(seed, S=1024, bh=4):
    z = splitmix(seed)
    csq = csk = S // 2
    b = np.arange(bh, dtype=np.uint64)[:, None, None]; qb = np.arange(csq, dtype=np.uint64)[None, :, None]; kb = np.arange(csk, dtype=np.uint64)[None, None, :]
    idx = (b * np.uint64(csq) + qb) * np.uint64(csk) + kb
    return h_new((idx & M32) ^ np.uint64(z & 0xFFFFFFFF), u32(np.uint64(z >> 32) + mul24(idx >> np.uint64(32), 0x632BE5)))
base = 0x2B3C4D5E6F708192 & ((1 << 62) - 1)
w0, w1 = fields2(base)
f = np.concatenate([(w0 & np.uint64(0xFFFF)).ravel(), (w0 >> np.uint64(16)).ravel(), (w1 & np.uint64(0xFFFF)).ravel(), (w1 >> np.uint64(16)).ravel()])
k0 = (f >= 6553).astype(float); c0 = k0 - k0.mean()
for label, s2 in (("lo+1", base + 1), ("lo+2", base + 2), ("lo^0x100", base ^ 0x100), ("hi+1", base + (1 << 32)), ("hi^bit20", base ^ (1 << 52)), ("random", 0x1122334455667788 & ((1 << 62) - 1)), ("0", 0), ("1", 1)):
    a0, a1 = fields2(s2)
    g = np.concatenate([(a0 & np.uint64(0xFFFF)).ravel(), (a0 >> np.uint64(16)).ravel(), (a1 & np.uint64(0xFFFF)).ravel(), (a1 >> np.uint64(16)).ravel()])
    k1 = (g >= 6553).astype(float); c1 = k1 - k1.mean()
    print("   corr with seed", label, round((c0 * c1).mean() / c0.var(), 5), " keep", round(k1.mean(), 5))
