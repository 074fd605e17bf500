"""Per-kernel HBM traffic from rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE collected in separate
runs, as gpurun requires).  Writes a per-kernel table and profiles/r02_gemm_traffic.json (read by bench.py).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md
section HBM), so read bytes = 2 x FETCH_SIZE x 1024.

    python tools/prof_traffic.py <dir with fetch/ and write/ subdirs> <out prefix>"""
import collections
import csv
import glob
import json
import sys

root, prefix = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r02"
agg = {"fetch": collections.defaultdict(lambda: [0.0, 0]), "write": collections.defaultdict(lambda: [0.0, 0])}
for which in agg:
    for f in glob.glob(f"{root}/{which}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            agg[which][k][0] += float(row["Counter_Value"])
            agg[which][k][1] += 1
lines = []
gemm = {"fetch": [0.0, 0], "write": [0.0, 0]}
for k in sorted(set(agg["fetch"]) | set(agg["write"]), key=lambda k: -(agg["fetch"][k][0] + agg["write"][k][0])):
    f, nf = agg["fetch"][k]
    w, nw = agg["write"][k]
    n = max(nf, nw, 1)
    lines.append(f"{k[:100]:100s} dispatches {n:5d}  read_MB/launch {2 * f * 1024 / n / 1e6:10.1f}  "
                 f"write_MB/launch {w * 1024 / n / 1e6:10.1f}")
    if "gemm_" in k:
        gemm["fetch"][0] += f
        gemm["fetch"][1] += nf
        gemm["write"][0] += w
        gemm["write"][1] += nw
open(prefix + "_pmc_per_kernel.txt", "w").write("\n".join(lines[:40]) + "\n")
if gemm["fetch"][1] and gemm["write"][1]:
    fb = 2 * gemm["fetch"][0] * 1024 / gemm["fetch"][1]
    wb = gemm["write"][0] * 1024 / gemm["write"][1]
    summary = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
               "gemm_launches_profiled": gemm["fetch"][1],
               "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1`; "
                      "all tamd::gemm_* dispatches; read bytes = 2 x FETCH_SIZE KiB (gfx950 correction)"}
    for path in (f"profiles/{rnd}_gemm_traffic.json", prefix + "_gemm_traffic.json"):  # (gpurun merges gpurun_out/ back, not profiles/)
        json.dump(summary, open(path, "w"), indent=1)
print("\n".join(lines[:25]))
