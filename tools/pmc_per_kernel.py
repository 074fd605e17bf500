"""Per-kernel averages of whatever counters a `rocprofv3 --pmc ... --output-format csv` pass collected.

    python tools/pmc_per_kernel.py <dir> [--match substr] [--ratio NUM/DEN ...] [--out file]

One line per kernel name (template arguments kept, argument list cut): dispatches, then counter = mean per dispatch;
`--ratio A/B` appends A/B computed from the per-kernel sums."""
import argparse
import collections
import csv
import glob

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--match", default="")
ap.add_argument("--ratio", action="append", default=[])
ap.add_argument("--out")
ap.add_argument("--top", type=int, default=30)
args = ap.parse_args()

total = collections.defaultdict(lambda: collections.defaultdict(float))
count = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(f"{args.root}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if args.match in k:
            total[k][row["Counter_Name"]] += float(row["Counter_Value"])
            count[k][row["Counter_Name"]] += 1
lines = []
for k in sorted(total, key=lambda k: -sum(total[k].values()))[:args.top]:
    parts = [f"{k[:96]:96s} dispatches {max(count[k].values()):5d}"]
    for c in sorted(total[k]):
        parts.append(f"{c} {total[k][c] / count[k][c]:.4g}")
    for r in args.ratio:
        a, b = r.split("/")
        if total[k].get(b):
            parts.append(f"{r} {total[k].get(a, 0.0) / total[k][b]:.5f}")
    lines.append("  ".join(parts))
text = "\n".join(lines) + "\n"
if args.out:
    open(args.out, "w").write(text)
print(text, end="")
