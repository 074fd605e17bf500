"""Condense rocprofv3 CSV output (kernel stats / counter collection) into a small per-kernel table."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(root + "/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 25:
            print({k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage") if k in row})
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    print("==", f)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")[:70]
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(name, row["Counter_Name"])] += 1
    for name, d in agg.items():
        if not any(s in name for s in ("gemm", "Cijk", "attn", "rmsnorm", "swiglu")):
            continue
        print(name, {c: round(v / max(cnt[(name, c)], 1), 1) for c, v in d.items()}, "dispatches",
              max(cnt[(name, c)] for c in d))
