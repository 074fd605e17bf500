#!/bin/bash
exec < /dev/null
# L2 hit rate of the GEMM layouts at short and long K, and of hipBLASLt on the gate|up shape (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum)
R=$PWD
out=$R/gpurun_out/l2
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out -o l2 -- python $R/tools/gemm_l2_pmc.py > $out/run.log 2>&1
cd $R
tail -3 $out/run.log
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[f'{r["Kernel_Name"][:70]:70s} grid {int(r["Grid_Size"]) // max(int(r.get("Workgroup_Size", 256) or 256), 1):6d}'][r["Counter_Name"]] += float(r["Counter_Value"])
lines = []
for k, v in agg.items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    if h + m > 1e6:
        lines.append(f"{k:84s} TCC_HIT {int(h):>12d} TCC_MISS {int(m):>12d} hit rate {h / (h + m):.4f}")
open("gpurun_out/gemm_l2_hit_by_shape.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $out -name "*.csv" -size +2M -delete
