#!/bin/bash
tag=${1:-r3}
out=$PWD/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
cd $R
python tools/prof_summarize.py $out/$tag > $out/${tag}_summary.txt 2>&1
python - <<'PY' > $out/${tag}_pmc_per_kernel.txt 2>&1
import csv, glob, collections, sys
for which in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/r3/{which}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
        for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:15]:
            print(which, k, "total_KB", round(v), "dispatches", n, "avg_KB", round(v / n))
PY
find $out/$tag -name "*.csv" -size +3M -delete
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -3 $out/${tag}_tests.log
head -30 $out/${tag}_summary.txt | cut -c1-300
cat $out/${tag}_pmc_per_kernel.txt | head -40
cat $out/${tag}_bench.json
