#!/bin/bash
exec < /dev/null
# Quick GPU visit: rocprofv3 kernel stats of 2 steps of the llama bench (+1 warm-up), optionally extra commands.
#   tools/gpu_quick.sh <tag> [extra command ...]
tag=${1:-quick}; shift
out=$PWD/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
for c in "$@"; do timeout 600 bash -c "$c"; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
cd $R
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv 2>/dev/null
find $out/$tag -name "*.csv" -size +3M -delete
grep -o '"ms_per_step": [0-9.]*' $out/${tag}_prof_bench.log
head -16 $out/${tag}_kernel_stats.csv | cut -c1-150
