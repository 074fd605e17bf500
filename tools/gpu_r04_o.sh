#!/bin/bash
exec < /dev/null
# Round-4 visit o: is a batch-1 decode step of the 32-layer model bound by the host or by the GPU?  Three timings on one box,
# then the kernel stats of the same run (sum of kernel time per generated token against the wall time per token).
tag=${1:-r04o}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp DECODE_BENCH_LAYERS=32 DECODE_BENCH_BATCHES=1
for i in 1 2 3; do
  DECODE_BENCH_ARM=tamd timeout 300 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
done
DECODE_BENCH_ARM=sdpa timeout 300 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
cut -c1-330 $out/${tag}_decode_bench_32.jsonl
( cd /tmp && DECODE_BENCH_ARM=tamd timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o dec -- python $R/tools/decode_bench.py generate > /dev/null 2>&1 )
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_decode_kernel_stats.csv && head -30 $f | cut -c1-150
