#!/bin/bash
exec < /dev/null
# Round-4 visit q: M = 5 .. 16 products on the MFMA streaming kernel: tests, generate() at batch 8 / 16 (32 layers) against sdpa,
# kernel stats at batch 8.
tag=${1:-r04q}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp DECODE_BENCH_LAYERS=32
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "gemv or cache or generate or decode or llava or swiglu" > $out/${tag}_tests.log 2>&1
echo "tests exit $?"; tail -3 $out/${tag}_tests.log
for arm in tamd sdpa; do
  DECODE_BENCH_BATCHES=1,8,16 DECODE_BENCH_ARM=$arm timeout 400 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
done
cut -c1-300 $out/${tag}_decode_bench_32.jsonl
( cd /tmp && DECODE_BENCH_BATCHES=8 DECODE_BENCH_ARM=tamd timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o dec -- python $R/tools/decode_bench.py generate > /dev/null 2>&1 )
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_decode_b8_kernel_stats.csv && head -14 $f | cut -c1-150
