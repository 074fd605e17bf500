#!/bin/bash
exec < /dev/null
# Round-4 visit m: cached decode on the kernels (csrc/gemv.hip for M = batch, fused layer around the reference's cache object):
# tests, generate() tokens/s per arm at 8 and 32 layers, kernel stats of the decode steps.
tag=${1:-r04m}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "gemv or cache or generate or decode or split_k" > $out/${tag}_tests.log 2>&1
echo "tests exit $?"; tail -3 $out/${tag}_tests.log
timeout 500 python tools/decode_bench.py generate > $out/${tag}_decode_bench.jsonl 2> $out/${tag}_decode_bench.err; cat $out/${tag}_decode_bench.jsonl | cut -c1-330; tail -2 $out/${tag}_decode_bench.err
DECODE_BENCH_LAYERS=32 DECODE_BENCH_ARM=tamd timeout 300 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
DECODE_BENCH_LAYERS=32 DECODE_BENCH_ARM=sdpa timeout 300 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
cut -c1-330 $out/${tag}_decode_bench_32.jsonl
( cd /tmp && DECODE_BENCH_ARM=tamd timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o dec -- python $R/tools/decode_bench.py generate > /dev/null 2>&1 )
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_decode_kernel_stats.csv && head -22 $f | cut -c1-170
