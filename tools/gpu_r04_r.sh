#!/bin/bash
exec < /dev/null
# Round-4 visit r: the M = batch products one by one (cold weights) and generate() at batch 8 / 16 after the MFMA streaming
# kernel issues its loads eight steps at a time.
tag=${1:-r04s}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp DECODE_BENCH_LAYERS=32
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -x -k "gemv" > $out/${tag}_tests.log 2>&1; echo "tests exit $?"; tail -2 $out/${tag}_tests.log
timeout 400 python tools/gemv_bench.py > $out/${tag}_gemv_bench.jsonl 2> $out/${tag}_gemv_bench.err; cut -c1-260 $out/${tag}_gemv_bench.jsonl; tail -2 $out/${tag}_gemv_bench.err
DECODE_BENCH_BATCHES=1,8,16 DECODE_BENCH_ARM=tamd timeout 400 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
cut -c1-300 $out/${tag}_decode_bench_32.jsonl
TAMD_GEMV_VALU_ROWS=0 timeout 400 python tools/gemv_bench.py > $out/${tag}_gemv_bench_mfma_only.jsonl 2>> $out/${tag}_gemv_bench.err
grep -h '"M": [124],' $out/${tag}_gemv_bench_mfma_only.jsonl | cut -c1-200
