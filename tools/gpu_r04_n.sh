#!/bin/bash
exec < /dev/null
# Round-4 visit n: cached decode after the per-forward mask conversion and SiLU*up inside the M = batch gate|up product.
tag=${1:-r04n}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "gemv or cache or generate or decode or llava or swiglu" > $out/${tag}_tests.log 2>&1
echo "tests exit $?"; tail -3 $out/${tag}_tests.log
for arm in tamd sdpa; do
  DECODE_BENCH_LAYERS=32 DECODE_BENCH_ARM=$arm timeout 300 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
done
DECODE_BENCH_ARM=tamd timeout 300 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_8.jsonl 2>> $out/${tag}_decode_bench.err
cut -c1-330 $out/${tag}_decode_bench_32.jsonl $out/${tag}_decode_bench_8.jsonl
