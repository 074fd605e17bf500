#!/bin/bash
exec < /dev/null
# Round-4 closing visit: tools/gpu_r04_j.sh again on the final tree (whole GPU suite, smoke, bench line + rocprofv3 + PMC passes,
# secondary bench lines, GEMM library A/B), then generate() with a pre-allocated cache next to the dynamic one.
tag=${1:-r04t}
bash tools/gpu_r04_j.sh $tag
out=$PWD/gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp DECODE_BENCH_LAYERS=32 DECODE_BENCH_BATCHES=1,8
for cache in "" static; do
  for arm in tamd sdpa; do
    DECODE_BENCH_CACHE=$cache DECODE_BENCH_ARM=$arm timeout 400 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
  done
done
cut -c1-300 $out/${tag}_decode_bench_32.jsonl; tail -3 $out/${tag}_decode_bench.err
