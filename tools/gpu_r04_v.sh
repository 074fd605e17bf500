#!/bin/bash
exec < /dev/null
# Round-4 visit v: the decode products at 17 .. 128 rows (batched decode beyond the streaming kernels' 16 rows): ours (tile kernels /
# split-K by the library's policy) against torch.
tag=${1:-r04v}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
GEMV_BENCH_ROWS=16,17,32,64,128 timeout 500 python tools/gemv_bench.py > $out/${tag}_gemv_bench_rows.jsonl 2> $out/${tag}_gemv_bench.err; cut -c1-250 $out/${tag}_gemv_bench_rows.jsonl; tail -2 $out/${tag}_gemv_bench.err
