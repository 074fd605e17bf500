#!/bin/bash
exec < /dev/null
# Round 3: the dK/dV kernel with its plain (no mask test) loop body against every tile through the masked body
# (TAMD_DKDV_DBG=32, same library), interleaved; per-kernel micro-benchmarks; attention parity tests on the silicon.
# usage: gpurun --timeout 700 -- bash tools/gpu_r03_l.sh [tag]
tag=${1:-r03l}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python tools/attn_dkdv_dbg.py 0 32 0 32 0 32 > $out/${tag}_dkdv_plain_ab.txt 2>&1
cat $out/${tag}_dkdv_plain_ab.txt
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" --timeout 250 2>&1 | tail -4 | tee $out/${tag}_attn_tests.log
timeout 200 python tools/gpu_bench_kernels.py attn > $out/${tag}_attn_microbench.jsonl 2> $out/${tag}_attn_microbench.err
cut -c1-300 $out/${tag}_attn_microbench.jsonl
