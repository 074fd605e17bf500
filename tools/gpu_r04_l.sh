#!/bin/bash
exec < /dev/null
# Round-4 visit l: prompt-sized GEMMs with cold weights (HBM) against warm ones (Infinity Cache), every schedule.
tag=${1:-r04l}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python tools/gemm_cold_ab.py > $out/${tag}_gemm_cold_ab.jsonl 2> $out/${tag}_gemm_cold_ab.err; cat $out/${tag}_gemm_cold_ab.jsonl; tail -3 $out/${tag}_gemm_cold_ab.err
