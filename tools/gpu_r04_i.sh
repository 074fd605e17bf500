#!/bin/bash
# Round-4 visit i: GEMM library A/B on one box (the build before the segmented-output / grouped-launch changes against the tree)
# -- visit h's Llama line was 4 % slow on a box whose clock probe was 4 % slow too; this separates the box from the code.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04i}
timeout 300 python tools/gemm_lib_ab.py > gpurun_out/${T}_gemm_lib_ab.jsonl 2> gpurun_out/${T}_gemm_lib_ab.err; cut -c1-260 gpurun_out/${T}_gemm_lib_ab.jsonl; tail -3 gpurun_out/${T}_gemm_lib_ab.err
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>> gpurun_out/${T}_bench.err | grep -m1 '^{"metric' | tee gpurun_out/${T}_bench_llama.json | cut -c1-300
