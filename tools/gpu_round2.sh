#!/bin/bash
tag=${1:-r2}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q --timeout 600 -k "gemm or linear or rope" > gpurun_out/${tag}_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/${tag}_tests.log
timeout 600 python tools/gpu_bench_kernels.py gemm layer > gpurun_out/${tag}_kernels.jsonl 2> gpurun_out/${tag}_kernels.err
TAMD_GEMM=v1 timeout 600 python tools/gpu_bench_kernels.py layer > gpurun_out/${tag}_kernels_v1.jsonl 2>&1
timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -4 gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_kernels.jsonl gpurun_out/${tag}_kernels_v1.jsonl
cat gpurun_out/${tag}_bench.json
tail -3 gpurun_out/${tag}_bench.err
