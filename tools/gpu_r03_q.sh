#!/bin/bash
# Round-3 visit q: the 128 x 128 GEMM tile against the 256 x 256 kernel / split-K / torch.mm on small forward grids; GEMM
# tests on the hardware; LLaVA forward with the new default dispatch.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r03q}
timeout 200 python -m pytest tests/test_kernels.py -q -m gpu -k "gemm" -x > gpurun_out/${T}_gemm_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_gemm_tests.log
timeout 300 python tools/gemm_sm_ab.py > gpurun_out/${T}_gemm_sm_ab.jsonl 2> gpurun_out/${T}_gemm_sm_ab.err
echo "ab exit $?"; cut -c1-420 gpurun_out/${T}_gemm_sm_ab.jsonl; tail -3 gpurun_out/${T}_gemm_sm_ab.err
timeout 300 python bench.py --config llava --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${T}_bench_llava.err | tee gpurun_out/${T}_bench_llava.json | cut -c1-330
tail -2 gpurun_out/${T}_bench_llava.err
