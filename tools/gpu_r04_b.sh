#!/bin/bash
# Round-4 visit b: gemm_tw_kernel (two workgroups per CU) on the hardware: bit-identity tests, A/B against fl / sm / default /
# torch on bert-base, LLaVA and Llama-3-8B forward shapes; bert-base bench line after the activation went back to its own kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04b}
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -x -k "schedules_agree or bias_act_pre or colscale or embedding or accumulates" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
timeout 400 python tools/gemm_tw_ab.py > gpurun_out/${T}_gemm_tw_ab.jsonl 2> gpurun_out/${T}_gemm_tw_ab.err
cut -c1-420 gpurun_out/${T}_gemm_tw_ab.jsonl; tail -3 gpurun_out/${T}_gemm_tw_ab.err
timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${T}_bench_bert.err | tee gpurun_out/${T}_bench_bert.json | cut -c1-400
