#!/bin/bash
# Round-4 visit d: decode attention (split-KV) tests + benchmark; split-K with bias / residual epilogues (tests, LLaVA bench line).
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04d}
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "decode or split_k or small_grid or cache or llava or clip" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
timeout 300 python tools/decode_bench.py kernels > gpurun_out/${T}_decode_bench.jsonl 2> gpurun_out/${T}_decode_bench.err
timeout 500 python tools/decode_bench.py generate >> gpurun_out/${T}_decode_bench.jsonl 2>> gpurun_out/${T}_decode_bench.err
cut -c1-400 gpurun_out/${T}_decode_bench.jsonl; tail -3 gpurun_out/${T}_decode_bench.err
timeout 200 python bench.py --config llava --steps 20 --warmup 5 2> gpurun_out/${T}_bench_llava.err | tee gpurun_out/${T}_bench_llava.json | cut -c1-1200
timeout 200 python tools/gemm_tw_ab.py llava > gpurun_out/${T}_gemm_small_ab.jsonl 2> /dev/null
cut -c1-330 gpurun_out/${T}_gemm_small_ab.jsonl
