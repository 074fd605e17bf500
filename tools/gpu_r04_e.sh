#!/bin/bash
# Round-4 visit e: graph-safe dropout (device seeds; a captured BERT training step), CLIP patch-embed GEMM, bert-base as one
# HIP graph per step against eager, LLaVA bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04e}
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "dropout or captured or clip or llava or bert" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/${T}_tests.log
timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline --gemm-timer off 2> gpurun_out/${T}_bench_bert.err | tee gpurun_out/${T}_bench_bert.json | cut -c1-260
timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline --gemm-timer off --hip-graph 2> gpurun_out/${T}_bench_bert_graph.err | tee gpurun_out/${T}_bench_bert_graph.json | cut -c1-260
tail -3 gpurun_out/${T}_bench_bert_graph.err
timeout 200 python bench.py --config llava --steps 20 --warmup 5 2> gpurun_out/${T}_bench_llava.err | tee gpurun_out/${T}_bench_llava.json | cut -c1-260
