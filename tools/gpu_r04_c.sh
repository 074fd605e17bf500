#!/bin/bash
# Round-4 visit c: (1) the cheaper attention-dropout mix on the hardware: dropout tests, A/B against the round-3 opening build
# (same yardstick as r04a: base fwd 0.1007 / bwd 0.223 ms at bert-base, the 2 x 2 hash of r04a 0.0928 / 0.1953);
# (2) timeline of the big GEMMs (ramp, tail, XCD balance); (3) bert-base bench + kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04c}
R=$PWD
timeout 300 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "dropout" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
AB_SHAPES=bert-base,clip-l AB_DROPOUT=0.1 timeout 90 python tools/attn_lib_ab.py > gpurun_out/${T}_attn_dropout_ab.jsonl 2> gpurun_out/${T}_attn_dropout_ab.err
cut -c1-420 gpurun_out/${T}_attn_dropout_ab.jsonl; tail -2 gpurun_out/${T}_attn_dropout_ab.err
timeout 200 python tools/gemm_timeline.py > gpurun_out/${T}_gemm_timeline.jsonl 2> gpurun_out/${T}_gemm_timeline.err
cut -c1-900 gpurun_out/${T}_gemm_timeline.jsonl; tail -3 gpurun_out/${T}_gemm_timeline.err
timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${T}_bench_bert.err | tee gpurun_out/${T}_bench_bert.json | cut -c1-300
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o bert -- python $R/bench.py --config bert-base --steps 6 --warmup 2 --no-cpu-baseline --gemm-timer off > /dev/null 2>&1 )
cp $(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_bert_kernel_stats.csv 2>/dev/null
head -16 gpurun_out/${T}_bert_kernel_stats.csv | cut -c1-160
