#!/bin/bash
# Round-4 visit a: (1) the new GPU test cases (real-vocab BERT MLM parity, ABI 8 fusions, pre-scaled q records),
# (2) same-box A/B of the 2 x 2-block attention dropout against the round-3 opening build (tools/ab/libtamd_base.so),
# (3) bert-base bench line with its roofline object + rocprofv3 kernel stats, (4) GEMM micro-benchmarks next to torch.mm.
# usage: gpurun --timeout 900 -- bash tools/gpu_r04_a.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04a}
R=$PWD
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x \
  -k "bias_act_pre or colscale or accumulates or prescaled or layernorm or bias_act or dropout or bert or linear_autograd or schedules_agree or epilogues" \
  > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -4 gpurun_out/${T}_tests.log
cp gpurun_out/parity_hip.json gpurun_out/${T}_parity.json 2>/dev/null
AB_SHAPES=bert-base,clip-l AB_DROPOUT=0.1 timeout 90 python tools/attn_lib_ab.py > gpurun_out/${T}_attn_dropout_ab.jsonl 2> gpurun_out/${T}_attn_dropout_ab.err
AB_SHAPES=bert-base,clip-l timeout 90 python tools/attn_lib_ab.py > gpurun_out/${T}_attn_nodrop_ab.jsonl 2>> gpurun_out/${T}_attn_dropout_ab.err
cat gpurun_out/${T}_attn_dropout_ab.jsonl gpurun_out/${T}_attn_nodrop_ab.jsonl | cut -c1-420; tail -2 gpurun_out/${T}_attn_dropout_ab.err
timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${T}_bench_bert.err | tee gpurun_out/${T}_bench_bert.json | cut -c1-1500
TAMD_BERT_PRESCALE=0 timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline --gemm-timer off 2>> gpurun_out/${T}_bench_bert.err | tee gpurun_out/${T}_bench_bert_noprescale.json | cut -c1-300
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o bert -- python $R/bench.py --config bert-base --steps 6 --warmup 2 --no-cpu-baseline --gemm-timer off > /dev/null 2>&1 )
cp $(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_bert_kernel_stats.csv 2>/dev/null
head -24 gpurun_out/${T}_bert_kernel_stats.csv | cut -c1-200
timeout 200 python tools/gpu_bench_kernels.py gemm > gpurun_out/${T}_kernel_microbench.jsonl 2> gpurun_out/${T}_kernel_microbench.err
cut -c1-260 gpurun_out/${T}_kernel_microbench.jsonl
