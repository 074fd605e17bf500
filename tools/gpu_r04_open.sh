#!/bin/bash
# First GPU visit of the next round (prepared at the end of round 3, when the GPU budget was spent): what could not be
# measured any more.
#   1. attention with dropout 0.1 at the bert-base / CLIP shapes, the tree against the round-3 opening build
#      (tools/ab/libtamd_base.so: per-element hash, no srcC chains) -- the 2 x 2-block dropout's per-kernel gain;
#   2. the same without dropout (the srcC chains alone) for the split;
#   3. bert-base bench line + rocprofv3 kernel stats of it (attention rows: 116 / 131 / 154 us per layer before).
# (tools/ab/libtamd_base.so is git-ignored: tools/build_base_lib.sh rebuilds it from commit 23d1524 if it is gone)
# usage: gpurun --timeout 300 -- bash tools/gpu_r04_open.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04a}
R=$PWD
AB_SHAPES=bert-base,clip-l AB_DROPOUT=0.1 timeout 60 python tools/attn_lib_ab.py > gpurun_out/${T}_attn_dropout_ab.jsonl 2> gpurun_out/${T}_attn_dropout_ab.err
AB_SHAPES=bert-base,clip-l timeout 60 python tools/attn_lib_ab.py > gpurun_out/${T}_attn_nodrop_ab.jsonl 2>> gpurun_out/${T}_attn_dropout_ab.err
cat gpurun_out/${T}_attn_dropout_ab.jsonl gpurun_out/${T}_attn_nodrop_ab.jsonl | cut -c1-500; tail -2 gpurun_out/${T}_attn_dropout_ab.err
timeout 100 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tee gpurun_out/${T}_bench_bert.json | cut -c1-300
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o bert -- python $R/bench.py --config bert-base --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
cp $(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_bert_kernel_stats.csv 2>/dev/null
head -12 gpurun_out/${T}_bert_kernel_stats.csv | cut -c1-170
