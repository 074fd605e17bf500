#!/bin/bash
exec < /dev/null
# (RECORD of past visits: the kernel and tools/attn_fwd64_ab.py were removed, profiles/r03o_attn_fwd64_removed.patch)
# Round 3, GPU visit 6+: the 64-rows-per-wave attention forward (csrc/attention_fwd64.hip) against the 32-rows-per-wave
# kernel: bit-identity test on the silicon, stand-alone A/B over shapes, kernel stats of one arm each.
# usage: gpurun --timeout 600 -- bash tools/gpu_r03_f.sh [tag]
tag=${1:-r03f}
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 200 python -m pytest tests/test_kernels.py -m gpu -q -k "fwd64" --timeout 180 2>&1 | tail -5 > $out/${tag}_fwd64_test.log
cat $out/${tag}_fwd64_test.log
timeout 400 python tools/attn_fwd64_ab.py > $out/${tag}_attn_fwd64_ab.jsonl 2> $out/${tag}_attn_fwd64_ab.err
echo "ab exit $?"; cut -c1-400 $out/${tag}_attn_fwd64_ab.jsonl; tail -3 $out/${tag}_attn_fwd64_ab.err
