#!/bin/bash
exec < /dev/null
# Round 3, GPU visit 4: the 8-wave attention kernels (forward, dQ) against the product kernels -- stand-alone and inside one
# decoder layer / the whole step; the dK/dV kernel's ablations (what its tile feed and barrier cost); LLaVA's kernel profile
# (the forward is GPU-bound at 28 ms: where); bert-base twice (box variance); the re-set BERT gate.
# usage: gpurun --timeout 1500 -- bash tools/gpu_r03_d.sh [tag]
# RECORD of the visit as it ran: the 8-wave kernels, `tools/attn_fwd8_ab.py` and the TAMD_ATTN_FWD8 switch were removed
# afterwards (not promoted; `git apply profiles/r03d_attn_8wave.patch` on the commit it names brings them back).
tag=${1:-r03d}
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python tools/attn_fwd8_ab.py > $out/${tag}_attn_fwd8_ab.jsonl 2> $out/${tag}_attn_fwd8_ab.err
echo "attn ab exit $?"; cat $out/${tag}_attn_fwd8_ab.jsonl | cut -c1-600; tail -2 $out/${tag}_attn_fwd8_ab.err
for sw in 0 3 0 3; do
  TAMD_ATTN_FWD8=$sw timeout 150 python tools/gpu_bench_kernels.py layer 2>/dev/null | sed "s/^/{\"attn_fwd8\": $sw} /" >> $out/${tag}_layer_fwd8_ab.txt
done
cut -c1-200 $out/${tag}_layer_fwd8_ab.txt
for sw in 3 0; do
  TAMD_ATTN_FWD8=$sw timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2> $out/${tag}_bench_f$sw.err | tee $out/${tag}_bench_f$sw.json | cut -c1-260
done
timeout 300 python tools/attn_dkdv_dbg.py 0 8 16 1 2 4 > $out/${tag}_dkdv_ablation.txt 2>&1
cat $out/${tag}_dkdv_ablation.txt | cut -c1-200
cd /tmp
TAMD_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/llava -o llava -- python $R/bench.py --config llava --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_prof_llava.log 2>&1
cd $R
cp $(find $out/$tag/llava -name "*kernel_stats.csv" | head -1) $out/${tag}_llava_kernel_stats.csv 2>/dev/null
head -16 $out/${tag}_llava_kernel_stats.csv | cut -c1-170
for i in 1 2; do
  timeout 300 python bench.py --config bert-base --steps 30 --warmup 10 --no-cpu-baseline 2>> $out/${tag}_bench_bert.err | tee -a $out/${tag}_bench_bert.jsonl | cut -c1-200
done
timeout 300 python bench.py --config llava --steps 20 --warmup 5 2> $out/${tag}_bench_llava.err | tee $out/${tag}_bench_llava.json | cut -c1-200
timeout 400 python -m pytest tests/test_models.py tests/test_kernels.py -m gpu -q -x --timeout 300 -k "bert_masked or decoder_stack or piece_placements" > $out/${tag}_tests.log 2>&1
tail -4 $out/${tag}_tests.log
find $out/$tag -name "*.csv" -size +3M -delete
