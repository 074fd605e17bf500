#!/bin/bash
exec < /dev/null
# Round 3, GPU visit 1 (VERDICT r2 "next round" item 1 + the prepared A/Bs):
#   parity suite at HEAD (new: f1 vs the reference, Llama-2-7B-shaped layer, cross-attention), bench line, rocprofv3
#   kernel stats + FETCH/WRITE PMC passes of the SAME command, GEMM stagger / early-DMA / second-barrier A/B, rotary
#   epilogue A/B, attention pair-tile A/B, LLaVA eager vs HIP graph, bert-base.
# usage: gpurun --timeout 1500 -- bash tools/gpu_r03_a.sh [tag]
tag=${1:-r03a}
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cp $out/parity_hip.json $out/${tag}_parity.json 2>/dev/null
tail -5 $out/${tag}_tests.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cut -c1-400 $out/${tag}_bench.json; tail -2 $out/${tag}_bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
cd $R
python tools/prof_traffic.py $out/$tag $out/${tag} r03 > $out/${tag}_traffic.log 2>&1
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv 2>/dev/null
head -16 $out/${tag}_bench_kernel_stats.csv | cut -c1-160
head -14 $out/${tag}_traffic.log
# GEMM schedule variants vs the product kernel
timeout 300 python tools/gemm_stagger_ab.py > $out/${tag}_gemm_stagger_ab.jsonl 2> $out/${tag}_gemm_stagger_ab.err
echo "stagger exit $?"
python - <<PY
import json
for line in open("gpurun_out/${tag}_gemm_stagger_ab.jsonl"):
    r = json.loads(line)
    med = {c: sorted(v)[len(v) // 2] for c, v in r["tflops"].items()}
    top = sorted(med, key=med.get, reverse=True)[:5]
    print(r["shape"], r["layout"], "off", med["off"], " | ", "  ".join(f"{c} {med[c]}" for c in top))
PY
for f in 0 1 0 1; do
  TAMD_FUSE_ROPE_FWD=$f timeout 150 python tools/gpu_bench_kernels.py layer 2>/dev/null | sed "s/^/{\"rope_fwd_fused\": $f} /" >> $out/${tag}_rope_fwd_ab.txt
done
cat $out/${tag}_rope_fwd_ab.txt | cut -c1-260
timeout 200 python tools/attn_fwd64_ab.py > $out/${tag}_attn_fwd_ab.jsonl 2> $out/${tag}_attn_fwd_ab.err
cat $out/${tag}_attn_fwd_ab.jsonl
timeout 300 python tools/gpu_bench_kernels.py gemm attn hbm > $out/${tag}_kernel_microbench.jsonl 2> $out/${tag}_microbench.err
grep -E "attn|gate_up|down|o_proj|qkv|lm_head|norm|swiglu" $out/${tag}_kernel_microbench.jsonl | cut -c1-220
for extra in "" "--hip-graph"; do
  timeout 240 python bench.py --config llava --steps 20 --warmup 5 --no-cpu-baseline $extra 2> $out/${tag}_llava${extra}.err | tee -a $out/${tag}_llava_graph_ab.jsonl | cut -c1-330
done
timeout 300 python bench.py --config bert-base --steps 20 --warmup 5 2> $out/${tag}_bench_bert.err | tee $out/${tag}_bench_bert.json | cut -c1-330
find $out/$tag -name "*.csv" -size +3M -delete
