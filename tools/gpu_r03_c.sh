#!/bin/bash
exec < /dev/null
# Round 3, GPU visit 3: validation of HEAD with the compiled torch binding -- the whole parity suite, smoke, the default
# bench line (CPU baseline included), rocprofv3 kernel stats + FETCH/WRITE PMC passes of the same command, the secondary
# configurations (LLaVA with the decoder stack as a HIP graph, bert-base + its kernel stats), DDP gradient handling A/B,
# the piece-placement A/B, per-kernel micro-benchmarks, the attention phase trace.
# usage: gpurun --timeout 1800 -- bash tools/gpu_r03_c.sh [tag]
tag=${1:-r03c}
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q --timeout 600 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cp $out/parity_hip.json $out/${tag}_parity.json 2>/dev/null
tail -6 $out/${tag}_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
echo "smoke exit $?" >> $out/${tag}_smoke.log; tail -4 $out/${tag}_smoke.log
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench exit $?"; cut -c1-420 $out/${tag}_bench.json; tail -2 $out/${tag}_bench.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/bert -o bert -- python $R/bench.py --config bert-base --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_prof_bert.log 2>&1
cd $R
python tools/prof_traffic.py $out/$tag $out/${tag} r03 > $out/${tag}_traffic.log 2>&1
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv 2>/dev/null
cp $(find $out/$tag/bert -name "*kernel_stats.csv" | head -1) $out/${tag}_bert_kernel_stats.csv 2>/dev/null
head -14 $out/${tag}_bench_kernel_stats.csv | cut -c1-150
head -8 $out/${tag}_traffic.log
head -14 $out/${tag}_bert_kernel_stats.csv | cut -c1-150
timeout 300 python bench.py --config llava --steps 20 --warmup 5 2> $out/${tag}_bench_llava.err | tee $out/${tag}_bench_llava.json | cut -c1-330
TAMD_HIP_GRAPH=0 timeout 300 python bench.py --config llava --steps 20 --warmup 5 2> $out/${tag}_bench_llava_eager.err | tee $out/${tag}_bench_llava_eager.json | cut -c1-330
timeout 300 python bench.py --config bert-base --steps 20 --warmup 5 2> $out/${tag}_bench_bert.err | tee $out/${tag}_bench_bert.json | cut -c1-330
for g in none zero keep; do
  timeout 300 python bench.py --force-ddp --ddp-grads $g --steps 4 --warmup 2 --no-cpu-baseline 2> $out/${tag}_ddp_$g.err | tee -a $out/${tag}_ddp_grads_ab.jsonl | cut -c1-260
done
timeout 300 python tools/gemm_piece_ab.py --shapes qkv,gate_up,down,lm_head > $out/${tag}_gemm_piece_ab.jsonl 2> $out/${tag}_gemm_piece_ab.err
python - <<PY
import json
for line in open("gpurun_out/${tag}_gemm_piece_ab.jsonl"):
    r = json.loads(line)
    print(r["shape"], r["leg"], r["median_vs_fl"], {c: sorted(x for x in v if not isinstance(x, str))[len(v) // 2] for c, v in r["tflops"].items()})
PY
timeout 300 python tools/gpu_bench_kernels.py gemm attn hbm layer > $out/${tag}_kernel_microbench.jsonl 2> $out/${tag}_microbench.err
grep -E "attn|layer|gate_up|down|o_proj|qkv|lm_head|norm|swiglu" $out/${tag}_kernel_microbench.jsonl | cut -c1-200
timeout 100 python tools/attn_phases.py > $out/${tag}_attn_phases.txt 2>&1; tail -8 $out/${tag}_attn_phases.txt
find $out/$tag -name "*.csv" -size +3M -delete
