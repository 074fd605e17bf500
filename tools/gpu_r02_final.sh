#!/bin/bash
exec < /dev/null
# Round-2 final visit: full parity suite, every bench line, kernel microbench, rocprof stats (llama + bert), PMC traffic.
tag=${1:-r02z}
out=$PWD/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cp $out/parity_hip.json $out/${tag}_parity.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
echo "smoke exit $?" >> $out/${tag}_smoke.log
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 600 python bench.py --fused-lm-head-loss --no-cpu-baseline --steps 5 --warmup 2 > $out/${tag}_bench_f1.json 2> $out/${tag}_bench_f1.err
timeout 600 python bench.py --config bert-base --steps 20 --warmup 5 > $out/${tag}_bench_bert.json 2> $out/${tag}_bench_bert.err
timeout 600 python bench.py --config llava --steps 10 --warmup 3 > $out/${tag}_bench_llava.json 2> $out/${tag}_bench_llava.err
timeout 600 python tools/bench_secondary.py sdpa,tamd bb > $out/${tag}_secondary_bb.jsonl 2> $out/${tag}_secondary_bb.err
timeout 600 python tools/gpu_bench_kernels.py gemm attn hbm layer > $out/${tag}_microbench.jsonl 2> $out/${tag}_microbench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/bert -o bert -- python $R/bench.py --config bert-base --steps 5 --warmup 2 > $out/${tag}_prof_bert.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
cd $R
python tools/prof_traffic.py $out/$tag $out/${tag} > $out/${tag}_traffic.log 2>&1
cp $(find $out/$tag/bert -name "*kernel_stats.csv" | head -1) $out/${tag}_bert_kernel_stats.csv 2>/dev/null
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv 2>/dev/null
find $out/$tag -name "*.csv" -size +3M -delete
tail -4 $out/${tag}_tests.log
tail -2 $out/${tag}_smoke.log
for f in bench bench_f1 bench_bert bench_llava; do cut -c1-330 $out/${tag}_$f.json; tail -1 $out/${tag}_$f.err; done
cat $out/${tag}_secondary_bb.jsonl
grep -E "layer|attn|gate_up|down|o_proj|qkv" $out/${tag}_microbench.jsonl | cut -c1-200
head -14 $out/${tag}_kernel_stats.csv | cut -c1-150
head -12 $out/${tag}_traffic.log
