#!/bin/bash
exec < /dev/null
# Round 2, visit B: full parity suite, every bench line, kernel microbench, rocprof stats of the llama and bert benches.
tag=${1:-r02b}
out=$PWD/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cp $out/parity_hip.json $out/${tag}_parity.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
echo "smoke exit $?" >> $out/${tag}_smoke.log
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 600 python bench.py --config bert-base --steps 20 --warmup 5 > $out/${tag}_bench_bert.json 2> $out/${tag}_bench_bert.err
timeout 600 python tools/bench_secondary.py sdpa,tamd bb > $out/${tag}_secondary_bb.jsonl 2> $out/${tag}_secondary_bb.err
timeout 600 python tools/gpu_bench_kernels.py gemm attn layer > $out/${tag}_microbench.jsonl 2> $out/${tag}_microbench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/bert -o bert -- python $R/bench.py --config bert-base --steps 5 --warmup 2 > $out/${tag}_prof_bert.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
cd $R
cp $(find $out/$tag/bert -name "*kernel_stats.csv" | head -1) $out/${tag}_bert_kernel_stats.csv 2>/dev/null
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv 2>/dev/null
find $out/$tag -name "*.csv" -size +3M -delete
tail -4 $out/${tag}_tests.log
tail -2 $out/${tag}_smoke.log
for f in bench bench_bert; do cut -c1-400 $out/${tag}_$f.json; tail -2 $out/${tag}_$f.err; done
cat $out/${tag}_secondary_bb.jsonl
cat $out/${tag}_microbench.jsonl
head -12 $out/${tag}_bert_kernel_stats.csv | cut -c1-160
head -12 $out/${tag}_kernel_stats.csv | cut -c1-160
