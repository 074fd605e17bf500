#!/bin/bash
exec < /dev/null
# Round 2, visit A: parity tests on the refactored (torch.ops) path, bench lines of every config, BERT profile.
out=$PWD/gpurun_out
mkdir -p $out/r02a
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $out/r02a_tests.log 2>&1
echo "tests exit $?" >> $out/r02a_tests.log
cp $out/parity_hip.json $out/r02a_parity.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/r02a_smoke.log 2>&1
echo "smoke exit $?" >> $out/r02a_smoke.log
timeout 900 python bench.py > $out/r02a_bench.json 2> $out/r02a_bench.err
timeout 600 python bench.py --fused-lm-head-loss --no-cpu-baseline > $out/r02a_bench_f1.json 2> $out/r02a_bench_f1.err
timeout 600 python bench.py --config bert-base --steps 20 --warmup 5 > $out/r02a_bench_bert.json 2> $out/r02a_bench_bert.err
timeout 600 python bench.py --config llava --steps 10 --warmup 3 > $out/r02a_bench_llava.json 2> $out/r02a_bench_llava.err
timeout 600 python tools/bench_secondary.py sdpa,tamd bb > $out/r02a_secondary_bb.jsonl 2> $out/r02a_secondary_bb.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/r02a/bert -o bert -- python $R/bench.py --config bert-base --steps 5 --warmup 2 > $out/r02a_prof_bert.log 2>&1
cd $R
cp $(find $out/r02a/bert -name "*kernel_stats.csv" | head -1) $out/r02a_bert_kernel_stats.csv 2>/dev/null
find $out/r02a -name "*.csv" -size +3M -delete
tail -4 $out/r02a_tests.log
tail -2 $out/r02a_smoke.log
for f in bench bench_f1 bench_bert bench_llava; do cut -c1-600 $out/r02a_$f.json; tail -2 $out/r02a_$f.err; done
cat $out/r02a_secondary_bb.jsonl
head -14 $out/r02a_bert_kernel_stats.csv | cut -c1-160
