#!/bin/bash
exec < /dev/null
# Round-4 validation visit: the whole GPU suite, smoke, the default bench line (CPU baseline included), rocprofv3 kernel stats +
# FETCH / WRITE PMC passes of the same command, bert-base / bert-base as one HIP graph / LLaVA bench lines, the GEMM library A/B
# against the build before the segmented-output / grouped-launch changes.
# usage: gpurun --timeout 1500 -- bash tools/gpu_r04_j.sh [tag]
tag=${1:-r04j}
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q --timeout 400 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cp $out/parity_hip.json $out/${tag}_parity_hip.json 2>/dev/null
tail -6 $out/${tag}_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
echo "smoke exit $?" >> $out/${tag}_smoke.log; tail -4 $out/${tag}_smoke.log
timeout 300 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench exit $?"; cut -c1-420 $out/${tag}_bench.json; tail -2 $out/${tag}_bench.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
cd $R
python tools/prof_traffic.py $out/$tag $out/${tag} r04 > $out/${tag}_traffic.log 2>&1
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv 2>/dev/null
head -14 $out/${tag}_bench_kernel_stats.csv | cut -c1-150
head -8 $out/${tag}_traffic.log
timeout 200 python bench.py --config llava --steps 20 --warmup 5 2> $out/${tag}_bench_llava.err | grep -m1 '^{"metric' | tee $out/${tag}_bench_llava.json | cut -c1-330
timeout 200 python bench.py --config bert-base --steps 20 --warmup 5 2> $out/${tag}_bench_bert.err | grep -m1 '^{"metric' | tee $out/${tag}_bench_bert.json | cut -c1-330
timeout 200 python bench.py --config bert-base --steps 20 --warmup 5 --hip-graph --no-cpu-baseline 2> $out/${tag}_bench_bert_graph.err | grep -m1 '^{"metric' | tee $out/${tag}_bench_bert_graph.json | cut -c1-330
timeout 200 python tools/gemm_lib_ab.py > $out/${tag}_gemm_lib_ab.jsonl 2> $out/${tag}_gemm_lib_ab.err; cut -c1-260 $out/${tag}_gemm_lib_ab.jsonl; tail -2 $out/${tag}_gemm_lib_ab.err
find $out/$tag -name "*.csv" -size +3M -delete
find $out/$tag -name "*.db" -delete
