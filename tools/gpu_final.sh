#!/bin/bash
exec < /dev/null  # nothing here may wait on stdin (an empty $(find ...) once turned `head` into a 15-minute hang)
# Round-end GPU visit: parity tests, rocprofv3 kernel stats + HBM counters of the bench command, the bench line.
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
cd $R
python tools/prof_summarize.py $out/$tag > $out/${tag}_summary.txt 2>&1
python tools/prof_traffic.py $out/$tag $out/${tag} > $out/${tag}_traffic.log 2>&1
cp profiles/r01_gemm_traffic.json $out/${tag}_gemm_traffic.json 2>/dev/null
cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv 2>/dev/null
find $out/$tag -name "*.csv" -size +3M -delete
timeout 1200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -3 $out/${tag}_tests.log
head -12 $out/${tag}_kernel_stats.csv | cut -c1-200
head -12 $out/${tag}_traffic.log
cat $out/${tag}_bench.json
