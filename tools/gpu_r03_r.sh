#!/bin/bash
# Round-3 visit r: where the LLaVA forward's 21 ms go with the 128 x 128 GEMM tile in place: rocprofv3 kernel stats of
# bench.py --config llava (eager and with the decoder stack as a HIP graph) next to its wall-clock line.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r03r}
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o llava -- python $R/bench.py --config llava --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${T}_prof_llava.log 2>&1 )
cp $(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_llava_kernel_stats.csv 2>/dev/null
head -14 gpurun_out/${T}_llava_kernel_stats.csv | cut -c1-160
tail -1 gpurun_out/${T}_prof_llava.log | cut -c1-300
TAMD_HIP_GRAPH=0 timeout 200 python bench.py --config llava --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/${T}_bench_llava_eager.json | cut -c1-300
