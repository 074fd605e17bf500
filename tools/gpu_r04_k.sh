#!/bin/bash
exec < /dev/null
# Round-4 visit k: fp32 atomic-add throughput probe (one-pass attention backward decision), LLaVA kernel stats of the final tree,
# fused SwiGLU epilogue against GEMM + kernel at prompt-sized M.
tag=${1:-r04k}
R=$PWD
out=$R/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 tools/probes/atomic_add_probe.hip -o /tmp/aprobe 2> /dev/null && timeout 120 /tmp/aprobe > $out/${tag}_atomic_add_probe.jsonl; cat $out/${tag}_atomic_add_probe.jsonl
timeout 200 python tools/gemm_swiglu_small_ab.py > $out/${tag}_swiglu_small_ab.jsonl 2> $out/${tag}_swiglu_small_ab.err; cat $out/${tag}_swiglu_small_ab.jsonl; tail -2 $out/${tag}_swiglu_small_ab.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o llava -- python $R/bench.py --config llava --steps 8 --warmup 3 --no-cpu-baseline --gemm-timer off > /dev/null 2>&1 )
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_llava_kernel_stats.csv && head -24 $f | cut -c1-170
