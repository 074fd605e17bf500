#!/bin/bash
exec < /dev/null
# Round-4: bench.py under DDP over RCCL (world size 1) on the last commit.
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --force-ddp 2> $out/r04y_bench_ddp.err | grep -m1 '^{"metric' | tee $out/r04y_bench_ddp.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['parallelism'], d['ddp_zero_copy'])"
tail -2 $out/r04y_bench_ddp.err
