#!/bin/bash
exec < /dev/null
# First GPU visit of the next round (what round 4 left unmeasured or open; every line is one existing command):
#  1. the suite + smoke + default bench on the new box (box spread of round 4: 1242.8-1311.8 ms; clock probe 1.68-1.75 GHz);
#  2. the multi-GPU facts if the box has more than one GPU: bench.py under torchrun at N = 2 with and without the zero-copy
#     hand-over (never run over RCCL at world size > 1);
#  3. generate() arms (dynamic / static cache) as the baseline for a decode step captured in one HIP graph.
tag=${1:-r05a}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q --timeout 400 > $out/${tag}_tests.log 2>&1; echo "tests exit $?"; tail -3 $out/${tag}_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke exit $?"
timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench.json 2> $out/${tag}_bench.err; cut -c1-300 $out/${tag}_bench.json
n=$(python -c "import torch; print(torch.cuda.device_count())")
if [ "$n" -ge 2 ]; then
  for arm in "" "--no-ddp-zero-copy"; do
    timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline $arm 2>> $out/${tag}_bench_n2.err | grep -m1 '^{"metric' >> $out/${tag}_bench_n2.jsonl
  done
  cut -c1-300 $out/${tag}_bench_n2.jsonl
fi
export DECODE_BENCH_LAYERS=32 DECODE_BENCH_BATCHES=1,8
for cache in "" static; do
  DECODE_BENCH_CACHE=$cache DECODE_BENCH_ARM=tamd timeout 400 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
done
cut -c1-300 $out/${tag}_decode_bench_32.jsonl
