#!/bin/bash
exec < /dev/null
# Round-4 visit u: the secondary bench lines again on a fresh box (visit t's box ran bert-base 22.1 / LLaVA 22.8 ms where visits
# h / j had 18.6-19.3 / 20.6), then generate() with a pre-allocated cache now that accelerate() turns generate's torch.compile off.
tag=${1:-r04u}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for i in 1 2; do
  timeout 200 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2>> $out/${tag}_bench.err | grep -m1 '^{"metric' | tee -a $out/${tag}_bench_bert.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bert', d['ms_per_step'], (d.get('roofline') or {}).get('achieved'))"
  timeout 200 python bench.py --config llava --steps 20 --warmup 5 --no-cpu-baseline 2>> $out/${tag}_bench.err | grep -m1 '^{"metric' | tee -a $out/${tag}_bench_llava.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('llava', d['ms_per_step'], (d.get('roofline') or {}).get('achieved'))"
done
timeout 200 python tools/gemm_lib_ab.py > $out/${tag}_gemm_lib_ab.jsonl 2>> $out/${tag}_bench.err; cut -c1-200 $out/${tag}_gemm_lib_ab.jsonl | head -3
export DECODE_BENCH_LAYERS=32 DECODE_BENCH_BATCHES=1,8
for arm in tamd sdpa; do
  DECODE_BENCH_CACHE=static DECODE_BENCH_ARM=$arm timeout 400 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
done
DECODE_BENCH_ARM=tamd timeout 400 python tools/decode_bench.py generate >> $out/${tag}_decode_bench_32.jsonl 2>> $out/${tag}_decode_bench.err
cut -c1-300 $out/${tag}_decode_bench_32.jsonl
