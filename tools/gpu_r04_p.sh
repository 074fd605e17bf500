#!/bin/bash
exec < /dev/null
# Round-4 visit p: decode after the parallel split-KV merge and the in-place rotary of the cached path: tests, then visit o's
# measurements again (batch 1, 32 layers: three timings, sdpa, kernel stats) and batch 8.
tag=${1:-r04p}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "gemv or cache or generate or decode or llava" > $out/${tag}_tests.log 2>&1
echo "tests exit $?"; tail -3 $out/${tag}_tests.log
bash tools/gpu_r04_o.sh $tag
DECODE_BENCH_LAYERS=32 DECODE_BENCH_BATCHES=8 DECODE_BENCH_ARM=tamd timeout 300 python tools/decode_bench.py generate 2>> $out/${tag}_decode_bench.err | tee -a $out/${tag}_decode_bench_32.jsonl | cut -c1-300
timeout 120 python tools/decode_bench.py kernels > $out/${tag}_decode_kernels.jsonl 2>> $out/${tag}_decode_bench.err; cut -c1-300 $out/${tag}_decode_kernels.jsonl | head -12
