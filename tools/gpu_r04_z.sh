#!/bin/bash
exec < /dev/null
# Round-4 visit z: llama_layer_bwd with the layer's four weight gradients as one grouped launch (TAMD_LLAMA_GROUP_DW=1) against
# one product each (split-K for q|k|v and down_proj): the step, interleaved twice on one box; the layer / model tests with it.
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAMD_LLAMA_GROUP_DW=1 timeout 300 python -m pytest tests/test_models.py tests/test_torch_ops.py -q -m gpu -x -k "llama or layer" > $out/r04z_tests.log 2>&1; echo "tests exit $?"; tail -2 $out/r04z_tests.log
for v in 0 1 0 1; do
  TAMD_LLAMA_GROUP_DW=$v timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>> $out/r04z_bench.err | grep -m1 '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['group_dw']=$v; print(json.dumps(d))" | tee -a $out/r04z_bench_ab.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('group_dw', d['group_dw'], d['ms_per_step'], d['roofline']['achieved'], d['max_memory_gb'])"
done
