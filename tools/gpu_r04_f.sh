#!/bin/bash
# Round-4 visit f: zero-copy DDP gradients over RCCL at world size 1 (the DDP wrapper's cost against the plain step, with the
# hand-over by copy and by writing into the buckets), the tests added since visit e (captured training step, segmented dW).
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04f}
timeout 500 python -m pytest tests/test_kernels.py tests/test_models.py tests/test_ddp_gloo.py -q -m gpu -x -k "segmented or captured or bert_layer_op or rccl or seed_from_device" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -4 gpurun_out/${T}_tests.log
for arm in "" "--force-ddp --no-ddp-zero-copy" "--force-ddp"; do
  timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline $arm 2>> gpurun_out/${T}_ddp.err | tee -a gpurun_out/${T}_ddp_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('ms_per_step','max_memory_gb','ddp_zero_copy')}, d['config']['parallelism'], d['roofline']['achieved'])"
done
tail -3 gpurun_out/${T}_ddp.err
