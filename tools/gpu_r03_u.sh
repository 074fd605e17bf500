#!/bin/bash
# Round-3 visit u (the round's last GPU seconds): attention dropout with one hash per 2 x 2 block on the hardware: the dropout
# kernel tests, the BERT / dropout model tests, the bert-base bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r03u}
timeout 45 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -k "dropout or bert_layer_op or bert_masked" -x > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
timeout 40 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/${T}_bench_bert.err | tee gpurun_out/${T}_bench_bert.json | cut -c1-300
