#!/bin/bash
# Round-3 visit t: ABI 7 (pre-scaled queries from the rotary kernel) on the hardware: attention / rotary kernel tests, the
# Llama model parity tests, a short bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r03t}
timeout 250 python -m pytest tests/test_kernels.py tests/test_models.py tests/test_torch_ops.py -q -m gpu -k "attention or rope or llama or full_size or torch_ops" -x > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -4 gpurun_out/${T}_tests.log
cp gpurun_out/parity_hip.json gpurun_out/${T}_parity.json 2>/dev/null
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2> gpurun_out/${T}_bench.err | tee gpurun_out/${T}_bench.json | cut -c1-400
tail -2 gpurun_out/${T}_bench.err
