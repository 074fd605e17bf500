#!/bin/bash
# One GPU visit: probe, parity tests, micro-benchmarks, headline bench.  Everything lands in gpurun_out/.
tag=${1:-r1}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.mem_get_info())" > gpurun_out/${tag}_env.log 2>&1
nproc >> gpurun_out/${tag}_env.log
timeout 900 python -m pytest tests/test_gpu_probe.py -m gpu -q --timeout 300 > gpurun_out/${tag}_probe.log 2>&1
echo "probe exit $?" >> gpurun_out/${tag}_probe.log
timeout 1200 python -m pytest tests/test_kernels.py tests/test_models.py -m gpu -q --timeout 600 > gpurun_out/${tag}_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/${tag}_tests.log
timeout 900 python tools/gpu_bench_kernels.py > gpurun_out/${tag}_kernels.jsonl 2> gpurun_out/${tag}_kernels.err
timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench exit $?" >> gpurun_out/${tag}_bench.err
tail -5 gpurun_out/${tag}_probe.log gpurun_out/${tag}_tests.log
cat gpurun_out/${tag}_kernels.jsonl | tail -40
cat gpurun_out/${tag}_bench.json
tail -5 gpurun_out/${tag}_bench.err
