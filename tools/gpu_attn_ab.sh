#!/bin/bash
export PYTHONUNBUFFERED=1
echo "== base"; timeout 300 python tools/gpu_bench_kernels.py attn 2>&1 | grep -v amdgpu
cp transformers_amd/libtamd_vgprform.so transformers_amd/libtamd.so
echo "== vgpr-form"; timeout 300 python tools/gpu_bench_kernels.py attn 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "attention" 2>&1 | tail -2
