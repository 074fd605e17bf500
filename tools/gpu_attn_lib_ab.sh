#!/bin/bash
# The attention kernels of the tree against a build of an earlier HEAD (tools/ab/libtamd_base.so), interleaved in one
# process (tools/attn_lib_ab.py); per-kernel times of the tree's kernels at the Llama-3-8B shape (rocprofv3); the attention
# tests on the hardware.   usage: TAG=r03p gpurun --timeout 600 -- bash tools/gpu_attn_lib_ab.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 python tools/attn_lib_ab.py > gpurun_out/${TAG:-r03p}_attn_lib_ab.jsonl 2> gpurun_out/${TAG:-r03p}_attn_lib_ab.err
echo "ab exit $?"
cat gpurun_out/${TAG:-r03p}_attn_lib_ab.jsonl
tail -3 gpurun_out/${TAG:-r03p}_attn_lib_ab.err
( cd /tmp && AB_LIBS=new AB_SHAPES=llama3-8b timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG:-r03p} -o ab -- python $OLDPWD/tools/attn_lib_ab.py > /dev/null 2>&1 )
cp $(find /tmp/prof_${TAG:-r03p} -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG:-r03p}_attn_kernel_stats.csv 2>/dev/null
head -6 gpurun_out/${TAG:-r03p}_attn_kernel_stats.csv | cut -c1-200
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "attention and not fwd64" -x > gpurun_out/${TAG:-r03p}_attn_tests.log 2>&1
echo "tests exit $?"
tail -5 gpurun_out/${TAG:-r03p}_attn_tests.log
