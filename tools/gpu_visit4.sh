#!/bin/bash
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python tools/gemm_sched_ab.py 3 > $out/v4_gemm_ab.log 2>&1
echo "ab exit $?" >> $out/v4_gemm_ab.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $out/v4_tests.log 2>&1
echo "tests exit $?" >> $out/v4_tests.log
timeout 300 python tools/bw_probe.py > $out/v4_bw_probe.log 2>&1
cat $out/v4_gemm_ab.log
tail -5 $out/v4_tests.log
