#!/bin/bash
exec < /dev/null
# Round-4: quick check of the last commit's binding on hardware (the grouped / segmented weight-gradient ops, RCCL world-1 DDP tests, smoke).
tag=${1:-r04x}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q -x -k "gemm_group or segmented or rccl or ddp or bert_layer_op or gemv" > $out/${tag}_tests.log 2>&1
echo "tests exit $?"; tail -3 $out/${tag}_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke exit $?"; tail -2 $out/${tag}_smoke.log
