#!/bin/bash
exec < /dev/null
# Round-4: the GPU suite and smoke on the round's last commit.
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q --timeout 300 > $out/r04zz_tests.log 2>&1
echo "tests exit $?" >> $out/r04zz_tests.log; tail -3 $out/r04zz_tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $out/r04zz_smoke.log 2>&1; echo "smoke exit $?"; tail -1 $out/r04zz_smoke.log
