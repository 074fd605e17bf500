#!/bin/bash
exec < /dev/null  # nothing here may wait on stdin (an empty $(find ...) once turned `head` into a 15-minute hang)
# What the driver runs at round end, in one visit: GPU parity tests, smoke(), the default bench line.
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 300 > $out/chk_tests.log 2>&1 < /dev/null
echo "tests exit $?" >> $out/chk_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/chk_smoke.log 2>&1 < /dev/null
echo "smoke exit $?" >> $out/chk_smoke.log
timeout 400 python bench.py > $out/chk_bench.json 2> $out/chk_bench.err < /dev/null
tail -3 $out/chk_tests.log
tail -3 $out/chk_smoke.log
cat $out/chk_bench.json | cut -c1-700
