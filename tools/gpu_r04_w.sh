#!/bin/bash
exec < /dev/null
# Round-4 last visit: the GPU suite, smoke and the default bench line on the final commit.
tag=${1:-r04w}
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q --timeout 400 > $out/${tag}_tests.log 2>&1
echo "tests exit $?" >> $out/${tag}_tests.log; tail -4 $out/${tag}_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
echo "smoke exit $?" >> $out/${tag}_smoke.log; tail -2 $out/${tag}_smoke.log
timeout 300 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench exit $?"; cut -c1-330 $out/${tag}_bench.json
