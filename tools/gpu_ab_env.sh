#!/bin/bash
exec < /dev/null
# Same-box A/B of the default bench line under an environment switch, arms interleaved A B A B:
#     gpurun -- bash tools/gpu_ab_env.sh <tag> <VAR> <value A> <value B> [bench.py args...]
tag=${1:?tag}; var=${2:?variable}; a=$3; b=$4
shift 4
out=$PWD/gpurun_out
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for arm in "$a" "$b" "$a" "$b"; do
  env $var=$arm timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" 2>> $out/${tag}_ab.err |
    grep -m1 '^{"metric' | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(json.dumps({'$var': '$arm', 'ms_per_step': d['ms_per_step'], 'roofline_frac': d['roofline']['frac'], 'launches': d['roofline']['launches_per_step'], 'layer_forward_ms': (d.get('layer_forward') or {}).get('ms'), 'clock_GHz': d['roofline'].get('clock_probe', {}).get('clock_GHz'), 'loss': d['loss'], 'swiglu_bwd': d.get('swiglu_bwd'), 'fused_ways_out': d['roofline'].get('fused_ways_out')}))" | tee -a $out/${tag}_ab.jsonl
done
