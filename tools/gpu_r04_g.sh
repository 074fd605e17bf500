#!/bin/bash
# Round-4 visit g: the DDP wrapper's cost at world size 1 again -- visit f had the zero-copy arm 50 ms SLOWER than the copies
# (suspect: RCCL's one-rank AVG kernel over 16 GB beside the backward).  Arms: plain, copies, zero-copy with the one-rank
# shortcut, zero-copy with the collective forced, plain again (clock drift over the visit); then a kernel trace of two
# zero-copy steps (what runs beside the GEMMs).
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04g}
run() {  # name, env, flags
  env $2 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline $3 2>> gpurun_out/${T}_ddp.err | grep -m1 '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['arm']='$1'; print(json.dumps(d))" | tee -a gpurun_out/${T}_ddp_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['arm'], {k: d.get(k) for k in ('ms_per_step','max_memory_gb','ddp_zero_copy')}, d['roofline']['achieved'])"
}
run plain "A=1" ""
run ddp-copies "A=1" "--force-ddp --no-ddp-zero-copy"
run ddp-zero-copy "A=1" "--force-ddp"
run ddp-zero-copy-collective "TAMD_DDP_WORLD1_COLLECTIVE=1" "--force-ddp"
run plain-again "A=1" ""
cd /tmp
for arm in zc:0 zccoll:1; do
  n=${arm%%:*}; v=${arm##*:}
  TAMD_DDP_WORLD1_COLLECTIVE=$v timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o $n -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --gemm-timer off --force-ddp --output-format csv > $GRAFT_REPO_ROOT/gpurun_out/${T}_prof_$n.log 2>&1
  echo "rocprof $n exit $?"; find /tmp/prof_$n -name "*.csv" | head -5
  f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/${T}_${n}_kernel_stats.csv && head -12 $f | cut -c1-160
done
tail -3 $GRAFT_REPO_ROOT/gpurun_out/${T}_ddp.err
