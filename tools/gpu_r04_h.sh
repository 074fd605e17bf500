#!/bin/bash
# Round-4 visit h: the grouped weight-gradient launch (tamd_gemm_group) -- tests, micro A/B, bert-base step A/B with its kernel
# stats; the Llama step after the dW kernel lost its 4 spilled registers (segmented-output epilogue).
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${TAG:-r04h}
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels.py tests/test_models.py -q -m gpu -x -k "gemm_group or segmented or split_k or norm or bias_act or colsum or bert" > gpurun_out/${T}_tests.log 2>&1
echo "tests exit $?"; tail -3 gpurun_out/${T}_tests.log
timeout 200 python tools/gemm_group_ab.py > gpurun_out/${T}_gemm_group_ab.jsonl 2> gpurun_out/${T}_gemm_group_ab.err; cut -c1-330 gpurun_out/${T}_gemm_group_ab.jsonl
for v in 0 1 0 1; do
  TAMD_BERT_GROUP_DW=$v timeout 150 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline 2>> gpurun_out/${T}_bench_bert.err | grep -m1 '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['group_dw']=$v; print(json.dumps(d))" | tee -a gpurun_out/${T}_bench_bert_ab.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('group_dw', d['group_dw'], d['ms_per_step'], (d.get('roofline') or {}).get('achieved'))"
done
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$T -o bert -- python $R/bench.py --config bert-base --steps 6 --warmup 2 --no-cpu-baseline --gemm-timer off > /dev/null 2>&1 )
f=$(find /tmp/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${T}_bert_kernel_stats.csv && head -16 $f | cut -c1-150
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline 2>> gpurun_out/${T}_bench.err | grep -m1 '^{"metric' | tee gpurun_out/${T}_bench_llama.json | cut -c1-300
