#!/bin/bash
exec < /dev/null
# Round 3, GPU visit 2: the persistent / XCD-aligned GEMM walk and the piece placements against the product schedule
# (TFLOP/s, fabric traffic and L2 hits per variant), their effect on one decoder layer and on the step, the DDP
# CU-sharing experiment, bert-base with the masked-LM head on the kernels, the new parity tests.
# usage: gpurun --timeout 1500 -- bash tools/gpu_r03_b.sh [tag]
tag=${1:-r03b}
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 python tools/gemm_persist_ab.py > $out/${tag}_gemm_persist_ab.jsonl 2> $out/${tag}_gemm_persist_ab.err
echo "persist ab exit $?"; tail -2 $out/${tag}_gemm_persist_ab.err
python - <<PY
import json
for line in open("gpurun_out/${tag}_gemm_persist_ab.jsonl"):
    r = json.loads(line)
    med = {c: sorted(x for x in v if not isinstance(x, str))[len(v) // 2] for c, v in r["tflops"].items()}
    bad = [c for c, v in r["tflops"].items() if "MISMATCH" in v]
    print(r["shape"], r["leg"], "splitK" if r["split_k"] else "", " ".join(f"{c} {m}" for c, m in med.items()), "MISMATCH " + str(bad) if bad else "")
PY
cd /tmp
for shp in "gate_up dW,dX" "down fwd,dW"; do
  set -- $shp
  for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$out/$tag/pmc_$1_$(echo $pmc | cut -c1-5)
    timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $d -o p -- python $R/tools/gemm_persist_ab.py --rounds 1 --iters 2 --no-diag --shapes $1 --legs $2 > $d.log 2>&1
    echo "== $1 $2 : $pmc" >> $out/${tag}_gemm_persist_pmc.txt
    python $R/tools/pmc_per_kernel.py $d --match gemm_fl --ratio TCC_HIT_sum/TCC_MISS_sum >> $out/${tag}_gemm_persist_pmc.txt
  done
done
cd $R
cat $out/${tag}_gemm_persist_pmc.txt | cut -c1-220
for pm in 0 1 2 0 1 2; do
  TAMD_GEMM_PERSIST=$pm timeout 150 python tools/gpu_bench_kernels.py layer 2>/dev/null | sed "s/^/{\"gemm_persist\": $pm} /" >> $out/${tag}_layer_persist_ab.txt
done
cut -c1-200 $out/${tag}_layer_persist_ab.txt
for pm in 0 1 2; do
  TAMD_GEMM_PERSIST=$pm timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2> $out/${tag}_bench_p$pm.err | tee $out/${tag}_bench_p$pm.json | cut -c1-260
done
timeout 300 python tools/ddp_interference.py > $out/${tag}_ddp_interference.jsonl 2> $out/${tag}_ddp_interference.err
cat $out/${tag}_ddp_interference.jsonl; tail -3 $out/${tag}_ddp_interference.err
timeout 300 python bench.py --config bert-base --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_bert.json 2> $out/${tag}_bench_bert.err
echo "bert exit $?"; cut -c1-400 $out/${tag}_bench_bert.json; tail -5 $out/${tag}_bench_bert.err
timeout 400 python bench.py --config bert-base --steps 20 --warmup 5 > $out/${tag}_bench_bert_cpu.json 2> $out/${tag}_bench_bert_cpu.err
echo "bert+cpu exit $?"; python -c "
import json
try:
    d = json.load(open('gpurun_out/${tag}_bench_bert_cpu.json')); print(d['ms_per_step'], d.get('cpu_baseline'))
except Exception as e: print('no json', e)"; tail -3 $out/${tag}_bench_bert_cpu.err
timeout 500 python -m pytest tests/test_kernels.py tests/test_torch_ops.py tests/test_models.py -m gpu -q -x --timeout 400 -k "gemm or cross_entropy or padded or bert or opcheck or fused_lm" > $out/${tag}_tests.log 2>&1
tail -4 $out/${tag}_tests.log
find $out/$tag -name "*.csv" -size +3M -delete
