#!/bin/bash
exec < /dev/null
# ONE parameterised GPU visit (replaces the per-visit gpu_r0N_x.sh scripts of rounds 2-4):
#     gpurun --timeout 900 -- bash tools/gpu_visit.sh <tag> <step> [<step> ...]
# Every step writes gpurun_out/<tag>_<what>; copy what is to be judged into profiles/ afterwards.  Steps:
#   tests [pytest -k expr]   the -m gpu suite                      smoke        __graft_entry__.smoke()
#   bench                    the default bench line (CPU baseline)  bench_fast   ... without the CPU baseline leg
#   prof                     rocprofv3 --kernel-trace --stats of the bench command + FETCH_SIZE / WRITE_SIZE passes -> traffic json
#   kernels [which...]       tools/gpu_bench_kernels.py (gemm attn hbm layer)
#   gemm_pmc                 tools/gemm_vs_vendor_pmc.py under one rocprofv3 --pmc pass per counter set (ours vs hipBLASLt)
#   gemm_ab / attn_ab        this tree's library against tools/ab/libtamd_base.so (tools/build_base_lib.sh), interleaved
#   attn_variants            ... and every tools/ab/libtamd_[p-z]*.so (tools/build_variant.py), attention entry points
#   attn_pmc                 rocprofv3 --pmc passes over the attention kernels of this tree (tools/attn_pmc.py)
#   attn_prof                per-kernel times of the attention kernels at the Llama-3-8B shape (rocprofv3)
#   bert / bert_graph / llava   the other BASELINE configurations' bench lines      prof_cfg <config>   their rocprofv3 kernel stats
#   train                    bench.py --train-step (fwd+bwd+clip+AdamW), TamdAdamW then torch's clip + fused AdamW
#   ddp                      bench.py --force-ddp over RCCL at world size 1: zero-copy on / off / collective forced / --verify-ddp
#   ddp2                     (boxes with >= 2 GPUs only) bench.py --gpus 2 under torchrun, both gradient hand-overs
#   py <script> [args]       any tools/ script, stdout to <tag>_<script>.jsonl
tag=${1:?tag}
shift
R=$PWD
out=$R/gpurun_out
mkdir -p $out/$tag
export PYTHONUNBUFFERED=1 TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
round=${tag:0:3}

step_tests() {
  timeout 560 python -m pytest tests -m gpu -q --timeout 400 ${1:+-k "$1"} > $out/${tag}_tests.log 2>&1
  echo "tests exit $?" >> $out/${tag}_tests.log
  cp $out/parity_hip.json $out/${tag}_parity_hip.json 2>/dev/null
  tail -6 $out/${tag}_tests.log
}
step_smoke() {
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1
  echo "smoke exit $?" >> $out/${tag}_smoke.log
  tail -3 $out/${tag}_smoke.log
}
step_bench() {
  timeout 900 python bench.py "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
  echo "bench exit $?"; cut -c1-400 $out/${tag}_bench.json; tail -2 $out/${tag}_bench.err
  python - <<PY
import json
try:
    d = json.load(open("$out/${tag}_bench.json"))
    print("ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"], "layer_forward", d.get("layer_forward"))
    print("secondary", json.dumps(d.get("secondary"))[:1500])
except Exception as e:
    print("bench line unreadable:", e)
PY
}
step_bench_fast() { step_bench --no-cpu-baseline "$@"; }
step_prof() {
  cd /tmp
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_prof_bench.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/$tag/fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/$tag/write -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
  cd $R
  python tools/prof_traffic.py $out/$tag $out/${tag} $round > $out/${tag}_traffic.log 2>&1
  cp $(find $out/$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv 2>/dev/null
  head -16 $out/${tag}_bench_kernel_stats.csv | cut -c1-160
  head -8 $out/${tag}_traffic.log
}
step_kernels() {
  timeout 400 python tools/gpu_bench_kernels.py "$@" > $out/${tag}_kernel_microbench.jsonl 2> $out/${tag}_kernel_microbench.err
  cut -c1-260 $out/${tag}_kernel_microbench.jsonl; tail -2 $out/${tag}_kernel_microbench.err
}
step_gemm_pmc() {
  local i=0
  mkdir -p $out/$tag/gemm_pmc
  cd /tmp
  python $R/tools/gemm_vs_vendor_pmc.py passes | while read -r set; do
    i=$((i + 1))
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag/gemm_pmc/pass$i -o p -- python $R/tools/gemm_vs_vendor_pmc.py run > $out/$tag/gemm_pmc/pass$i.log 2>&1
    echo "pass $i exit $?: $set"
  done
  cd $R
  find $out/$tag/gemm_pmc -name "*.csv" -size +4M -delete
  python tools/gemm_vs_vendor_pmc.py table $out/$tag/gemm_pmc $out/${tag}_gemm_vs_hipblaslt_pmc.md | head -60
}
step_gemm_ab() {
  timeout 300 python tools/gemm_lib_ab.py > $out/${tag}_gemm_lib_ab.jsonl 2> $out/${tag}_gemm_lib_ab.err
  cut -c1-260 $out/${tag}_gemm_lib_ab.jsonl; tail -2 $out/${tag}_gemm_lib_ab.err
}
step_attn_ab() {
  timeout 300 python tools/attn_lib_ab.py > $out/${tag}_attn_lib_ab.jsonl 2> $out/${tag}_attn_lib_ab.err
  cut -c1-300 $out/${tag}_attn_lib_ab.jsonl; tail -2 $out/${tag}_attn_lib_ab.err
}
step_attn_variants() {  # tools/ab/libtamd_base.so, this tree's library and every tools/ab/libtamd_[p-z]*.so (tools/build_variant.py)
  AB_SHAPES=${AB_SHAPES:-llama3-8b,bidir-128} timeout 500 python tools/attn_lib_ab.py tools/ab/libtamd_base.so transformers_amd/libtamd.so $(ls tools/ab/libtamd_[p-z]*.so 2>/dev/null) > $out/${tag}_attn_variants_ab.jsonl 2> $out/${tag}_attn_variants_ab.err
  python - <<PY
import json
for l in open("$out/${tag}_attn_variants_ab.jsonl"):
    r = json.loads(l)
    print(r["shape"])
    for k, v in r.items():
        if isinstance(v, dict) and "fwd_ms" in v:
            print(f"  {k:6s} fwd {v['fwd_ms']:.4f} ms {v['fwd_TF']:5d} TF | bwd {v['bwd_ms']:.4f} ms {v['bwd_TF']:5d} TF | err {r.get(k + '_err')} same {r.get(k + '_same_bits')}")
PY
  tail -3 $out/${tag}_attn_variants_ab.err
}
step_attn_pmc() {  # counters of the three attention kernels of THIS tree's library (one rocprofv3 --pmc pass per counter set)
  local i=0
  mkdir -p $out/$tag/attn_pmc
  cd /tmp
  python $R/tools/attn_pmc.py passes | while read -r set; do
    i=$((i + 1))
    AB_LIBS=new AB_SHAPES=llama3-8b AB_NOREF=1 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag/attn_pmc/pass$i -o p -- python $R/tools/attn_lib_ab.py > $out/$tag/attn_pmc/pass$i.log 2>&1
    echo "pass $i exit $?: $set"
  done
  cd $R
  find $out/$tag/attn_pmc -name "*.csv" -size +4M -delete
  python tools/attn_pmc.py table $out/$tag/attn_pmc $out/${tag}_attn_pmc.md | tail -40
}
step_attn_prof() {
  ( cd /tmp && AB_LIBS=new AB_SHAPES=llama3-8b timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/attn -o ab -- python $R/tools/attn_lib_ab.py > /dev/null 2>&1 )
  cp $(find $out/$tag/attn -name "*kernel_stats.csv" | head -1) $out/${tag}_attn_kernel_stats.csv 2>/dev/null
  head -6 $out/${tag}_attn_kernel_stats.csv | cut -c1-200
}
bench_line() {  # <file suffix> <bench.py args...>
  local name=$1
  shift
  timeout 300 python bench.py "$@" 2> $out/${tag}_bench_${name}.err | grep -m1 '^{"metric' | tee -a $out/${tag}_bench_${name}.jsonl | cut -c1-330
}
step_prof_cfg() {  # rocprofv3 kernel stats of one of the other BASELINE configurations: prof_cfg bb | lv  (bert-base | llava)
  local cfg=$1
  [ "$cfg" = lv ] && cfg=llava
  [ "$cfg" = bb ] && cfg=bert-base
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag/stats_$cfg -o bench -- python $R/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --gemm-timer off > $out/${tag}_prof_${cfg}.log 2>&1 )
  cp $(find $out/$tag/stats_$cfg -name "*kernel_stats.csv" | head -1) $out/${tag}_${cfg}_kernel_stats.csv 2>/dev/null
  head -14 $out/${tag}_${cfg}_kernel_stats.csv | cut -c1-170
}
step_bert() { bench_line bert --config bert-base --steps 20 --warmup 5 "$@"; }
step_bert_graph() { bench_line bert_graph --config bert-base --steps 20 --warmup 5 --hip-graph --no-cpu-baseline; }
step_llava() { bench_line llava --config llava --steps 20 --warmup 5; }
step_train() {  # the whole optimizer step of Trainer: ours, then torch's clip_grad_norm_ + fused AdamW
  bench_line train --train-step --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
  bench_line train --train-step --optimizer torch --steps 3 --warmup 2 --no-cpu-baseline --no-secondary
}
step_ddp() {
  for arm in "" "--no-ddp-zero-copy" "--verify-ddp"; do
    bench_line ddp --force-ddp --steps 4 --warmup 3 --no-cpu-baseline $arm
  done
  TAMD_DDP_WORLD1_COLLECTIVE=1 bench_line ddp --force-ddp --steps 4 --warmup 3 --no-cpu-baseline --verify-ddp
}
step_ddp2() {
  n=$(python -c "import torch; print(torch.cuda.device_count())")
  if [ "$n" -ge 2 ]; then
    for arm in "--verify-ddp" "--no-ddp-zero-copy"; do
      timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline $arm 2>> $out/${tag}_bench_n2.err | grep -m1 '^{"metric' >> $out/${tag}_bench_n2.jsonl
    done
    cut -c1-300 $out/${tag}_bench_n2.jsonl
  else
    echo "ddp2: $n GPU visible, skipped"
  fi
}
step_py() {
  local s=$1
  shift
  timeout 400 python tools/$s "$@" > $out/${tag}_$(basename $s .py).jsonl 2> $out/${tag}_$(basename $s .py).err
  cut -c1-300 $out/${tag}_$(basename $s .py).jsonl | tail -40; tail -3 $out/${tag}_$(basename $s .py).err
}

while [ $# -gt 0 ]; do
  s=$1
  shift
  args=()
  while [ $# -gt 0 ] && ! declare -F "step_$1" > /dev/null; do
    args+=("$1")
    shift
  done
  echo "=== $s ${args[*]}"
  t0=$(date +%s)
  "step_$s" "${args[@]}"
  echo "=== $s done in $(( $(date +%s) - t0 )) s"
done
find $out/$tag -name "*.csv" -size +4M -delete 2>/dev/null
find $out/$tag -name "*.db" -delete 2>/dev/null
exit 0
