#!/bin/bash
exec < /dev/null
# Round-3 opener for the open question of profiles/r02_regression_note.md: is the slow regime of the fused
# SwiGLU-backward GEMM an address-translation (placement) effect?
#   1. tools/swiglu_bwd_placement.py: timing legs in one process (fresh / behind 120 GB / in holes / rotating sets)
#   2. UTCL1 (per-CU TLB) request / miss counters of the same legs
#   3. the same counters for the kernels of the 32-layer model with the fused epilogue on
# usage: gpurun --timeout 900 -- bash tools/gpu_r03_tlb.sh [tag]
tag=${1:-r03tlb}
R=$PWD
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 240 python tools/swiglu_bwd_placement.py > $out/placement.jsonl 2> $out/placement.err
cat $out/placement.jsonl
cd /tmp
PMC="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
timeout 240 rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_legs -o legs -- \
    python $R/tools/swiglu_bwd_placement.py --iters 4 --legs fresh,ballast,rotate > $out/pmc_legs.log 2>&1
TAMD_FUSE_SWIGLU_BWD=1 timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $out/pmc_model -o model -- \
    python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/pmc_model.log 2>&1
cd $R
R1="TCP_UTCL1_TRANSLATION_MISS_sum/TCP_UTCL1_REQUEST_sum"
python tools/pmc_per_kernel.py $out/pmc_legs --match tamd:: --ratio $R1 --out $out/tlb_legs.txt
python tools/pmc_per_kernel.py $out/pmc_model --match tamd:: --ratio $R1 --out $out/tlb_model.txt
find $out -name "*.csv" -size +2M -delete
