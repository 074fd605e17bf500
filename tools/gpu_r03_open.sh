#!/bin/bash
exec < /dev/null
# First GPU visit of round 3: the three experiments prepared after round 2's GPU budget was spent.
#   1. tools/gemm_stagger_ab.py   staggered K start / early LDS-DMA pieces vs the product GEMM schedule (all layouts)
#   2. tools/attn_fwd64_ab.py     causal forward with two query tiles per workgroup (and the 64-rows-per-wave kernel)
#   3. tools/gpu_r03_tlb.sh       operand placement / address translation and the fused SwiGLU-backward GEMM regime
# usage: gpurun --timeout 1200 -- bash tools/gpu_r03_open.sh [tag]
tag=${1:-r03a}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/gemm_stagger_ab.py > gpurun_out/${tag}_gemm_stagger_ab.jsonl 2> gpurun_out/${tag}_gemm_stagger_ab.err
echo "stagger exit $?"
python - <<PY
import json
for line in open("gpurun_out/${tag}_gemm_stagger_ab.jsonl"):
    r = json.loads(line)
    med = {c: sorted(v)[len(v) // 2] for c, v in r["tflops"].items()}
    top = sorted(med, key=med.get, reverse=True)[:4]
    print(r["shape"], r["layout"], "off", med["off"], " | ", "  ".join(f"{c} {med[c]}" for c in top))
PY
timeout 200 python tools/attn_fwd64_ab.py > gpurun_out/${tag}_attn_fwd_ab.jsonl 2> gpurun_out/${tag}_attn_fwd_ab.err
cat gpurun_out/${tag}_attn_fwd_ab.jsonl
bash tools/gpu_r03_tlb.sh ${tag}_tlb
bash tools/gpu_l2.sh
