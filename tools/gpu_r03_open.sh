#!/bin/bash
exec < /dev/null
# First GPU visit of round 3: the three experiments prepared after round 2's GPU budget was spent.
#   1. tools/gemm_stagger_ab.py   staggered K start / early LDS-DMA pieces vs the product GEMM schedule (all layouts)
#   1b. TAMD_FUSE_ROPE_FWD=0/1     rotary in the q|k|v GEMM epilogue (rewritten) vs the rotary kernel, one-layer bench
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
# rotary epilogue of the q|k|v GEMM (rewritten way out) vs GEMM + rope_kernel: one-layer forward / forward+backward, interleaved
for f in 0 1 0 1; do
  TAMD_FUSE_ROPE_FWD=$f timeout 150 python tools/gpu_bench_kernels.py layer 2>/dev/null | sed "s/^/{\"rope_fwd_fused\": $f} /" >> gpurun_out/${tag}_rope_fwd_ab.txt
done
cat gpurun_out/${tag}_rope_fwd_ab.txt
timeout 200 python tools/attn_fwd64_ab.py > gpurun_out/${tag}_attn_fwd_ab.jsonl 2> gpurun_out/${tag}_attn_fwd_ab.err
cat gpurun_out/${tag}_attn_fwd_ab.jsonl
bash tools/gpu_r03_tlb.sh ${tag}_tlb
bash tools/gpu_l2.sh
# LLaVA forward: host-bound?  the same step eagerly and as a replayed HIP graph
for extra in "" "--hip-graph"; do
  timeout 240 python bench.py --config llava --steps 20 --warmup 5 --no-cpu-baseline $extra 2> gpurun_out/${tag}_llava${extra}.err | tee -a gpurun_out/${tag}_llava_graph_ab.jsonl | cut -c1-300
done
