#!/bin/bash
# ISA of single GEMM kernel instantiations (seconds instead of the minutes the whole file takes):
#   tools/gemm_isa.sh "gemm_fr_kernel<tamd::bf16_t, false, false, TAMD_EPI_NONE, TAMD_ACT_NONE>" [more ...]
# writes /tmp/gemm_isa/k<i>.s and prints registers / spills / scratch of each.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/gemm_isa
src=/tmp/gemm_isa/quick.hip
{
  echo '#define TAMD_GEMM_KERNELS_ONLY 1'
  echo "#include \"$R/transformers_amd/csrc/gemm.hip\""
  for a in "$@"; do echo "template __global__ void tamd::$a(tamd::GemmArgs);"; done
} > $src
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I $R/transformers_amd/csrc -I $R/include -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage $src -o /tmp/gemm_isa/quick.s 2> /tmp/gemm_isa/quick.rpass || { grep -m5 error /tmp/gemm_isa/quick.rpass; exit 1; }
grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill" /tmp/gemm_isa/quick.rpass | sed 's/.*remark: *//; s/ \[-Rpass.*//'
