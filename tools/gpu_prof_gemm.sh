#!/bin/bash
exec < /dev/null  # nothing here may wait on stdin (an empty $(find ...) once turned `head` into a 15-minute hang)
tag=${1:-p1}
out=$PWD/gpurun_out/${tag}
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $out/$name -o $name -- python $R/tools/prof_gemm.py 32768 4096 4096 nt > $out/$name.log 2>&1; }
run stats --kernel-trace --stats
run pmc1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run pmc2 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run pmc3 --pmc TCC_HIT_sum TCC_MISS_sum
run pmc4 --pmc FETCH_SIZE
run pmc5 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum
cd $R
python tools/prof_summarize.py $out > gpurun_out/${tag}_summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete
find $out -name "*.db" -delete
cat gpurun_out/${tag}_summary.txt | cut -c1-400
