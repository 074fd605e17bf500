#!/bin/bash
exec < /dev/null  # nothing here may wait on stdin (an empty $(find ...) once turned `head` into a 15-minute hang)
tag=${1:-pa}
out=$PWD/gpurun_out/${tag}
mkdir -p $out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --output-format csv -d $out/$name -o $name -- python $R/tools/prof_attn.py > $out/$name.log 2>&1; }
run stats --kernel-trace --stats
run pmc1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run pmc2 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
cd $R
python tools/prof_summarize.py $out > gpurun_out/${tag}_summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete
grep -E "attn|==" gpurun_out/${tag}_summary.txt | cut -c1-420
