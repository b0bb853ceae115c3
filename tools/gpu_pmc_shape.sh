#!/bin/bash
# PMC counters of one shape through tools/run_shape.py (args as run_shape.py); summary on stdout
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_shape
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/tools/run_shape.py "$@" > $OUT/trace.log 2>&1
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o bench -- python3 $ROOT/tools/run_shape.py "$@" > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
rm -rf $OUT/pmc_*/ $OUT/trace
grep -v "^#" $OUT/summary.txt | grep "k_chain_mfma\|calls\|pmc pass" | cut -c1-150
