#!/bin/bash
# PMC counters of the fused matrix-core kernel (args: chains mode group)
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/xp_pmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-ess --no-extra --chains $1 --mode $2 --group $3 > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
rm -rf $OUT/pmc_*/
cat $OUT/summary.txt
