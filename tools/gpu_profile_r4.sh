#!/bin/bash
# Round-4 profile run (on the GPU box via gpurun): ONE command -- `python3 bench.py --no-cpu-baseline --no-ess`, which
# times config 2 and then configs 3, 4, 5 -- under rocprofv3: a kernel trace and separate PMC passes.
# Outputs under gpurun_out/<name>; tools/summarize_prof.py turns them into the text committed under profiles/.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-prof_r4}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-ess > $OUT/trace.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-ess > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep "^{" $OUT/trace.log | tail -1 > $OUT/bench_line.json
# the rocpd databases are tens of MB per pass (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $OUT/trace $OUT/pmc_*/
