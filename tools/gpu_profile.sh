#!/bin/bash
# rocprofv3 passes of ONE command on the GPU box (via gpurun): a kernel trace and separate PMC passes (never combined).
#   tools/gpu_profile.sh <outdir-name> [--pmc-only|--trace-only] -- python3 <script> [args...]
# default command: python3 bench.py --no-cpu-baseline --no-ess   (the round's profile of record: tools/refresh_profiles.py <tag>)
# Writes gpurun_out/<outdir-name>/summary.txt (tools/summarize_prof.py) and bench_line.json when the command printed a JSON line;
# the rocpd databases (tens of MB per pass; gpurun merges at most 64 MiB back) are deleted.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
NAME=${1:-prof}; shift || true
WHAT=all
case "${1:-}" in --pmc-only) WHAT=pmc; shift;; --trace-only) WHAT=trace; shift;; esac
[ "${1:-}" = "--" ] && shift
if [ $# -eq 0 ]; then set -- python3 $ROOT/bench.py --no-cpu-baseline --no-ess; fi
OUT=$ROOT/gpurun_out/$NAME
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
echo "$*" > $OUT/command.txt
if [ $WHAT != pmc ]; then
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- "$@" > $OUT/trace.log 2>&1
fi
if [ $WHAT != trace ]; then
  for pmc in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
    timeout 900 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o bench -- "$@" > $OUT/pmc_$name.log 2>&1
  done
fi
cd $ROOT
python3 tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -h "^{" $OUT/trace.log 2>/dev/null | tail -1 > $OUT/bench_line.json
rm -rf $OUT/trace $OUT/pmc_*/
