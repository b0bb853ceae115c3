#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_c}; mkdir -p $OUT
python3 tools/traj2_check.py > $OUT/check.txt 2>&1
for opt in wide_traj=1 wide_traj=2; do
  LOGREG_DEBUG_OPTS=$opt python3 tools/cfg5_whole.py 4096 8192 16384 > $OUT/$opt.txt 2>&1
done
export LOGREG_HIPCC_FLAGS=-DLR_STAMPS
timeout 600 python -m logreg_amd.build --force > $OUT/stamps_build.log 2>&1
LOGREG_DEBUG_OPTS=wide_traj=2 timeout 300 python3 tools/stamps_traj.py 8192 > $OUT/stamps_traj2.txt 2>&1
tail -3 $OUT/check.txt; for f in $OUT/wide_traj=*.txt; do echo $f; cut -c1-150 $f; done; cat $OUT/stamps_traj2.txt
