#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2}; mkdir -p $OUT
python3 tools/traj2_check.py > $OUT/check.txt 2>&1
for opt in wide_traj=1 wide_traj=2; do
  LOGREG_DEBUG_OPTS=$opt python3 tools/cfg5_whole.py 2048 4096 6144 8192 16384 > $OUT/$opt.txt 2>&1
done
cat $OUT/check.txt; for f in $OUT/wide_traj=*.txt; do echo $f; cut -c1-200 $f; done
