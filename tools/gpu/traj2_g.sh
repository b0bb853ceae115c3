#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_g}; mkdir -p $OUT
for rep in 1 2; do
for opt in wide_traj=1 wide_traj=2; do
  LOGREG_DEBUG_OPTS=$opt python3 tools/cfg5_whole.py 8192 --iters 8 >> $OUT/ab.txt 2>&1
done
done
cut -c1-120 $OUT/ab.txt

