#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_order}; mkdir -p $OUT
LOGREG_DEBUG_OPTS=wide_traj=1 python3 tools/cfg5_whole.py 8192 --iters 8 >> $OUT/ab.txt 2>&1
LOGREG_DEBUG_OPTS=wide_traj=2 python3 tools/cfg5_whole.py 8192 --iters 8 >> $OUT/ab.txt 2>&1
for o in 1 2 3; do
  export LOGREG_HIPCC_FLAGS="-DLR_TRAJ2_ORDER=$o"
  timeout 600 python -m logreg_amd.build --force > $OUT/build.log 2>&1; tail -2 $OUT/build.log | head -1 >> $OUT/ab.txt
  echo "## order $o" >> $OUT/ab.txt
  LOGREG_DEBUG_OPTS=wide_traj=2 python3 tools/cfg5_whole.py 8192 --iters 8 >> $OUT/ab.txt 2>&1
done
cut -c1-120 $OUT/ab.txt
