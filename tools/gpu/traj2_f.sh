#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_f}; mkdir -p $OUT
python3 tools/traj2_check.py > $OUT/check.txt 2>&1
LOGREG_DEBUG_OPTS=wide_traj=2 python3 tools/cfg5_whole.py 4096 8192 16384 > $OUT/traj2.txt 2>&1
tail -3 $OUT/check.txt; cut -c1-150 $OUT/traj2.txt
for extra in "" "-DLR_TRAJ2_DMA_SHADOW=1"; do
  export LOGREG_HIPCC_FLAGS="-DLR_STAMPS $extra"
  timeout 600 python -m logreg_amd.build --force > $OUT/stamps_build.log 2>&1
  echo "## $LOGREG_HIPCC_FLAGS" >> $OUT/stamps.txt
  LOGREG_DEBUG_OPTS=wide_traj=2 timeout 300 python3 tools/stamps_traj.py 8192 >> $OUT/stamps.txt 2>&1
done
cat $OUT/stamps.txt
