#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_onepiece}; mkdir -p $OUT
LOGREG_DEBUG_OPTS=wide_traj=2 python3 tools/cfg5_whole.py 8192 --iters 12 >> $OUT/ab.txt 2>&1
python3 tools/cfg5_whole.py 8192 --iters 12 --prec full >> $OUT/ab.txt 2>&1
export LOGREG_HIPCC_FLAGS="-DLR_STAMPS -DLR_TRAJ2_EXP=4"
timeout 600 python -m logreg_amd.build --force > $OUT/build.log 2>&1
echo "## one-piece beta (LR_TRAJ2_EXP=4)" >> $OUT/ab.txt
LOGREG_DEBUG_OPTS=wide_traj=2 python3 tools/cfg5_whole.py 8192 --iters 12 >> $OUT/ab.txt 2>&1
cut -c1-330 $OUT/ab.txt
