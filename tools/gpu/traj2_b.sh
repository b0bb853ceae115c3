#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_b}; mkdir -p $OUT


export LOGREG_HIPCC_FLAGS=-DLR_STAMPS
timeout 600 python -m logreg_amd.build --force > $OUT/stamps_build.log 2>&1
LOGREG_DEBUG_OPTS=wide_traj=2 timeout 300 python3 tools/stamps_traj.py 8192 > $OUT/stamps_traj2.txt 2>&1
LOGREG_DEBUG_OPTS=wide_traj=2 timeout 300 python3 tools/stamps_traj.py 4096 >> $OUT/stamps_traj2.txt 2>&1
cat $OUT/stamps_traj2.txt
