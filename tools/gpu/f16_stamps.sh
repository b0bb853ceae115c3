#!/bin/bash
# clocks / power of the default build first, then the -DLR_STAMPS build's per-phase cycles of the two-tile trajectory kernel
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-f16_stamps}; mkdir -p $OUT
timeout 300 python3 tools/clock_sample.py > $OUT/clocks.txt 2>&1; tail -12 $OUT/clocks.txt
for o in "" wide_f16=0; do LOGREG_DEBUG_OPTS=$o python3 tools/cfg5_whole.py 8192 --iters 8 | cut -c1-160; done
export LOGREG_HIPCC_FLAGS="-DLR_STAMPS"
timeout 900 python -m logreg_amd.build --force > $OUT/stamps_build.log 2>&1; grep -E "error|wrote" $OUT/stamps_build.log | head -3
for o in wide_traj=2 wide_traj=2,wide_f16=0; do echo "== $o"; LOGREG_DEBUG_OPTS=$o timeout 300 python3 tools/stamps_traj.py 8192 2>&1 | tee -a $OUT/stamps_traj2.txt; done
