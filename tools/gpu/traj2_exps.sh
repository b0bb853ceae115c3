#!/bin/bash
# marginal costs inside k_wide_traj2_bf16: one development build per experiment mask (results wrong by design, timing only)
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_exps}; mkdir -p $OUT
: > $OUT/exps.txt
for mask in ${2:-0 1 2 4 8 32 7 47}; do
  export LOGREG_HIPCC_FLAGS="-DLR_STAMPS -DLR_TRAJ2_EXP=$mask"
  timeout 600 python -m logreg_amd.build --force > $OUT/build_$mask.log 2>&1
  echo "## LR_TRAJ2_EXP=$mask" >> $OUT/exps.txt
  LOGREG_DEBUG_OPTS=wide_traj=2 timeout 300 python3 tools/stamps_traj.py 8192 >> $OUT/exps.txt 2>&1
done
cat $OUT/exps.txt
