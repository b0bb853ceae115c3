#!/bin/bash
# round 6, VERDICT r5 item 4: config 5 whole is power-bound (1300 W at 2.04 GHz) -- where do the joules go?  One development build per
# LR_TRAJ2_EXP mask (lr_stamps.h: results knowingly wrong, timing only), each timed for 5 s under a rocm-smi sampler:
#   bit 0 (1) no exp / rcp    bit 1 (2) one gradient MFMA per tile instead of 8    bit 3 (8) no L2 -> LDS DMA after the first ring fill
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-r6_cfg5_power}; mkdir -p $OUT
: > $OUT/table.txt
for mask in ${2:-prod 0 8 1 2 9 10 3 11}; do
  # (the masks live behind -DLR_STAMPS, lr_stamps.h; the unmasked row is built the same way, and the production build is timed beside it)
  if [[ $mask == prod ]]; then unset LOGREG_HIPCC_FLAGS; else export LOGREG_HIPCC_FLAGS="-DLR_STAMPS -DLR_TRAJ2_EXP=$mask"; fi
  timeout 900 python3 -m logreg_amd.build --force > $OUT/build_$mask.log 2>&1
  timeout 300 python3 tools/cfg5_power_table.py "LR_TRAJ2_EXP=$mask" >> $OUT/table.txt 2>&1
done
cat $OUT/table.txt
