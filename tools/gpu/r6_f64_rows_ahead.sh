#!/bin/bash
# k_chain_f64x / the float64 lane-group kernels: how many LDS rows are fetched ahead of their use (lr_device.h LR_ROWS_AHEAD64)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-r6_f64_rows_ahead}; mkdir -p $OUT
: > $OUT/t.txt
for ahead in 4 2 1 8 3; do
  export LOGREG_HIPCC_FLAGS="-DLR_ROWS_AHEAD64=$ahead"
  timeout 900 python3 -m logreg_amd.build --force > $OUT/build_$ahead.log 2>&1
  echo "## LR_ROWS_AHEAD64=$ahead  $(grep -c scratch $OUT/build_$ahead.log) build lines mention scratch" >> $OUT/t.txt
  timeout 300 python3 tools/f64_full_check.py 2>&1 | grep "all-float64 \(4096\|8192\|2048\) chains \(lds/16\|lds/8\|auto\)" >> $OUT/t.txt
done
cat $OUT/t.txt | cut -c1-170
