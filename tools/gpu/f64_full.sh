#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-f64_full}; mkdir -p $OUT
python3 tools/f64_full_speed.py > $OUT/speed.txt 2>&1
for ub in 2 8; do
  export LOGREG_HIPCC_FLAGS="-DLR_ROWS_AHEAD64=$ub"
  timeout 600 python -m logreg_amd.build --force > $OUT/build.log 2>&1
  echo "## rows ahead $ub" >> $OUT/speed.txt
  python3 tools/f64_full_speed.py >> $OUT/speed.txt 2>&1
done
cat $OUT/speed.txt
