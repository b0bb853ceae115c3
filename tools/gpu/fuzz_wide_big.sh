#!/bin/bash
# parity fuzz on wide models with MANY chains (the trajectory kernels with one and two tiles per workgroup, float32 and float64 models)
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-fuzz_wide_big}; mkdir -p $OUT
export FUZZ_P=33,40,64,100,128 FUZZ_BIG=0.6
for spec in "250 621 auto float32" "250 622 auto float64" "120 623 full float32" "120 624 full float64"; do
  echo "== p > 32, 60 % many-chain cases: $spec" >> $OUT/fuzz.txt
  timeout 1500 python3 tests/fuzz_parity.py $spec 2>&1 | tail -5 >> $OUT/fuzz.txt
done
cat $OUT/fuzz.txt
