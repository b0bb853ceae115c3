#!/bin/bash
# the parity fuzz at the round's final library: all widths and the wide models, both policies, both dtypes
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-fuzz_final}; mkdir -p $OUT
for spec in "700 601 auto float32" "700 602 auto float64" "500 603 full float32" "500 604 full float64"; do
  echo "== all widths: $spec" >> $OUT/fuzz.txt
  timeout 1500 python3 tests/fuzz_parity.py $spec 2>&1 | tail -4 >> $OUT/fuzz.txt
done
export FUZZ_P=33,40,64,100,128
for spec in "400 611 auto float32" "400 612 auto float64"; do
  echo "== p > 32: $spec" >> $OUT/fuzz.txt
  timeout 900 python3 tests/fuzz_parity.py $spec 2>&1 | tail -4 >> $OUT/fuzz.txt
done
cat $OUT/fuzz.txt
