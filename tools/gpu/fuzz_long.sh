#!/bin/bash
# a longer parity-fuzz campaign at the round's final library (all widths; both policies; both dtypes)
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-fuzz_long}; mkdir -p $OUT
for spec in "1500 701 auto float32" "1500 702 auto float64" "1200 703 full float32" "1200 704 full float64"; do
  echo "== all widths: $spec" >> $OUT/fuzz.txt
  timeout 2400 python3 tests/fuzz_parity.py $spec 2>&1 | grep -v "^SKIP" | tail -4 >> $OUT/fuzz.txt
done
cat $OUT/fuzz.txt
