#!/bin/bash
# parity fuzz on the wide models (p > 32) under the default policy (the f16 interior) and in exact mode, float32 and float64 models
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-fuzz_wide}; mkdir -p $OUT
export FUZZ_P=33,40,64,100,128
for spec in "500 501 auto float32" "500 502 auto float64" "300 503 full float32" "300 504 full float64"; do
  echo "== $spec" >> $OUT/fuzz.txt
  timeout 900 python3 tests/fuzz_parity.py $spec 2>&1 | tail -6 >> $OUT/fuzz.txt
done
cat $OUT/fuzz.txt
