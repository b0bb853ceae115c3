#!/bin/bash
# round 6, after the last kernel change: the oracle parity fuzz at every width (both policies, both dtypes, narrow widths weighted towards
# float64 p <= 8 where this round's kernel lives), the build-differential fuzz, the fresh-model stress campaign
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r6_final}; mkdir -p $OUT
: > $OUT/fuzz.txt
for spec in "700 701 auto float32" "700 702 auto float64" "600 703 full float32" "900 704 full float64"; do
  echo "== all widths: $spec" >> $OUT/fuzz.txt
  timeout 1500 python3 tests/fuzz_parity.py $spec 2>&1 | tail -4 >> $OUT/fuzz.txt
done
export FUZZ_P=5,6,7,8
for spec in "900 711 full float64" "400 712 auto float64"; do
  echo "== 4 < p <= 8 (k_chain_f64x under 'full'): $spec" >> $OUT/fuzz.txt
  timeout 1500 python3 tests/fuzz_parity.py $spec 2>&1 | tail -4 >> $OUT/fuzz.txt
done
export FUZZ_P=33,40,64,100,128
for spec in "300 721 auto float32" "300 722 auto float64"; do
  echo "== p > 32: $spec" >> $OUT/fuzz.txt
  timeout 900 python3 tests/fuzz_parity.py $spec 2>&1 | tail -4 >> $OUT/fuzz.txt
done
unset FUZZ_P
for seed in 801 802 803 804; do
  echo "== two builds, seed $seed" >> $OUT/fuzz.txt
  timeout 1500 python3 tests/fuzz_builds.py 2000 $seed 2>&1 | tail -3 >> $OUT/fuzz.txt
done
python3 tools/fresh_model_stress.py 60 > $OUT/stress.txt 2>&1
cat $OUT/fuzz.txt | cut -c1-260; cat $OUT/stress.txt | cut -c1-150
