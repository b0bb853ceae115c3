#!/bin/bash
# round 6: a longer campaign at the final library -- 20 000 build-differential cases, 6 000 oracle-fuzz cases, 3 x 60-model fresh-model stress
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-r6_long}; mkdir -p $OUT
: > $OUT/long.txt
for seed in 901 902 903 904; do
  echo "== two builds, seed $seed" >> $OUT/long.txt
  timeout 1500 python3 tests/fuzz_builds.py 5000 $seed 2>&1 | tail -2 >> $OUT/long.txt
done
for spec in "1500 911 auto float32" "1500 912 auto float64" "1500 913 full float32" "1500 914 full float64"; do
  echo "== oracle fuzz: $spec" >> $OUT/long.txt
  timeout 1700 python3 tests/fuzz_parity.py $spec 2>&1 | grep -v "^SKIP" | tail -3 >> $OUT/long.txt
done
for i in 1 2 3; do python3 tools/fresh_model_stress.py 60 2>&1 | grep -c " 0 of 120 runs differ" >> $OUT/long.txt; done
cat $OUT/long.txt | cut -c1-200
