#!/bin/bash
# after the last kernel change of the round: fresh-model stress of every family, the wide-model fuzz (few and many chains), the trajectory rule's measurements
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-final_checks}; mkdir -p $OUT
python3 tools/fresh_model_stress.py 60 > $OUT/stress.txt 2>&1; cat $OUT/stress.txt | cut -c1-140
for i in 1 2; do python3 tools/traj_stress.py 150 float64 64 500; done >> $OUT/stress.txt 2>&1; tail -2 $OUT/stress.txt
export FUZZ_P=33,40,64,100,128
for spec in "300 631 auto float32" "300 632 auto float64"; do echo "== p > 32: $spec" >> $OUT/fuzz.txt; timeout 900 python3 tests/fuzz_parity.py $spec 2>&1 | tail -3 >> $OUT/fuzz.txt; done
export FUZZ_BIG=0.6
for spec in "200 641 auto float32" "200 642 auto float64"; do echo "== p > 32, many chains: $spec" >> $OUT/fuzz.txt; timeout 900 python3 tests/fuzz_parity.py $spec 2>&1 | tail -3 >> $OUT/fuzz.txt; done
cat $OUT/fuzz.txt
unset FUZZ_P FUZZ_BIG
python3 tools/traj_rule_check.py > $OUT/rule.txt 2>&1; tail -40 $OUT/rule.txt | cut -c1-160
