#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-suite_b}; mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
timeout 600 python3 tools/f64_p32_speed.py > $OUT/f64_p32_speed.txt 2>&1; cat $OUT/f64_p32_speed.txt
for spec in "600 501 full float64" "600 502 full float32" "400 503 auto float64" "400 504 auto float32"; do
  FUZZ_P="9,12,16,17,20,24,31,32" timeout 1200 python3 tests/fuzz_parity.py $spec > $OUT/fuzz_$(echo $spec | tr ' ' '_').txt 2>&1
  echo "fuzz $spec: $(tail -1 $OUT/fuzz_$(echo $spec | tr ' ' '_').txt)"
  grep -c "^FAIL" $OUT/fuzz_$(echo $spec | tr ' ' '_').txt
done
