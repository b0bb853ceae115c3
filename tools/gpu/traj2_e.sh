#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_e}; mkdir -p $OUT
for opt in wide_traj=0 wide_traj=2; do
  LOGREG_DEBUG_OPTS=$opt python3 tools/cfg5_whole.py 12288 16384 32768 65536 > $OUT/$opt.txt 2>&1
done
python3 tools/cfg5_whole.py 1024 4096 4112 6144 8192 > $OUT/auto.txt 2>&1
for f in $OUT/*.txt; do echo $f; cut -c1-175 $f; done
timeout 900 python -m pytest tests -m gpu -x -q -k "wide or traj or config5 or cfg5 or fullsize" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
