#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj2_bf16}; mkdir -p $OUT
for prec in auto bf16; do python3 tools/cfg5_whole.py 4096 8192 16384 --iters 8 --prec $prec >> $OUT/ab.txt 2>&1; done
cut -c1-330 $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "config5" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
