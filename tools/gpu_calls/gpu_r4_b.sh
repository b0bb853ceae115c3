#!/bin/bash
# round 4, GPU call B: full GPU suite after the prune / debug-opts refactor + float64 wide engine; config-5 prologue volume experiment
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_b.log
tail -25 gpurun_out/r4/gpu_tests_b.log
export LOGREG_HIPCC_FLAGS=-DLR_STAMPS
timeout 600 python -m logreg_amd.build --force > gpurun_out/r4/stamps_build.log 2>&1 || tail -20 gpurun_out/r4/stamps_build.log
rm -f gpurun_out/r4/cfg5_volume.txt
for e in 0 4 8; do
  echo "# LOGREG_DEBUG_EXP=$e (bit 2: one slice partial read instead of four; bit 3: no partials, no momentum)" >> gpurun_out/r4/cfg5_volume.txt
  LOGREG_DEBUG_EXP=$e timeout 300 python tools/stamps.py 5 >> gpurun_out/r4/cfg5_volume.txt 2>&1
done
grep -E "^#|launch 20|loads issued -> all|update arith|lifetime|period|row loop|entry -> addr|prologue barrier" gpurun_out/r4/cfg5_volume.txt
