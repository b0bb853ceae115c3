#!/bin/bash
# round 4, GPU call A: full GPU suite, kernarg-size probe, config-5 prologue volume experiment (stamps build)
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gpu_tests_a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_a.log
tail -15 gpurun_out/r4/gpu_tests_a.log
tools/bin/ka; echo "5 KB kernarg struct launch rc=$?" | tee gpurun_out/r4/kernarg_probe.txt
export LOGREG_HIPCC_FLAGS=-DLR_STAMPS
timeout 600 python -m logreg_amd.build --force > gpurun_out/r4/stamps_build.log 2>&1
for e in 0 4 8; do
  echo "# LOGREG_DEBUG_EXP=$e (bit 2: one slice partial read instead of four; bit 3: no partials, no momentum)" >> gpurun_out/r4/cfg5_volume.txt
  LOGREG_DEBUG_EXP=$e timeout 300 python tools/stamps.py 5 >> gpurun_out/r4/cfg5_volume.txt 2>&1
done
tail -60 gpurun_out/r4/cfg5_volume.txt
