#!/bin/bash
# round 4, GPU call AM: float64 at padded p = 32 on the stepwise engine only -- full suite, the float64 fuzz again (600 cases)
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_am.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_am.log; tail -12 gpurun_out/r4/gpu_tests_am.log
for args in "300 31 full float64" "300 32 auto float64"; do
  f=gpurun_out/r4/fuzz4_$(echo $args | tr ' ' '_').log
  timeout 1500 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep -A16 "FAIL" $f | head -40; tail -1 $f
done
