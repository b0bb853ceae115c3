#!/bin/bash
# round 4, GPU call X: tall float64 models under the default policy (bf16 interior) -- stepwise tests, config 4's shape in both dtypes
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "stepwise or tall or planner or fullsize" > gpurun_out/r4/gpu_tests_x.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_x.log; tail -12 gpurun_out/r4/gpu_tests_x.log
timeout 900 python tools/f64_tall_probe.py > gpurun_out/r4/f64_tall_probe.txt 2>&1; cat gpurun_out/r4/f64_tall_probe.txt
