#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "tall or stepwise or float64 or fullsize or sixteen" > gpurun_out/r4/gpu_tests_ab.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ab.log; tail -8 gpurun_out/r4/gpu_tests_ab.log
timeout 900 python tools/f64_tall_probe.py 2>&1
