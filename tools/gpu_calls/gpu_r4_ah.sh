#!/bin/bash
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_ah.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ah.log; tail -30 gpurun_out/r4/gpu_tests_ah.log
