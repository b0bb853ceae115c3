#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "mixed or float64 or wide or planner" > gpurun_out/r4/gpu_tests_z.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_z.log; tail -30 gpurun_out/r4/gpu_tests_z.log
