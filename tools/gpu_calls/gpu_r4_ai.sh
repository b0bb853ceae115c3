#!/bin/bash
# round 4, GPU call AI: two-part plans of the float64 default policy
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "two_part or two_parts or planner_engine or float64" > gpurun_out/r4/gpu_tests_ai.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ai.log; tail -12 gpurun_out/r4/gpu_tests_ai.log
timeout 900 python tools/chain_grid.py f64 4096 4608 5120 6144 7168 8192 9216 > gpurun_out/r4/chain_grid_f64_ai.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_ai.txt
