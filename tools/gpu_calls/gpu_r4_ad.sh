#!/bin/bash
# round 4, GPU call AD: k_chain_mfma_f64 -- its test, the float64 chain grid at many chains
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -x -k "matrix_core_kernel_of_the_float64 or planner_engine" -s > gpurun_out/r4/gpu_tests_ad.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ad.log; tail -25 gpurun_out/r4/gpu_tests_ad.log
timeout 900 python tools/chain_grid.py f64 4096 8192 10240 16384 32768 65536 > gpurun_out/r4/chain_grid_f64_ad.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_ad.txt
