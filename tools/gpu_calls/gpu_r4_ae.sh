#!/bin/bash
# round 4, GPU call AE: the float32-interior kernel on 32 / 64 lanes per chain (few chains); float64 planner timing
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "float64 or mixed or planner_engine" -s > gpurun_out/r4/gpu_tests_ae.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ae.log; grep -v "^HMC z\|^$" gpurun_out/r4/gpu_tests_ae.log | tail -25
timeout 900 python tools/chain_grid.py f64 1 64 256 1024 2048 4096 > gpurun_out/r4/chain_grid_f64_ae.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_ae.txt
