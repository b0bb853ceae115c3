#!/bin/bash
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_aj.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_aj.log; tail -12 gpurun_out/r4/gpu_tests_aj.log
timeout 900 python tools/chain_grid.py f64 8192 9216 16384 18432 20480 24576 > gpurun_out/r4/chain_grid_f64_aj.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_aj.txt
