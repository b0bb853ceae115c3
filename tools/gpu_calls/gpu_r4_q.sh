#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stats.py -m gpu -q -x -k "two_part or planner_engine" > gpurun_out/r4/gpu_tests_q.log 2>&1; tail -8 gpurun_out/r4/gpu_tests_q.log
timeout 600 python tools/chain_grid.py 4096 4608 5120 6144 8192 9216 10240 16384 18432 20480 24576 > gpurun_out/r4/chain_grid_q.txt 2>&1; cat gpurun_out/r4/chain_grid_q.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_q_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_q_full.log; tail -5 gpurun_out/r4/gpu_tests_q_full.log
