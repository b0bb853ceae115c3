#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_l.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_l.log; tail -6 gpurun_out/r4/gpu_tests_l.log
timeout 600 python tools/chain_grid.py 4096 4608 5120 6144 8192 9216 10240 13312 14336 16384 > gpurun_out/r4/chain_grid_l.txt 2>&1; cat gpurun_out/r4/chain_grid_l.txt
timeout 300 python tools/planner_bench.py 200,8,5120,hmc,full 200,8,5120,mala,auto 200,8,9216,mala,auto > gpurun_out/r4/planner_bench_l.txt 2>&1; cat gpurun_out/r4/planner_bench_l.txt
