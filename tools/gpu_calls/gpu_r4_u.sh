#!/bin/bash
# round 4, GPU call U: float64 row term with one exponential and the fdlibm log kernel -- f64 tests, f64 chain grid, f64 planner bench
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -x -k "float64 or f64 or closures or variant or posterior_matches or planner or free_running or shapes or hessian or map" > gpurun_out/r4/gpu_tests_u.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_u.log; tail -8 gpurun_out/r4/gpu_tests_u.log
timeout 600 python tools/chain_grid.py f64 1024 4096 8192 > gpurun_out/r4/chain_grid_f64_u.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_u.txt
PLANNER_BENCH_DTYPE=float64 timeout 900 python tools/planner_bench.py 200,8,4096,mala,auto 200,8,8192,mala,auto 200,8,8192,rwmh,auto 200,8,8192,hmc,full > gpurun_out/r4/planner_bench_f64_u.txt 2>&1; cat gpurun_out/r4/planner_bench_f64_u.txt
