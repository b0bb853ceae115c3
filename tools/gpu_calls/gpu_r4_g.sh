#!/bin/bash
# round 4, GPU call G: float64 -- Newton reciprocal parity, rows in LDS with 16 lanes per chain against 32 x 7 in registers
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "float64 or closures or f64 or single_iteration or every_kernel_variant" > gpurun_out/r4/gpu_tests_g.log 2>&1; tail -5 gpurun_out/r4/gpu_tests_g.log
PLANNER_BENCH_DTYPE=float64 timeout 900 python tools/planner_bench.py 200,8,2048,hmc,full 200,8,4096,hmc,full 200,8,8192,hmc,full 200,8,16384,hmc,full 200,8,8192,mala,auto > gpurun_out/r4/planner_bench_f64.txt 2>&1; cat gpurun_out/r4/planner_bench_f64.txt
