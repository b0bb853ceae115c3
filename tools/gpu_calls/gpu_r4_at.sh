#!/bin/bash
# round 4, GPU call AT: float64 rows padded in LDS (bank conflicts) -- float64 tests, float64 chain grid, float64 planner bench, fuzz slice
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "float64 or f64 or closures or variant or mixed or fuzz or planner" 2>&1 | tail -4
timeout 600 python tools/chain_grid.py f64 1024 4096 8192 16384 > gpurun_out/r4/chain_grid_f64_at.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_at.txt
PLANNER_BENCH_DTYPE=float64 timeout 900 python tools/planner_bench.py 200,8,4096,mala,auto 200,8,8192,mala,auto 200,8,8192,rwmh,auto 200,8,8192,hmc,full 200,12,4096,mala,auto > gpurun_out/r4/planner_bench_f64_at.txt 2>&1; cat gpurun_out/r4/planner_bench_f64_at.txt
