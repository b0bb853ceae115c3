#!/bin/bash
# round 4, GPU call AG: the float32-interior kernel at other widths / row counts (float64 models): AUTO against the forced alternatives
mkdir -p gpurun_out/r4
PLANNER_BENCH_DTYPE=float64 timeout 1500 python tools/planner_bench.py 200,3,4096,hmc,auto 200,3,1024,hmc,auto 900,4,4096,hmc,auto 200,12,4096,hmc,auto 200,12,1024,hmc,auto 500,16,4096,hmc,auto 600,8,4096,hmc,auto 600,8,1024,hmc,auto 1000,8,4096,hmc,auto 400,8,8192,hmc,auto > gpurun_out/r4/planner_bench_f64_ag.txt 2>&1; cat gpurun_out/r4/planner_bench_f64_ag.txt
