#!/bin/bash
# round 4, GPU call E: where do LDS rows with 8 lanes per chain / the fp32 matrix-core kernel overtake one chain per wave in registers?
mkdir -p gpurun_out/r4
S=""
for shape in 500,16 400,16 300,12 800,8 600,8 1000,8 400,8; do for C in 4096 8192 16384; do for k in mala,auto hmc,full; do S="$S $shape,$C,$k"; done; done; done
timeout 2400 python tools/planner_bench.py $S > gpurun_out/r4/planner_bench_e.txt 2>&1
cat gpurun_out/r4/planner_bench_e.txt
