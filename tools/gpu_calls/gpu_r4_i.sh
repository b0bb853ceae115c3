#!/bin/bash
# round 4, GPU call I: full suite after the generalised fp32 matrix-core rule; smoke; planner bench of the new shapes
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_i.log; tail -12 gpurun_out/r4/gpu_tests_i.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/planner_bench.py 200,12,4096,mala,auto 200,24,4096,hmc,full 400,30,4096,mala,auto 900,16,4096,mala,auto 200,24,32768,mala,auto 300,12,16384,mala,auto > gpurun_out/r4/planner_bench_i.txt 2>&1; cat gpurun_out/r4/planner_bench_i.txt
