#!/bin/bash
# round 4, GPU call D: two-part plans (tests + chain grid), planner at many chains (intermediate counts), last-arriver probe (fixed)
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_part or planner_engine or bit_exact or full_size_properties or distributed_state" > gpurun_out/r4/gpu_tests_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_d.log
tail -12 gpurun_out/r4/gpu_tests_d.log
timeout 600 python tools/chain_grid.py 4096 4608 5120 6144 7168 8192 9216 10240 12288 13312 16384 > gpurun_out/r4/chain_grid.txt 2>&1; cat gpurun_out/r4/chain_grid.txt
tools/bin/last_arriver_probe > gpurun_out/r4/last_arriver_probe.txt 2>&1; cat gpurun_out/r4/last_arriver_probe.txt
timeout 1200 python tools/planner_bench.py 800,8,16384,mala,auto 800,8,32768,mala,auto 800,8,65536,mala,auto 800,8,32768,hmc,full 800,8,65536,hmc,full 500,16,16384,mala,auto 500,16,32768,mala,auto 500,16,65536,mala,auto 1000,8,65536,mala,auto 1000,8,131072,mala,auto 500,16,65536,hmc,full > gpurun_out/r4/planner_bench_d.txt 2>&1
cat gpurun_out/r4/planner_bench_d.txt
