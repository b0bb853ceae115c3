#!/bin/bash
# round 4, GPU call F: clocks / power per workload; planner test; rocprofv3 kernel trace + PMC passes of bench.py
mkdir -p gpurun_out/r4
timeout 300 python tools/clock_sample.py > gpurun_out/r4/clocks.txt 2>&1; cat gpurun_out/r4/clocks.txt
timeout 900 python -m pytest tests/test_gpu_planner.py tests/test_gpu_parity.py -m gpu -q -x -k "planner" > gpurun_out/r4/gpu_tests_f.log 2>&1; tail -5 gpurun_out/r4/gpu_tests_f.log
timeout 1800 tools/gpu_profile_r4.sh r4/prof > gpurun_out/r4/prof.log 2>&1; tail -3 gpurun_out/r4/prof.log; head -c 1500 gpurun_out/r4/prof/bench_line.json; echo; head -60 gpurun_out/r4/prof/summary.txt
