#!/bin/bash
# round 4, GPU call AO: final state -- full suite, smoke, bench, rocprofv3 passes
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_ao.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ao.log; tail -5 gpurun_out/r4/gpu_tests_ao.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r4/bench_ao.json 2> gpurun_out/r4/bench_ao.err; tail -c 300 gpurun_out/r4/bench_ao.json; echo
bash tools/gpu_profile_r4.sh prof_r4d; ls gpurun_out/prof_r4d | head -3
