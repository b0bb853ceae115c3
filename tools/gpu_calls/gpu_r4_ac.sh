#!/bin/bash
# round 4, GPU call AC: full suite, smoke, then the rocprofv3 passes of bench.py (tools/gpu_profile_r4.sh)
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_ac.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_ac.log; tail -6 gpurun_out/r4/gpu_tests_ac.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/gpu_profile_r4.sh prof_r4b; ls gpurun_out/prof_r4b; tail -c 600 gpurun_out/prof_r4b/bench_line.json
