#!/bin/bash
# round 4, GPU call AF: full suite, smoke, bench, rocprofv3 passes (final state of the round)
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_af.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_af.log; tail -6 gpurun_out/r4/gpu_tests_af.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > gpurun_out/r4/bench_af.json 2> gpurun_out/r4/bench_af.err ) 2>&1 | grep real; tail -c 700 gpurun_out/r4/bench_af.json; echo
bash tools/gpu_profile_r4.sh prof_r4c; ls gpurun_out/prof_r4c
