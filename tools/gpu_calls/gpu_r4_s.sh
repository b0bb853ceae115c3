#!/bin/bash
# round 4, GPU call S: validation after the "remainder after the head" rule -- full suite, smoke, bench, chain grid
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_s.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_s.log; tail -6 gpurun_out/r4/gpu_tests_s.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r4/bench_s.json 2> gpurun_out/r4/bench_s.err; tail -c 300 gpurun_out/r4/bench_s.json; echo
timeout 600 python tools/chain_grid.py 4096 4608 5120 6144 8192 9216 10240 16384 18432 20480 24576 > gpurun_out/r4/chain_grid_s.txt 2>&1; cat gpurun_out/r4/chain_grid_s.txt
