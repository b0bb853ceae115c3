#!/bin/bash
# round 4, GPU call J: final validation -- full suite twice (flakiness), smoke, bench, chain grid
mkdir -p gpurun_out/r4
for i in 1 2; do timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_j$i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_j$i.log; tail -4 gpurun_out/r4/gpu_tests_j$i.log; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r4/bench_j.json 2> gpurun_out/r4/bench_j.err; tail -c 400 gpurun_out/r4/bench_j.json; echo
timeout 600 python tools/chain_grid.py 4096 5120 6144 8192 9216 10240 13312 14336 16384 > gpurun_out/r4/chain_grid_j.txt 2>&1; cat gpurun_out/r4/chain_grid_j.txt
