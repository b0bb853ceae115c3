#!/bin/bash
# round 4, GPU call T: float64 HMC with float32 interior gradients -- tests, then the chain grid of the float64 model
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "float64 or posterior_matches_reference or planner" -s > gpurun_out/r4/gpu_tests_t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_t.log; tail -30 gpurun_out/r4/gpu_tests_t.log
timeout 900 python tools/chain_grid.py f64 64 256 1024 2048 4096 5120 8192 16384 > gpurun_out/r4/chain_grid_f64_t.txt 2>&1; cat gpurun_out/r4/chain_grid_f64_t.txt
