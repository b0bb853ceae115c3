#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -5
