#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1800 tools/gpu_profile_r4.sh r4/prof > gpurun_out/r4/prof.log 2>&1; tail -2 gpurun_out/r4/prof.log; head -25 gpurun_out/r4/prof/summary.txt
