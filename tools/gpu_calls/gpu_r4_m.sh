#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python tools/two_part_check.py > gpurun_out/r4/two_part_check.txt 2>&1; cat gpurun_out/r4/two_part_check.txt
