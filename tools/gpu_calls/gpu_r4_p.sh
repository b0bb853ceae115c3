#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python tools/two_part_auto_probe.py > gpurun_out/r4/two_part_auto_probe.txt 2>&1; cat gpurun_out/r4/two_part_auto_probe.txt
