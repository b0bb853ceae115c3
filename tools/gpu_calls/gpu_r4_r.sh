#!/bin/bash
mkdir -p gpurun_out/r4
for i in 1 2 3; do timeout 300 python tools/chain_grid.py 5120 9216 18432 20480 2>&1 | grep -v "^#"; done > gpurun_out/r4/chain_grid_r.txt; cat gpurun_out/r4/chain_grid_r.txt
