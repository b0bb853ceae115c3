#!/bin/bash
# round 4, GPU call AR: the fuzz with on-device statistics checks
mkdir -p gpurun_out/r4
for args in "400 61 auto float32" "400 62 full float64" "400 63 auto float64"; do
  f=gpurun_out/r4/fuzz7_$(echo $args | tr ' ' '_').log
  timeout 3000 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep -A3 "FAIL\|Traceback" $f | head -40; tail -1 $f
done
