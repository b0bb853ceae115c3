#!/bin/bash
# round 4, GPU call AS: the fuzz with checkpoint / resume checks
mkdir -p gpurun_out/r4
for args in "300 71 auto float32" "300 72 auto float64" "300 73 full float64"; do
  f=gpurun_out/r4/fuzz8_$(echo $args | tr ' ' '_').log
  timeout 3000 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep -A3 "FAIL\|Traceback" $f | head -40; tail -1 $f
done
