#!/bin/bash
# round 4, GPU call AP: a longer fuzz campaign (4 x 1000 cases)
mkdir -p gpurun_out/r4
for args in "1000 51 auto float32" "1000 52 full float32" "1000 53 auto float64" "1000 54 full float64"; do
  f=gpurun_out/r4/fuzz6_$(echo $args | tr ' ' '_').log
  timeout 3000 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep -A14 "FAIL\|Traceback" $f | head -60; tail -1 $f
done
