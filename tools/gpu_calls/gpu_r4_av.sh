#!/bin/bash
# round 4, GPU call AV: a last long fuzz campaign at the final library (4 x 1500 cases; bf16 policy included)
mkdir -p gpurun_out/r4
for args in "1500 91 auto float32" "1500 92 full float64" "1500 93 auto float64" "1500 94 bf16 float32"; do
  f=gpurun_out/r4/fuzz9_$(echo $args | tr ' ' '_').log
  timeout 3000 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep -A12 "FAIL\|Traceback" $f | head -50; tail -1 $f
done
