#!/bin/bash
# round 4, GPU call AN: the fuzz with many-chain cases (two-part plans, matrix-core kernels, planned shards)
mkdir -p gpurun_out/r4
for args in "300 41 auto float32" "300 42 full float32" "300 43 auto float64" "300 44 full float64"; do
  f=gpurun_out/r4/fuzz5_$(echo $args | tr ' ' '_').log
  timeout 1500 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep -A3 "FAIL\|Traceback" $f | head -30; grep -c "C=1031\|C=4097\|C=5120\|C=9000\|C=10240\|C=17000" $f; tail -1 $f
done
