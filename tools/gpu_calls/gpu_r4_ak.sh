#!/bin/bash
# round 4, GPU call AK: randomised parity fuzz -- float64 models under both policies, float32 under the default policy
mkdir -p gpurun_out/r4
for args in "150 11 auto float64" "120 12 full float64" "100 13 auto float32"; do
  timeout 1500 python tests/fuzz_parity.py $args > gpurun_out/r4/fuzz_$(echo $args | tr ' ' '_').log 2>&1; echo "rc=$? ($args)"; grep -c SKIP gpurun_out/r4/fuzz_$(echo $args | tr ' ' '_').log; grep "FAIL\|^fuzz:" gpurun_out/r4/fuzz_$(echo $args | tr ' ' '_').log | tail -12
done
