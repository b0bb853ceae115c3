#!/bin/bash
# round 4, GPU call AL: the fuzz again, more cases and other seeds, after the float64 p = 32 fix; the new regression test
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -q -k "padded_width_32" 2>&1 | tail -3
for args in "300 21 auto float64" "300 22 full float64" "250 23 auto float32" "250 24 full float32"; do
  f=gpurun_out/r4/fuzz2_$(echo $args | tr ' ' '_').log
  timeout 1500 python tests/fuzz_parity.py $args > $f 2>&1; echo "rc=$? ($args)"; grep "FAIL\|^fuzz:" $f | tail -12
done
