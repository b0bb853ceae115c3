#!/bin/bash
# round 4, GPU call V/W: float64 under the default policy (narrow: float32 interior; wide: bf16 interior) -- full suite, bench's f64 blocks
mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_w.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_w.log; tail -12 gpurun_out/r4/gpu_tests_w.log
timeout 600 python bench.py > gpurun_out/r4/bench_w.json 2> gpurun_out/r4/bench_w.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_w.json').read().strip().splitlines()[-1])
print('value',d['value'],d['roofline']['frac'])
for k in ('f64','f64_wide'):
    e=d['extra'][k]; print(k, {x:e[x] for x in e if x not in ('default_policy','note','workload')}); print('  default', {x:e['default_policy'][x] for x in e['default_policy'] if x!='note'})
PY
