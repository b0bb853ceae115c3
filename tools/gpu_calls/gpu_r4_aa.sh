#!/bin/bash
# round 4, GPU call AA: the row-split interior kernel behind float64 wide models -- tests, float64 probe, bench (config 5 float32 must not move)
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "wide or float64 or fullsize or stepwise" > gpurun_out/r4/gpu_tests_aa.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_aa.log; tail -8 gpurun_out/r4/gpu_tests_aa.log
timeout 900 python tools/f64_wide_slices_probe.py 2>&1 | grep "group=0"
timeout 600 python bench.py > gpurun_out/r4/bench_aa.json 2> gpurun_out/r4/bench_aa.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_aa.json').read().strip().splitlines()[-1])
print('value',d['value'],d['roofline']['frac'])
for r in d['extra']['configs']:
    print(r.get('config'), {k:r[k] for k in r if k in ('chain_iterations_per_s','us_per_evaluation_all_chains','us_per_gradient_evaluation')})
e=d['extra']['f64_wide']; print('f64_wide', e['us_per_evaluation_all_chains'], e['default_policy']['us_per_evaluation_all_chains'], e['default_policy']['accept_rate'])
PY
