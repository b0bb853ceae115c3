#!/bin/bash
# round 4, GPU call Y: float32 partials behind the float64 models' interior kernels -- stepwise / wide / float64 tests, the two probes
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -k "stepwise or tall or wide or float64 or stats" > gpurun_out/r4/gpu_tests_y.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_y.log; tail -6 gpurun_out/r4/gpu_tests_y.log
timeout 900 python tools/f64_tall_probe.py 2>&1 | head -2
timeout 600 python - <<'PY'
import json, sys, ctypes as Ct
sys.path.insert(0, '.')
import numpy as np, bench
import logreg_amd as la
from logreg_amd import _lib
L = _lib.load(); st = Ct.c_void_p(); _lib.check(L.lr_stream_create(0, Ct.byref(st)))
r = bench.f64_wide_run(la, L, _lib.check, 0, st)
print({k: r[k] for k in ("us_per_evaluation_all_chains", "accept_rate")}, {k: r["default_policy"][k] for k in ("us_per_evaluation_all_chains", "accept_rate")})
PY
