#!/bin/bash
# the f16 interior on every wide interior kernel: parity tests, config 5 at 1024 chains and as a whole, f32 and f64 models, A/B against the bf16 pieces
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-f16_all}; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "half_precision or config5 or wide or trajector or short_traj" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
run() { echo "== LOGREG_DEBUG_OPTS=$1 prec=$2" >> $OUT/ab.txt; LOGREG_DEBUG_OPTS=$1 python3 tools/cfg5_whole.py 256 1024 2048 4096 8192 --iters 8 --prec $2 >> $OUT/ab.txt 2>&1; }
run "" auto
run wide_f16=0 auto
python3 - $OUT/ab.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("=="): print(l.strip()); continue
    if not l.startswith("{"): print(l.strip()[:200]); continue
    d = json.loads(l)
    print("  chains %6d  %.2f us  frac %.3f  accept %.4f  %s" % (d["chains"], d["us_per_evaluation_all_chains"], d["frac_bf16_peak"], d["accept_rate"], d["plan"]))
PY
