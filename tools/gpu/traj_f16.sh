#!/bin/bash
# A/B of the trajectory kernels' interior formats on config 5: f16 one piece (default) | bf16 x two pieces | bf16 x one piece | exact
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-traj_f16}; mkdir -p $OUT
run() { echo "== LOGREG_DEBUG_OPTS=$1 prec=$2" >> $OUT/ab.txt; LOGREG_DEBUG_OPTS=$1 python3 tools/cfg5_whole.py 4096 8192 16384 --iters 8 --prec $2 >> $OUT/ab.txt 2>&1; }
run "" auto
run wide_f16=0 auto
run "" bf16
run wide_f16=2 bf16
run "" full
python3 - $OUT/ab.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("=="): print(l.strip()); continue
    if not l.startswith("{"): print(l.strip()[:200]); continue
    d = json.loads(l)
    print("  chains %6d  %.2f us  frac %.3f  accept %.4f  %s" % (d["chains"], d["us_per_evaluation_all_chains"], d["frac_bf16_peak"], d["accept_rate"], d["debug_opts"]))
PY
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -k "config5 or wide" > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
