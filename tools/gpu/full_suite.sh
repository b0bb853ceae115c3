#!/bin/bash
# the round-end checks on a GPU box: the whole GPU suite, smoke(), the default bench line
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-suite}; mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.4f  roofline %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for r in d["extra"]["configs"]:
    print(r["config"], r.get("us_per_evaluation_all_chains"), r.get("chain_iterations_per_s"), r.get("roofline", {}).get("frac"))
cb = d["cpu_baseline"]
print("cpu", cb["value"], cb["cores"], cb["one_core"]["value"], cb["fp32"]["all_cores"]["value"], cb["fp32"]["one_core"]["value"])
PY
