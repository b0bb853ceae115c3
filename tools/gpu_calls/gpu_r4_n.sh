#!/bin/bash
# round 4, GPU call N: two-pass draw batches + Box-Muller on the hardware transcendentals: parity, MALA / RWMH rates, full suite
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "normals or single_iteration or distributed_state or bit_exact or posterior or groups_agree" > gpurun_out/r4/gpu_tests_n.log 2>&1; tail -6 gpurun_out/r4/gpu_tests_n.log
timeout 600 python tools/two_part_check.py > gpurun_out/r4/two_part_check_n.txt 2>&1; grep -E "4608|5120|10240" gpurun_out/r4/two_part_check_n.txt
timeout 300 python tools/planner_bench.py 200,8,8192,mala,auto 200,8,8192,rwmh,auto 200,8,65536,mala,auto 200,8,4096,hmc,full > gpurun_out/r4/planner_bench_n.txt 2>&1; cut -c1-200 gpurun_out/r4/planner_bench_n.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_n_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_n_full.log; tail -5 gpurun_out/r4/gpu_tests_n_full.log
timeout 600 python bench.py > gpurun_out/r4/bench_n.json 2> gpurun_out/r4/bench_n.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_n.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
for r in d["extra"]["configs"]:
    print(r["config"], r.get("chain_iterations_per_s"), r.get("us_per_evaluation_all_chains"))
PY
