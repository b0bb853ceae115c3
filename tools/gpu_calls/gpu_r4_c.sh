#!/bin/bash
# round 4, GPU call C: suite after rs8 / clamp removal; rs8 vs rs16; planner at very many chains; last-arriver probe; bench
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r4/gpu_tests_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_c.log
tail -8 gpurun_out/r4/gpu_tests_c.log
timeout 900 python tools/planner_bench.py 200,8,8192,mala,auto 200,8,16384,mala,auto 200,8,65536,mala,auto 200,8,8192,rwmh,auto 800,8,131072,mala,auto 800,8,131072,hmc,full 500,16,131072,mala,auto > gpurun_out/r4/planner_bench_c.txt 2>&1
cat gpurun_out/r4/planner_bench_c.txt
tools/bin/last_arriver_probe > gpurun_out/r4/last_arriver_probe.txt 2>&1; cat gpurun_out/r4/last_arriver_probe.txt
timeout 600 python bench.py > gpurun_out/r4/bench_c.json 2> gpurun_out/r4/bench_c.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_c.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
for r in d["extra"]["configs"]:
    print(r["config"], r.get("chain_iterations_per_s"), r.get("us_per_evaluation_all_chains"), r.get("roofline", {}).get("frac"))
print("f64", d["extra"]["f64"]["chain_iterations_per_s"], d["extra"]["f64"]["frac_of_fp64_vector_peak"])
PY
