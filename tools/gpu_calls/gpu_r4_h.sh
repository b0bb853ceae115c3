#!/bin/bash
# round 4, GPU call H: full suite; bench; padded p = 16 at small n in all-fp32 arithmetic (register plans with 32 lanes per chain)
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r4/gpu_tests_h.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests_h.log; tail -6 gpurun_out/r4/gpu_tests_h.log
timeout 600 python tools/planner_bench.py 200,12,2048,mala,auto 200,12,4096,mala,auto 200,12,8192,mala,auto 200,12,4096,hmc,full 200,16,4096,mala,auto 250,16,4096,hmc,full 200,24,4096,mala,auto 200,24,4096,hmc,full 400,30,4096,mala,auto > gpurun_out/r4/planner_bench_h.txt 2>&1; cat gpurun_out/r4/planner_bench_h.txt
timeout 600 python bench.py > gpurun_out/r4/bench_h.json 2> gpurun_out/r4/bench_h.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_h.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
for r in d["extra"]["configs"]:
    print(r["config"], r.get("chain_iterations_per_s"), r.get("us_per_evaluation_all_chains"), r.get("roofline", {}).get("frac"))
print("f64", d["extra"]["f64"]["kernel_variant"], d["extra"]["f64"]["chain_iterations_per_s"], d["extra"]["f64"]["frac_of_fp64_vector_peak"])
print("f64 wide", d["extra"]["f64_wide"]["us_per_evaluation_all_chains"], d["extra"]["f64_wide"]["frac_of_fp64_matrix_peak"])
PY
