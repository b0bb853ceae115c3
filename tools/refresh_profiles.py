#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_profile.sh run (gpurun_out/<dir>) into profiles/ (tracked).
usage: refresh_profiles.py gpurun_out/prof_xxx gpurun_out/bench_xxx.json [cfg45_summary.txt]"""
import json, os, shutil, sqlite3, subprocess, sys

src, bench_json = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(here, "profiles")
subprocess.check_call([sys.executable, os.path.join(here, "tools", "summarize_prof.py"), src,
                       os.path.join(prof, "r1_hmc_reg16x13_rocprof.txt")], stdout=subprocess.DEVNULL)
lines = []
for c in (3, 4, 5):
    con = sqlite3.connect(f"{src}/trace_cfg{c}/cfg{c}_results.db")
    lines.append(f"## config {c}: rocprofv3 --kernel-trace --stats -- python3 tools/bench_configs.py {c}   (durations in us)")
    lines.append(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}  {name}")
    lines.append("")
open(os.path.join(prof, "r1_configs_3_4_5_kernel_trace.txt"), "w").write("\n".join(lines) + "\n")
shutil.copy(f"{src}/sweep.log", os.path.join(prof, "r1_variant_sweep.log"))
shutil.copy(f"{src}/configs.jsonl", os.path.join(prof, "r1_configs_1_3_4_5.jsonl"))


def avg(db, ctr):
    con = sqlite3.connect(db)
    return con.execute("select avg(value) from counters_collection where counter_name=? and kernel_name like '%k_chain%'", (ctr,)).fetchone()[0]


f = avg(f"{src}/pmc_FETCH_SIZE/bench_results.db", "FETCH_SIZE")
w = avg(f"{src}/pmc_WRITE_SIZE/bench_results.db", "WRITE_SIZE")
tp = os.path.join(prof, "r1_traffic.json")
t = json.load(open(tp))
t["FETCH_SIZE_KB_per_launch"], t["WRITE_SIZE_KB_per_launch"], t["hbm_bytes_per_launch"] = f, w, (2 * f + w) * 1024
json.dump(t, open(tp, "w"), indent=1)
line = open(bench_json).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(prof, "r1_bench_line.json"), "w").write(line + "\n")
if len(sys.argv) > 3:
    shutil.copy(sys.argv[3], os.path.join(prof, "r1_configs_4_5_pmc.txt"))
print(line[:240])
