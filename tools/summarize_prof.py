#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd sqlite .db files written by tools/gpu_profile.sh) into a
small text file for profiles/.   usage: summarize_prof.py gpurun_out/<dir> profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
cmd_file = os.path.join(src, "command.txt")
cmd = open(cmd_file).read().strip() if os.path.exists(cmd_file) else "python3 bench.py --no-cpu-baseline --no-ess"
lines = [f"# rocprofv3 summary of `{cmd}` ({os.path.basename(src)})",
         "# produced by tools/gpu_profile.sh on one MI355X; kernel-trace and each --pmc set are separate runs", ""]
dbs = glob.glob(os.path.join(src, "trace", "**", "*_results.db"), recursive=True)
db = dbs[0] if dbs else os.path.join(src, "trace", "bench_results.db")
if os.path.exists(db):
    con = sqlite3.connect(db)
    lines.append("## --kernel-trace --stats (durations in us)")
    lines.append(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}  {name}")
    try:
        cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
        want = [c for c in ("name", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_size", "grid_size") if c in cols]
        if want:
            lines.append("")
            lines.append("## dispatch resources: " + ", ".join(want))
            for row in con.execute(f"select distinct {','.join(want)} from kernels"):
                lines.append("  " + " | ".join(str(v) for v in row))
    except sqlite3.Error:
        pass
    lines.append("")
for db in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*_results.db"), recursive=True)):
    con = sqlite3.connect(db)
    pass_name = os.path.relpath(db, src).split(os.sep)[0]
    lines.append(f"## --pmc pass {pass_name} (per-dispatch average over all dispatches of the kernel)")
    for kname, cname, avg, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                            "group by kernel_name, counter_name order by kernel_name, counter_name"):
        if "lr::" in kname:
            lines.append(f"  {cname:28s} {avg:18.1f}  (n={n})  {kname[:70]}")
    lines.append("")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
