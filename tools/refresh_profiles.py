#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_profile.sh run of bench.py (gpurun_out/<dir>) into profiles/ (tracked).
usage: refresh_profiles.py <tag> gpurun_out/<dir>      e.g.  refresh_profiles.py r5 gpurun_out/prof_r5
Writes profiles/<tag>_rocprof_summary.txt (kernel trace + PMC passes of `python3 bench.py --no-cpu-baseline --no-ess`),
profiles/<tag>_traffic.json (HBM bytes per launch of the headline kernel, for bench.py's roofline.traffic) and
profiles/<tag>_bench_line_profiled.json (the bench line of the traced run)."""
import json, os, re, shutil, sys

tag, src = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(here, "profiles")
txt = open(os.path.join(src, "summary.txt")).read()
open(os.path.join(prof, f"{tag}_rocprof_summary.txt"), "w").write(txt)
line = open(os.path.join(src, "bench_line.json")).read().strip()
d = json.loads(line)
open(os.path.join(prof, f"{tag}_bench_line_profiled.json"), "w").write(line + "\n")


def counter(name, kernel_pat):
    m = re.search(r"^\s*%s\s+([0-9.]+)\s+\(n=\d+\)\s+.*%s" % (name, kernel_pat), txt, re.M)
    return float(m.group(1))


kpat = r"k_chain<float, 8, 16, 0, 13, 2>"
f, w = counter("FETCH_SIZE", kpat), counter("WRITE_SIZE", kpat)
t = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --no-cpu-baseline --no-ess`, "
               "" + f"{tag} (tools/gpu_profile.sh; profiles/{tag}_rocprof_summary.txt)",
     "kernel": "lr::k_chain<float, 8, 16, 0, 13, 2>", "kernel_variant": d["config"]["kernel_variant"],
     "chains": d["config"]["chains_per_gpu"], "thin": d["config"]["thin"],
     "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
     "correction": "MI355X_MICROARCH.md HBM section: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE counts 128-B requests "
                   "at 64 B on gfx950 -> doubled (an upper bound: calibrated for wide coalesced reads only)",
     "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(t, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(t)[:300])
