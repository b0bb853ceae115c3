#!/bin/bash
# config 5 whole (8192 chains, one GPU): timings of the existing engines + kernel trace + MFMA-busy counters
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-cfg5_whole_base}; mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/cfg5_whole.py 1024 2048 4096 8192 16384 > $OUT/default.txt 2>&1
LOGREG_DEBUG_OPTS=wide_traj=0 python3 tools/cfg5_whole.py 4096 8192 > $OUT/traj0.txt 2>&1
LOGREG_DEBUG_OPTS=wide_traj=0,wide_waves=8 python3 tools/cfg5_whole.py 8192 > $OUT/traj0_w8.txt 2>&1
LOGREG_DEBUG_OPTS=wide_traj=1 python3 tools/cfg5_whole.py 4096 8192 16384 > $OUT/traj1.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c -- python3 $ROOT/tools/cfg5_whole.py 8192 > $OUT/trace.log 2>&1
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o c -- python3 $ROOT/tools/cfg5_whole.py 8192 > $OUT/pmc_$name.log 2>&1
done
python3 - <<PY
import sqlite3, glob, os, csv
out = "$OUT"
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        lines.append("  %-90s calls %6s avg_ns %12s pct %s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"]))
for db in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True)):
    con = sqlite3.connect(db)
    try:
        for k, cn, avg, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"):
            if "lr::" in k:
                lines.append(f"  {cn:28s} {avg:18.1f}  (n={n})  {k[:80]}")
    except Exception as e:
        lines.append(f"  {db}: {e}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/trace $OUT/pmc_*/
cat $OUT/default.txt $OUT/traj0.txt $OUT/traj0_w8.txt $OUT/traj1.txt
