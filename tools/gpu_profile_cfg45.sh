#!/bin/bash
# PMC evidence for configs 4 (tall) and 5 (wide): HBM bytes and MFMA-busy of the partial kernels.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-prof_cfg45}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in 4 5; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pmc | tr ' ' '_' | cut -c1-24)
    timeout 600 rocprofv3 --pmc $pmc -d $OUT/cfg${c}_$name -o c -- python3 $ROOT/tools/bench_configs.py $c > $OUT/cfg${c}_$name.log 2>&1
  done
done
python3 - <<PY
import sqlite3, glob, os
out = "$OUT"
lines = ["# rocprofv3 --pmc passes (separate runs) on python3 tools/bench_configs.py {4,5}; per-dispatch averages", ""]
for c in (4, 5):
    lines.append(f"## config {c}")
    for db in sorted(glob.glob(os.path.join(out, f"cfg{c}_*", "*.db"))):
        con = sqlite3.connect(db)
        for k, cn, avg, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"):
            if "lr::" in k:
                lines.append(f"  {cn:28s} {avg:18.1f}  (n={n})  {k[:72]}")
    lines.append("")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
