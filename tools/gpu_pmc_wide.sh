#!/bin/bash
# PMC passes on the wide partial kernel: tools/gpu_pmc_wide.sh <outdir-name> <chains> (engine via LOGREG_WIDE_BF16)
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmc_wide}; mkdir -p $OUT
CH=${2:-8192}
export TMPDIR=/tmp; cd /tmp
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $pmc -d $OUT/p$i -o c -- python3 $ROOT/tools/wide_sweep.py $CH > $OUT/p$i.log 2>&1
done
python3 - <<PY
import sqlite3, glob, os
out = "$OUT"
for db in sorted(glob.glob(os.path.join(out, "p*", "*.db"))):
    con = sqlite3.connect(db)
    for k, cn, avg, n in con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"):
        if "partial" in k:
            print(f"  {cn:32s} {avg:18.1f}  (n={n})  {k[:60]}")
PY
tail -2 $OUT/p1.log
