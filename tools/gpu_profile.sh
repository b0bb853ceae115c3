#!/bin/bash
# Run on the GPU box (via gpurun): variant sweep + rocprofv3 kernel trace + PMC passes for bench.py.
# Outputs under gpurun_out/ ; copy the summaries to profiles/ afterwards.
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt; python3 -c "import os; print(len(os.sched_getaffinity(0)))" >> $OUT/nproc.txt
echo "== variant sweep" > $OUT/sweep.log
for v in "reg 16" "reg 32" "reg 64" "lds 1" "lds 8" "lds 64" "global 64" "global 1" "mfma 1" "mfma 4"; do
  set -- $v
  echo "-- mode=$1 group=$2" >> $OUT/sweep.log
  timeout 300 python3 bench.py --mode $1 --group $2 --no-cpu-baseline --no-ess >> $OUT/sweep.log 2>&1
done
for c in 1024 16384 65536 262144; do
  echo "-- chains=$c auto" >> $OUT/sweep.log
  timeout 300 python3 bench.py --chains $c --no-cpu-baseline --no-ess >> $OUT/sweep.log 2>&1
done
cd /tmp
echo "== kernel trace" 
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-ess > $OUT/trace.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_WAIT_ANY"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $pmc -d $OUT/pmc_$name -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-ess > $OUT/pmc_$name.log 2>&1
done
# other configs: kernel traces (config 3 MALA shard, config 4 tall, config 5 wide)
for c in 3 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg$c -o cfg$c -- python3 $ROOT/tools/bench_configs.py $c > $OUT/cfg$c.log 2>&1
done
cd $ROOT; python3 tools/bench_configs.py 1 3 4 5 > $OUT/configs.jsonl 2>&1
ls -R $OUT | head -60
