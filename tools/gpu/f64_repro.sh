#!/bin/bash
# tools/f64_p32_repro.hip under several sets of compiler flags (run on the GPU box): today's replicated-state kernel against the
# distributed-state kernel, and the ROUND-4 replicated-state kernel (tools/experiments/f64_p32_r4) against the same reference dump.
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-f64_repro}; mkdir -p $OUT
BASE="--offload-arch=gfx950 -std=c++17 -I include -Wno-unused-function -Wno-unused-value"
LIBF="-O3 -fno-slp-vectorize -falign-loops=64 -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
: > $OUT/repro.txt
i=0
for flags in "$LIBF" "$LIBF -mllvm -amdgpu-spill-sgpr-to-vgpr=0" "-O3" "-O1"; do
  i=$((i+1))
  echo "## today's source: hipcc $flags" >> $OUT/repro.txt
  if timeout 900 /opt/rocm/bin/hipcc $BASE -I logreg_amd/csrc $flags tools/f64_p32_repro.hip -o /tmp/f64_repro_$i >> $OUT/repro.txt 2>&1; then
    timeout 120 /tmp/f64_repro_$i --dump /tmp/ref_$i.txt >> $OUT/repro.txt 2>&1; echo "exit code $?" >> $OUT/repro.txt
  else
    echo "compile failed" >> $OUT/repro.txt
  fi
  echo "## ROUND-4 source (commit 1fee576^), replicated-state kernel alone: hipcc $flags" >> $OUT/repro.txt
  if timeout 900 /opt/rocm/bin/hipcc $BASE -DREPRO_R4 -I tools/experiments/f64_p32_r4 $flags tools/f64_p32_repro.hip -o /tmp/f64_repro_r4_$i >> $OUT/repro.txt 2>&1; then
    timeout 120 /tmp/f64_repro_r4_$i --dump /tmp/r4_$i.txt | sed 's/vs distributed-state kernel/vs ITSELF/' | grep -o "^.*lanes/chain=[0-9]*: replicated-state kernel ([^)]*)" >> $OUT/repro.txt 2>&1
    python3 - <<PY >> $OUT/repro.txt
import re
def load(p):
    cases, cur = {}, None
    for ln in open(p):
        if ln.startswith("case "): cur = ln.strip(); cases[cur] = []
        else: cases[cur].append(float.fromhex(ln.strip()))
    return cases
ref, r4 = load("/tmp/ref_1.txt"), load("/tmp/r4_$i.txt")
for k in ref:
    a, b = ref[k], r4.get(k)
    if b is None: print("   ", k, "missing"); continue
    m = re.search(r"p=(\d+) C=(\d+)", k); p, C = int(m.group(1)), int(m.group(2))
    bad = sum(1 for c in range(C) if max(abs(a[c * p + j] - b[c * p + j]) for j in range(p)) > 1e-9)
    print("    round-4 kernel vs the distributed-state kernel of today's default build, %-40s chains apart %d of %d, max |d| %.3g" % (k, bad, C, max(abs(x - y) for x, y in zip(a, b))))
PY
  else
    echo "compile failed" >> $OUT/repro.txt
  fi
done
cat $OUT/repro.txt
