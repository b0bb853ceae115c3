#!/bin/bash
# Round 6: bisect round 4's wrong-result kernel (k_chain<double, 32, 64 lanes per chain, rows in LDS, MALA> of the ROUND-4 source,
# tools/experiments/f64_p32_r4) over the library's four non-default compiler flags, one build per subset (2^4 = 16), plus each
# subset with -amdgpu-spill-sgpr-to-vgpr=0.
#   bash tools/gpu/f64_bisect.sh build      (here, no GPU: hipcc cross-compiles; binaries + device assembly under tools/bin/bisect/)
#   gpurun -- bash tools/gpu/f64_bisect.sh run <outdir>   (GPU box: runs every binary, compares its dump with the reference dump)
# Flag letters in the build names: S = -fno-slp-vectorize, A = -falign-loops=64, I = -mllvm -amdgpu-sched-strategy=max-ilp,
# M = -mllvm -amdgpu-mfma-vgpr-form ('-' = absent); suffix _nov = -mllvm -amdgpu-spill-sgpr-to-vgpr=0.
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); BIN=$ROOT/tools/bin/bisect
BASE="--offload-arch=gfx950 -std=c++17 -I include -Wno-unused-function -Wno-unused-value -O3"
LIBF="-fno-slp-vectorize -falign-loops=64 -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-mfma-vgpr-form"
combos() {
  for s in S -; do for a in A -; do for i in I -; do for m in M -; do echo "$s$a$i$m"; done; done; done; done
}
flags_of() {
  local f=""
  [[ ${1:0:1} == S ]] && f="$f -fno-slp-vectorize"
  [[ ${1:1:1} == A ]] && f="$f -falign-loops=64"
  [[ ${1:2:1} == I ]] && f="$f -mllvm -amdgpu-sched-strategy=max-ilp"
  [[ ${1:3:1} == M ]] && f="$f -mllvm -amdgpu-mfma-vgpr-form"
  echo "$f"
}
if [[ ${1:-build} == build ]]; then
  mkdir -p $BIN
  # the reference: today's source, library flags, all cases (its distributed-state kernel is verified against the float64 oracle)
  /opt/rocm/bin/hipcc $BASE -DREPRO_ONLY_MALA64 -I logreg_amd/csrc $LIBF tools/f64_p32_repro.hip -o $BIN/ref_today || exit 1
  for c in $(combos); do
    for nov in "" _nov; do
      extra=""; [[ -n $nov ]] && extra="-mllvm -amdgpu-spill-sgpr-to-vgpr=0"
      tmp=$(mktemp -d)
      /opt/rocm/bin/hipcc $BASE -DREPRO_R4 -DREPRO_ONLY_MALA64 -I tools/experiments/f64_p32_r4 $(flags_of $c) $extra --save-temps=obj \
          tools/f64_p32_repro.hip -o $tmp/f64_p32_repro || exit 1
      mv $tmp/f64_p32_repro $BIN/r4_$c$nov
      mv $tmp/f64_p32_repro-hip-amdgcn-amd-amdhsa-gfx950.s $BIN/r4_$c$nov.s
      rm -rf $tmp
      echo "built r4_$c$nov"
    done
  done
  exit 0
fi
OUT=$ROOT/gpurun_out/${2:-f64_bisect}; mkdir -p $OUT
R=$OUT/bisect.txt
: > $R
timeout 120 $BIN/ref_today --dump /tmp/ref.txt > /tmp/ref.log 2>&1 || { echo "reference build failed or differs:"; cat /tmp/ref.log; } >> $R
for c in $(combos); do
  for nov in "" _nov; do
    b=r4_$c$nov
    timeout 120 $BIN/$b --dump /tmp/$b.txt > /tmp/$b.log 2>&1
    python3 - "$b" "$(flags_of $c)" <<'PY' >> $R
import re, sys
b, fl = sys.argv[1], sys.argv[2]
def load(p):
    cases, cur = {}, None
    for ln in open(p):
        if ln.startswith("case "): cur = ln.strip(); cases[cur] = []
        else: cases[cur].append(float.fromhex(ln.strip()))
    return cases
ref, r4 = load("/tmp/ref.txt"), load("/tmp/%s.txt" % b)
res = re.search(r"scratch (\d+) B, (\d+) regs", open("/tmp/%s.log" % b).read())
apart, tot, dmax = 0, 0, 0.0
for k in ref:
    a, x = ref[k], r4[k]
    m = re.search(r"p=(\d+) C=(\d+)", k); p, C = int(m.group(1)), int(m.group(2))
    apart += sum(1 for c in range(C) if max(abs(a[c * p + j] - x[c * p + j]) for j in range(p)) > 1e-9)
    tot += C
    dmax = max(dmax, max(abs(u - v) for u, v in zip(a, x)))
print("%-12s scratch %4s B  regs %s  chains apart %3d of %d  max|d| %-8.3g %s   flags: -O3%s%s" % (
    b, res.group(1), res.group(2), apart, tot, dmax, "WRONG" if apart else "exact", fl, " -mllvm -amdgpu-spill-sgpr-to-vgpr=0" if b.endswith("_nov") else ""))
PY
  done
done
cat $R
