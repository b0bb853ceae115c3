#!/bin/bash
# Round 6, second stage of the round-4 defect hunt: f64_bisect.sh names the flag (-amdgpu-sched-strategy=max-ilp).  This script
# narrows WHERE: the round-4 source is copied to tools/bin/bisect_src with (a) a __builtin_amdgcn_sched_barrier(0) -- a fence the
# machine scheduler may not move anything across -- at one numbered place of the MALA iteration per build, and (b) a build that
# stores the intermediates of the first iteration (proposal, gradient, value, proposal-density term, log ratio) to a debug buffer,
# compiled with and without max-ilp, so the first quantity that differs is named.
#   bash tools/gpu/f64_bisect_probe.sh build ; gpurun -- bash tools/gpu/f64_bisect_probe.sh run <outdir>
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); BIN=$ROOT/tools/bin/bisect; SRC=$ROOT/tools/bin/bisect_src
BASE="--offload-arch=gfx950 -std=c++17 -I include -Wno-unused-function -Wno-unused-value -O3 -fno-slp-vectorize -DREPRO_R4 -DREPRO_ONLY_MALA64"
ILP="-mllvm -amdgpu-sched-strategy=max-ilp"
if [[ ${1:-build} == build ]]; then
  mkdir -p $BIN $SRC
  cp tools/experiments/f64_p32_r4/*.h $SRC/
  python3 - $SRC <<'PY'
import sys
src = sys.argv[1]
k = open(src + "/lr_kernels.h").read()
SB = lambda n: "\n#if SB_AT == %d || SB_AT == 99\n__builtin_amdgcn_sched_barrier(0);\n#endif\n" % n
def ins_after(s, anchor, text, count=1):
    i = s.index(anchor) + len(anchor)
    return s[:i] + text + s[i:]
# the MALA branch of k_chain (the generic kernel, not the rs16 one): anchors are unique strings of that branch
k = ins_after(k, "else draw_group<T, P, G>(a.seed, gchain, iter, gl, z, logu_t);", SB(1))
k = ins_after(k, "vfma_o<T, P>(a.b, z, advx, xp);", SB(2))
a3 = "vfma_o<T, P>(a.b, z, advx, xp);" + SB(2) + "\n                    eval_lpost<T, P, G, true, true>(rows, m.prior, xp, gp, llp, lprp);"
assert a3 in k
k = k.replace(a3, a3 + SB(3))
k = ins_after(k, "vfma_o<T, P>(a.a, gp, xp, advp);", SB(4))
k = ins_after(k, "const T dq = vdiffsq<T, P>(a.c, x, advp, xp, advx);", SB(5))
dbg = r'''
#ifdef REPRO_DBG
                    if (it == 0 && jt == 0 && writer && chain < 8) {
                        double* d = lr_dbg + chain * 160;
                        for (int j = 0; j < P; ++j) { d[j] = (double)z[j]; d[32 + j] = (double)xp[j]; d[64 + j] = (double)gp[j]; d[96 + j] = (double)g[j]; }
                        d[128] = llp; d[129] = lprp; d[130] = (double)dq; d[131] = logr; d[132] = logu; d[133] = lp;
                    }
#endif
'''
k = ins_after(k, "logr = (llp + lprp) - lp - 0.5 * (double)dq;", SB(6) + dbg)
k = ins_after(k, "const bool acc = logu < logr;  // NaN -> reject, as `np.log(np.random.rand()) < a`", SB(7))
k = k.replace("namespace lr {", "namespace lr {\n#ifndef SB_AT\n#define SB_AT 0\n#endif\n#ifdef REPRO_DBG\n__device__ double lr_dbg[8 * 160];\n#endif\n", 1)
open(src + "/lr_kernels.h", "w").write(k)
d = open(src + "/lr_device.h").read()
old = "#pragma unroll\n            for (int j = 0; j < P; ++j) g[j] = group_sum<G>(g[j]);\n        }\n        vnmsub<T, P>(beta, pr.inv_var, g, grad);"
assert old in d
d = d.replace(old, SB(8).lstrip("\n") + "#pragma unroll\n            for (int j = 0; j < P; ++j) {\n" + SB(10) + "                g[j] = group_sum<G>(g[j]);\n            }\n        }" + SB(9) + "        vnmsub<T, P>(beta, pr.inv_var, g, grad);")
old = "    if constexpr (G >= 32) v = swap16_sum(v);       // rows 0<->1, 2<->3\n"
assert old in d
d = d.replace(old, SB(11) + old + SB(11))
d = d.replace("namespace lr {", "namespace lr {\n#ifndef SB_AT\n#define SB_AT 0\n#endif\n", 1)
open(src + "/lr_device.h", "w").write(d)
PY
  # the repro host: with -DREPRO_DBG it appends the debug buffer to the dump
  python3 - <<'PY'
s = open("tools/f64_p32_repro.hip").read()
hook = '''    if (g_dump) {  // the states of the SECOND kernel'''
add = '''#ifdef REPRO_DBG
    if (g_dump) {
        std::vector<double> dbg(8 * 160);
        CK(hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(lr::lr_dbg), dbg.size() * 8));
        const char* nm[] = {"z", "xp", "gp", "g"};
        for (int c = 0; c < (C < 8 ? C : 8); ++c) {
            for (int v = 0; v < 4; ++v) for (int j = 0; j < p; ++j) fprintf(g_dump, "dbg %s n=%d p=%d chain %d %s[%d] %a\\n", name, n, p, c, nm[v], j, dbg[c * 160 + 32 * v + j]);
            const char* sn[] = {"llp", "lprp", "dq", "logr", "logu", "lp"};
            for (int v = 0; v < 6; ++v) fprintf(g_dump, "dbg %s n=%d p=%d chain %d %s %a\\n", name, n, p, c, sn[v], dbg[c * 160 + 128 + v]);
        }
    }
#endif
'''
assert hook in s
open("tools/bin/bisect_src/f64_p32_repro_dbg.hip", "w").write(s.replace(hook, add + hook))
PY
  for at in 0 1 2 3 4 5 6 7 8 9 10 11 99; do
    /opt/rocm/bin/hipcc $BASE $ILP -I $SRC -DSB_AT=$at tools/f64_p32_repro.hip -o $BIN/sb_$at || exit 1
    echo built sb_$at
  done
  for v in ilp plain; do
    fl=""; [[ $v == ilp ]] && fl="$ILP"
    /opt/rocm/bin/hipcc $BASE $fl -I $SRC -I tools -DREPRO_DBG $SRC/f64_p32_repro_dbg.hip -o $BIN/dbg_$v || exit 1
    echo built dbg_$v
  done
  exit 0
fi
OUT=$ROOT/gpurun_out/${2:-f64_bisect_probe}; mkdir -p $OUT
R=$OUT/probe.txt
: > $R
timeout 120 $BIN/ref_today --dump /tmp/ref.txt > /tmp/ref.log 2>&1
cmp_dump() {  # <name>
  python3 - "$1" <<'PY'
import re, sys
b = sys.argv[1]
def load(p):
    cases, cur = {}, None
    for ln in open(p):
        if ln.startswith("dbg "): continue
        if ln.startswith("case "): cur = ln.strip(); cases[cur] = []
        else: cases[cur].append(float.fromhex(ln.strip()))
    return cases
ref, r4 = load("/tmp/ref.txt"), load("/tmp/%s.txt" % b)
apart, tot = 0, 0
for k in ref:
    a, x = ref[k], r4[k]
    m = re.search(r"p=(\d+) C=(\d+)", k); p, C = int(m.group(1)), int(m.group(2))
    apart += sum(1 for c in range(C) if max(abs(a[c * p + j] - x[c * p + j]) for j in range(p)) > 1e-9)
    tot += C
print("%-10s chains apart %3d of %d  %s" % (b, apart, tot, "WRONG" if apart else "exact"))
PY
}
echo "## one scheduling fence (sched_barrier(0)) per build, round-4 source, -O3 -fno-slp-vectorize + max-ilp" >> $R
echo "## places: 0 none | 1 after the draws | 2 after the proposal | 3 after eval_lpost(prop) | 4 after advance(prop) | 5 after vdiffsq | 6 after logr | 7 after the accept test" >> $R
echo "##         8 before the gradient's group sums | 9 after them | 10 between each coordinate's group sum | 11 around the row swap inside group_sum | 99 all" >> $R
for at in 0 1 2 3 4 5 6 7 8 9 10 11 99; do
  timeout 120 $BIN/sb_$at --dump /tmp/sb_$at.txt > /tmp/sb_$at.log 2>&1
  cmp_dump sb_$at >> $R
done
echo "## first-iteration intermediates, max-ilp build vs default-scheduler build (same source, both with the debug stores)" >> $R
for v in ilp plain; do
  timeout 120 $BIN/dbg_$v --dump /tmp/dbg_$v.txt > /tmp/dbg_$v.log 2>&1
  cmp_dump dbg_$v >> $R
done
python3 - <<'PY' >> $R
def load(p):
    d = {}
    for ln in open(p):
        if ln.startswith("dbg "):
            *key, val = ln.split()
            d[" ".join(key[1:])] = float.fromhex(val)
    return d
a, b = load("/tmp/dbg_ilp.txt"), load("/tmp/dbg_plain.txt")
bad = [(k, a[k], b[k]) for k in b if not (a.get(k) == b[k] or (a.get(k) != a.get(k) and b[k] != b[k]))]
print("debug values compared: %d, differing: %d" % (len(b), len(bad)))
seen = {}
for k, x, y in bad:
    name = k.split(" chain ")[1].split(" ", 1)[1].split("[")[0]
    seen.setdefault(name, []).append((k, x, y))
for name, rows in seen.items():
    print("  %-5s differs in %d entries; first: %s  max-ilp %r  default %r" % (name, len(rows), rows[0][0], rows[0][1], rows[0][2]))
PY
cat $R
