#!/bin/bash
# Round 6, third stage of the round-4 defect hunt: is the wrong coordinate a DATAFLOW error (wrong instructions) or a TIMING error
# (a hardware hazard nobody padded)?  tools/isa_symexec.py finds the dataflow of the wrong coordinate intact, so this script patches
# the ASSEMBLY of the wrong build -- `s_nop 7` after every instruction of a line range: eight idle issue slots, enough for any
# documented gfx950 hazard -- reassembles it (clang -x assembler, ld.lld) and runs it through a host that loads the code object with
# hipModuleLoad.  If padding a range makes the result exact, the defect is a hazard inside that range; halve the range and repeat.
#   bash tools/gpu/f64_isa_patch.sh build "<name>:<first>-<last>[,<first>-<last>...]" ... ; gpurun -- bash tools/gpu/f64_isa_patch.sh run <outdir>
# Needs tools/gpu/f64_bisect_probe.sh build (tools/bin/bisect_src, dbg_plain) first.
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); BIN=$ROOT/tools/bin/bisect; SRC=$ROOT/tools/bin/bisect_src; PAT=$BIN/patch
LLVM=/opt/rocm/lib/llvm/bin
BASE="--offload-arch=gfx950 -std=c++17 -I include -Wno-unused-function -Wno-unused-value -O3 -fno-slp-vectorize -DREPRO_R4 -DREPRO_ONLY_MALA64 -DREPRO_DBG"
if [[ ${1:-build} == build ]]; then
  shift
  mkdir -p $PAT
  if [[ ! -f $PAT/base.s ]]; then
    /opt/rocm/bin/hipcc $BASE -mllvm -amdgpu-sched-strategy=max-ilp -I $SRC -I tools $SRC/f64_p32_repro_dbg.hip --cuda-device-only -S -o $PAT/base.s 2>/dev/null || exit 1
  fi
  # the host: the same program, the replicated-state kernel launched from a code object given on the command line
  python3 - <<'PY'
s = open("tools/bin/bisect_src/f64_p32_repro_dbg.hip").read()
s = s.replace('static FILE* g_dump = nullptr;', 'static FILE* g_dump = nullptr;\nstatic hipModule_t g_mod;\n')
old = '''        if (k == 0) hipLaunchKernelGGL(k_old, grid, block, lds, 0, m, a);
        else hipLaunchKernelGGL(k_new, grid, block, lds, 0, m, a);'''
new = '''        {
            static_assert(KIND == KIND_MALA && G == 64, "the code object holds this one kernel");
            hipFunction_t f;
            CK(hipModuleGetFunction(&f, g_mod, "_ZN2lr7k_chainIdLi32ELi64ELi1ELi0ELi1EEEvNS_9ModelArgsIT_XT0_EEENS_9ChainArgsIS2_XT0_EEE"));
            struct { ModelArgs<double, P> m; ChainArgs<double, P> a; } args{m, a};
            size_t sz = sizeof(args);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            CK(hipModuleLaunchKernel(f, grid.x, 1, 1, 256, 1, 1, (unsigned)lds, nullptr, nullptr, cfg));
        }'''
assert old in s
s = s.replace(old, new)
s = s.replace('CK(hipMemcpyFromSymbol(dbg.data(), HIP_SYMBOL(lr::lr_dbg), dbg.size() * 8));',
              'hipDeviceptr_t dp; size_t dsz; CK(hipModuleGetGlobal(&dp, &dsz, g_mod, "_ZN2lr6lr_dbgE")); CK(hipMemcpy(dbg.data(), (void*)dp, dbg.size() * 8, hipMemcpyDeviceToHost));')
s = s.replace('if (argc > 2 && std::string(argv[1]) == "--dump") g_dump = fopen(argv[2], "w");',
              'if (argc > 2 && std::string(argv[1]) == "--dump") g_dump = fopen(argv[2], "w");\n    if (hipModuleLoad(&g_mod, argv[3]) != hipSuccess) { printf("cannot load %s\\n", argv[3]); return 3; }')
open("tools/bin/bisect_src/f64_p32_repro_mod.hip", "w").write(s)
PY
  /opt/rocm/bin/hipcc $BASE -I $SRC -I tools $SRC/f64_p32_repro_mod.hip -o $BIN/mod_host || exit 1
  for spec in "$@"; do
    name=${spec%%:*}; ranges=${spec#*:}
    python3 - $PAT/base.s $PAT/$name.s "$ranges" <<'PY'
import sys
src, dst, ranges = sys.argv[1:4]
rs = [tuple(int(x) for x in r.split("-")) for r in ranges.split(",") if r]
out = []
for i, l in enumerate(open(src).read().split("\n"), 1):
    out.append(l)
    t = l.split(";")[0].strip()
    if t and not t.startswith(".") and not t.endswith(":") and l.startswith("\t") and any(a <= i <= b for a, b in rs):
        out.append("\ts_nop 7")
open(dst, "w").write("\n".join(out))
PY
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $PAT/$name.s -o $PAT/$name.o || exit 1
    $LLVM/ld.lld -shared $PAT/$name.o -o $PAT/$name.hsaco || exit 1
    rm -f $PAT/$name.o
    echo "built $name ($ranges)"
  done
  exit 0
fi
OUT=$ROOT/gpurun_out/${2:-f64_isa_patch}; mkdir -p $OUT
R=$OUT/patch.txt
: > $R
timeout 120 $BIN/dbg_plain --dump /tmp/dbg_plain.txt > /tmp/dbg_plain.log 2>&1
for h in $PAT/*.hsaco; do
  name=$(basename $h .hsaco)
  timeout 300 $BIN/mod_host --dump /tmp/$name.txt $h > /tmp/$name.log 2>&1 || echo "$name: exit code $?" >> $R
  python3 - $name <<'PY' >> $R
import sys
name = sys.argv[1]
def load(p):
    d, st = {}, []
    for ln in open(p):
        if ln.startswith("dbg "):
            *key, val = ln.split()
            d[" ".join(key[1:])] = float.fromhex(val)
        elif not ln.startswith("case "):
            st.append(float.fromhex(ln.strip()))
    return d, st
try:
    (a, sa), (b, sb) = load("/tmp/%s.txt" % name), load("/tmp/dbg_plain.txt")
    bad = sorted(k for k in b if not (a.get(k) == b[k] or (a.get(k) != a.get(k) and b[k] != b[k])))
    apart = sum(1 for x, y in zip(sa, sb) if abs(x - y) > 1e-9)
    print("%-14s first-iteration values differing from the default-scheduler build: %3d of %d   final state entries apart: %d of %d   %s   %s" % (
        name, len(bad), len(b), apart, len(sb), "WRONG" if bad or apart else "exact", " ".join(sorted(set(k.split(" chain ")[1].split(" ", 1)[1] for k in bad))[:6])))
except Exception as e:
    print(name, "failed:", e)
PY
done
cat $R
