"""Build liblogreg_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

    python -m logreg_amd.build [--force] [-j N]

The kernels are instantiated once per (dtype, padded p) in separate translation units compiled
in parallel, then linked with the C-ABI unit into logreg_amd/lib/liblogreg_hip.so (in-tree: the
built library travels to the GPU box with the repo snapshot; it is git-ignored).
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "liblogreg_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
# the SECOND build of the same sources (round 6): the compiler's default scheduler, SLP vectorisation on, AGPR-form MFMA, no loop
# alignment.  Nothing loads it but tests/test_gpu_builds.py, which runs the parity fuzz through both libraries and demands
# bit-identical samples and accept counts: neither build has fast-math, the arithmetic is spelled out in the sources (explicit fma,
# explicit packed types), so scheduling, vectorisation and register allocation may not change a result -- 3000 random cases over every
# engine agree to the last bit (profiles/r6_build_differential.txt); a difference is a miscompile (profiles/r6_f64_p32_bisect.txt)
# or a missed hazard in the hand-written asm.  Scratch is permitted there (no speed is measured on it).
ALT_LIBDIR = os.path.join(HERE, "lib_alt")
ALT_LIB = os.path.join(ALT_LIBDIR, "liblogreg_hip.so")

ARCH = "gfx950"
# -fno-slp-vectorize: packing is explicit in the sources (f32x2); SLP would re-pack the DPP reduction adds
# into v_pk_add_f32 and lose the fused v_add_f32_dpp form (16 more instructions per evaluation)
# -amdgpu-sched-strategy=max-ilp: the kernels run one or two waves per SIMD by design, so latency hiding has
# to come from instruction-level parallelism, not occupancy (HMC hot loop: +7 % over the default strategy)
# -falign-loops=64: the 1.4 KB leapfrog loop of the fused HMC kernel runs one wave per SIMD, nothing hides its
# instruction fetch; when an unrelated edit elsewhere in the file moved the loop head to an address = 4 mod 8 the
# kernel lost 10 % (0.444 -> 0.489 ms per launch, identical instructions).  With every loop head on a 64-byte
# line the time no longer depends on what precedes the loop (8 / 16 / 32 / 64 / 128: 0.451 / 0.442 / 0.450 / 0.443 / 0.447 ms).
# -amdgpu-mfma-vgpr-form: MFMA results in VGPRs.  In the AGPR form the register allocator rotated the 32 gradient
# accumulators of the wide kernels through VGPRs on every trip of the block loop (48 v_accvgpr_read + 32
# v_accvgpr_write per 24 MFMAs: 184 -> 104 instructions per block with the flag); no kernel here needs more than
# 256 registers, so the accumulator file buys nothing.
BASE = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I", CSRC, "-I", INCLUDE]
TUNING = ["-fno-slp-vectorize", "-falign-loops=64", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-mfma-vgpr-form"]
COMMON = BASE[:2] + TUNING + BASE[2:]
ALT_COMMON = list(BASE)
# development flags (e.g. -DLR_STAMPS) come from the environment of the BUILDING process only: they are recorded next to
# the library, and every other process judges staleness against the recorded value, so ranks whose environments differ
# do not rebuild the library back and forth
EXTRA_ENV = "LOGREG_HIPCC_FLAGS"

INSTANCES = [(dt, dtype_id, ctype, p) for dt, dtype_id, ctype in (("f32", 0, "float"), ("f64", 1, "double"))
             for p in (4, 8, 16, 32)]



def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(INCLUDE, "logreg_hip.h")]


def _dirs(alt: bool):
    """-> (library directory, object directory, library path, build-id file, compile flags) of a variant"""
    d = ALT_LIBDIR if alt else LIBDIR
    return d, os.path.join(d, "obj"), os.path.join(d, "liblogreg_hip.so"), os.path.join(d, "build_id.txt"), ALT_COMMON if alt else COMMON


BUILD_ID_FILE = os.path.join(LIBDIR, "build_id.txt")


def built_extra(alt: bool = False) -> str:
    """The development flags the library on disk was built with (second line of build_id.txt; empty if none)."""
    try:
        with open(_dirs(alt)[3]) as f:
            lines = f.read().split("\n")
        return lines[1].strip() if len(lines) > 1 else ""
    except OSError:
        return ""


def source_hash(extra: str | None = None, alt: bool = False) -> str:
    """Content hash of every kernel source, the ABI header and the compile flags: the library carries it
    (`lr_build_id()`), so a stale .so is recognised whatever the file times say (the library is git-ignored and
    travels to the GPU box inside a snapshot whose mtimes mean nothing).  `extra`: development flags; None = the ones
    recorded for the library on disk."""
    import hashlib
    h = hashlib.sha256()
    for path in _sources():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    flags = [f for f in _dirs(alt)[4] if not os.path.isabs(f)] + (built_extra(alt) if extra is None else extra).split()
    h.update(" ".join(flags).encode())  # flags without the -I paths
    return h.hexdigest()[:16]


def built_id(alt: bool = False) -> str | None:
    try:
        with open(_dirs(alt)[3]) as f:
            return f.read().split("\n")[0].strip()
    except OSError:
        return None


def needs_build(alt: bool = False) -> bool:
    """Missing, built from other sources, or built with development flags other than the ones this process asks for
    explicitly (an UNSET variable asks for nothing: such a process takes the library as it is)."""
    if not os.path.exists(_dirs(alt)[2]) or built_id(alt) != source_hash(alt=alt):
        return True
    return EXTRA_ENV in os.environ and os.environ[EXTRA_ENV].split() != built_extra(alt).split()


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r.stderr


def build(force: bool = False, jobs: int | None = None, verbose: bool = True, alt: bool = False) -> str:
    """Compile and link the library (`alt`: the second, differently-compiled build under logreg_amd/lib_alt).  Safe to call from
    several processes at once (one rank per GPU importing the package on a box whose library is stale): an exclusive file lock
    serialises them, the late-comers find the library current and return; the .so is linked under a temporary name and renamed
    into place."""
    libdir, objdir, lib, _, _ = _dirs(alt)
    if not force and not needs_build(alt):
        return lib
    os.makedirs(objdir, exist_ok=True)
    import fcntl
    with open(os.path.join(libdir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build(alt):  # another process built it while this one waited
                return lib
            return _build_locked(jobs, verbose, alt)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _units(alt: bool):
    """-> [(object path, extra compile arguments, source file)] of a variant: one instantiation unit per (dtype, padded p), the wide
    units, the C-ABI unit"""
    objdir = _dirs(alt)[1]
    units = []
    for dt, dtype_id, ctype, p in INSTANCES:
        sfx = f"{dt}_p{p}"
        units.append((os.path.join(objdir, f"lr_inst_{sfx}.o"), [f"-DLR_T={ctype}", f"-DLR_P={p}", f"-DLR_SFX={sfx}", f"-DLR_DTYPE={dtype_id}"],
                      os.path.join(CSRC, "lr_inst.hip")))
    for dt, dtype_id in (("f32", 0), ("f64", 1)):  # wide models: MFMA stepwise engine (bf16 pipe for float32, f64 pipe for float64)
        for p in (64, 128):
            sfx = f"{dt}_p{p}"
            units.append((os.path.join(objdir, f"lr_inst_{sfx}.o"), [f"-DLR_P={p}", f"-DLR_SFX={sfx}", f"-DLR_DTYPE={dtype_id}"],
                          os.path.join(CSRC, "lr_inst_wide.hip")))
    units.append((os.path.join(objdir, "lr_api.o"), None, os.path.join(CSRC, "lr_api.hip")))
    return units


def unit_objects(alt: bool = False):
    """The object files of the CURRENT instantiation table (a stale object of a removed unit in obj/ is nobody's business)."""
    return [u[0] for u in _units(alt)]


def _build_locked(jobs: int | None, verbose: bool, alt: bool) -> str:
    hipcc = _hipcc()
    libdir, objdir, lib, id_file, common = _dirs(alt)
    extra = os.environ.get(EXTRA_ENV, "")
    flags = common + extra.split()
    bid = source_hash(extra, alt)
    jobs_list = [[hipcc, *flags, *(args if args is not None else [f'-DLR_BUILD_ID="{bid}"']), "-c", src, "-o", obj] for obj, args, src in _units(alt)]
    objs = unit_objects(alt)
    jobs = jobs or min(len(jobs_list), max(1, (os.cpu_count() or 2)))
    if verbose:
        print(f"[logreg_amd.build] compiling {len(jobs_list)} units for {ARCH} with {jobs} jobs" + (" (second build: lib_alt)" if alt else ""), flush=True)
    with cf.ThreadPoolExecutor(jobs) as ex:
        for warn in ex.map(_run, jobs_list):
            if warn.strip() and verbose:
                print(warn, file=sys.stderr)
    resource_gate(strict=not extra and not alt, verbose=verbose, alt=alt)
    exec_prologue_gate(strict=not extra, verbose=verbose, alt=alt)
    tmp = lib + f".tmp{os.getpid()}"
    _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp, *objs])
    os.replace(tmp, lib)
    with open(id_file, "w") as f:
        f.write(bid + "\n" + extra + "\n")
    if verbose:
        print(f"[logreg_amd.build] wrote {lib}", flush=True)
    return lib


def resource_gate(strict: bool = True, verbose: bool = True, alt: bool = False) -> None:
    """No kernel of the library may use scratch memory (`.private_segment_fixed_size` of every gfx950 code object must be 0): a
    register spill to scratch is a large, silent slowdown, and the one kernel family ever found computing wrong results (round 4,
    float64 at padded p = 32) was one that spilled.  No whitelist.  Production builds fail here, before the library is linked;
    development builds (LOGREG_HIPCC_FLAGS: instrumentation may cost registers) and the second build (lib_alt: the default
    scheduler keeps fewer values in registers) only report.  If the LLVM tools cannot be found a non-strict build warns and goes on."""
    try:
        rows = kernel_resources(alt=alt)
    except FileNotFoundError as e:
        if strict:
            raise
        print(f"[logreg_amd.build] resource gate skipped: {e}", file=sys.stderr)
        return
    if not rows:
        raise RuntimeError("resource gate: no kernel metadata found under " + _dirs(alt)[1])
    bad = [r for r in rows if r["scratch"]]
    if verbose:
        print(f"[logreg_amd.build] resource gate: {len(rows)} kernels, {len(bad)} with scratch, "
              f"{sum(1 for r in rows if r['vgpr_spills'])} with VGPR->AGPR spills, {sum(1 for r in rows if r['sgpr_spills'])} with SGPR->VGPR-lane spills", flush=True)
    if bad:
        text = "\n".join(f"  {r['unit']}: {r['scratch']} bytes of scratch ({r['vgpr_spills']} VGPR / {r['sgpr_spills']} SGPR spills)  {r['name']}" for r in bad)
        if strict:
            raise RuntimeError("kernels with scratch memory (logreg_amd/build.py resource_gate):\n" + text)
        if verbose:
            print("[logreg_amd.build] " + ("second build" if alt else "development build") + f": {len(bad)} kernels with scratch memory" +
                  ("" if alt else "\n" + text), file=sys.stderr)


def exec_prologue_gate(strict: bool = True, verbose: bool = True, alt: bool = False) -> None:
    """No join block of any kernel may hold an EXEC-dependent instruction ahead of its EXEC restore (logreg_amd/isa_gate.py): the
    register-allocator placement bug that made round 4's float64 p = 32 MALA kernel wrong in every chain
    (profiles/r6_f64_p32_bisect.txt).  A finding is a wrong-result kernel waiting for the input that reads the stale lanes: the build
    fails, whatever the variant."""
    from . import isa_gate
    try:
        findings = isa_gate.scan_paths(unit_objects(alt))
    except FileNotFoundError as e:
        if strict:
            raise
        print(f"[logreg_amd.build] EXEC-restore gate skipped: {e}", file=sys.stderr)
        return
    if verbose:
        print(f"[logreg_amd.build] EXEC-restore gate: {len(findings)} join blocks with EXEC-dependent instructions ahead of their restore", flush=True)
    if findings:
        text = "\n".join(f"  {f['unit']}: {_demangle([f['kernel']])[0]} at {f['addr']:#x}: {len(f['ahead'])} instructions ahead of `{f['restore']}`"
                         f" (first: {f['ahead'][0][1]})" for f in findings)
        if strict:
            raise RuntimeError("kernels with instructions ahead of a join block's EXEC restore (logreg_amd/build.py exec_prologue_gate; "
                               "profiles/r6_f64_p32_bisect.txt):\n" + text)
        print("[logreg_amd.build] development build: EXEC-restore findings\n" + text, file=sys.stderr)


def _demangle(names):
    from .isa_gate import llvm_tool
    try:
        tools = [llvm_tool("llvm-cxxfilt"), "c++filt"]
    except FileNotFoundError:
        tools = ["c++filt"]
    for tool in tools:
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            out = r.stdout.split("\n")
            if r.returncode == 0 and len(out) >= len(names):
                return out[:len(names)]
        except OSError:
            pass
    return list(names)


def kernel_resources(objdir: str | None = None, alt: bool = False):
    """-> one dict per kernel of every gfx950 code object of the current units (or of every *.o under `objdir`): unit, name
    (demangled), scratch bytes (.private_segment_fixed_size), sgpr_spills, vgpr_spills, vgprs, agprs, sgprs, lds (static bytes) -- read
    from the AMDGPU metadata note of the device code bundled in each object file."""
    import re
    import tempfile
    from .isa_gate import llvm_tool
    objdump, readelf = llvm_tool("llvm-objdump"), llvm_tool("llvm-readelf")
    paths = unit_objects(alt) if objdir is None else [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o")]
    rows = []
    with tempfile.TemporaryDirectory(prefix="lr_co_") as tmp:
        for path in paths:
            obj = os.path.basename(path)
            if not os.path.exists(path):
                continue
            link = os.path.join(tmp, obj)
            os.symlink(path, link)
            subprocess.run([objdump, "--offloading", link], capture_output=True, text=True)  # writes <link>.N.<target>
            cos = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(obj + ".") and f.endswith("gfx950")]
            if not cos:
                continue  # a host-only unit
            co = cos[0]
            notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for ln in notes.split("\n"):
                m = re.match(r"\s*(- )?\.(\w+):\s+(\S.*)$", ln)
                if not m:
                    continue
                k, v = m.group(2), m.group(3).strip()
                if k == "agpr_count":  # first per-kernel key (keys are sorted) of a new kernel entry
                    cur = {"unit": obj[:-2], "agprs": int(v)}
                    rows.append(cur)
                elif cur is not None:
                    if k == "name" and "name" not in cur:
                        cur["name"] = v.strip("'")
                    elif k in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "sgpr_count",
                               "group_segment_fixed_size"):
                        cur[{"private_segment_fixed_size": "scratch", "sgpr_spill_count": "sgpr_spills", "vgpr_spill_count": "vgpr_spills",
                             "vgpr_count": "vgprs", "sgpr_count": "sgprs", "group_segment_fixed_size": "lds"}[k]] = int(v)
    rows = [r for r in rows if "name" in r and "scratch" in r]
    for r, d in zip(rows, _demangle([r["name"] for r in rows])):
        r["name"] = d
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--alt", action="store_true", help="the second, differently-compiled build (logreg_amd/lib_alt; tests/test_gpu_builds.py)")
    ap.add_argument("-j", type=int, default=None)
    a = ap.parse_args()
    build(force=a.force, jobs=a.j, alt=a.alt)
