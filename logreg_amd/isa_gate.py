"""Static checks on the gfx950 code objects the library is linked from (run by logreg_amd/build.py after compiling).

`scan_paths` looks for the miscompile behind round 4's wrong-result kernel (profiles/r6_f64_p32_bisect.txt).  The pattern: a block
that re-enables lanes with `s_or_b64 exec, exec, s[..]` (the end of an `if` or of a loop with a per-lane trip count) must do so
BEFORE any instruction that depends on EXEC.  ROCm 7.2's register allocator places live-range-split copies and VGPR spills
(`v_mov_b64`, `v_accvgpr_write_b32`, `scratch_store`) at the very top of such a block when the block's prologue also holds SGPR
spills (`v_writelane_b32`) -- ahead of the EXEC restore -- so they run for the lanes of the incoming edge only, and the lanes the
`s_or_b64` switches back on read stale registers afterwards.  It needs register pressure at a divergent join: what
`-amdgpu-sched-strategy=max-ilp` produces in a kernel that already spills.  CLI: tools/exec_prologue_scan.py."""
import os
import re
import subprocess
import sys
import tempfile



def llvm_tool(name: str) -> str:
    """llvm-objdump / llvm-readelf / llvm-cxxfilt of the ROCm installation the compiler comes from: next to the resolved hipcc
    (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), under $ROCM_PATH, under /opt/rocm, else whatever PATH has."""
    import shutil
    roots = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.path.exists(hipcc):
        roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))
    roots += [os.environ.get("ROCM_PATH", ""), "/opt/rocm"]
    for r in roots:
        cand = os.path.join(r, "lib", "llvm", "bin", name) if r else ""
        if cand and os.path.exists(cand):
            return cand
    found = shutil.which(name)
    if not found:
        raise FileNotFoundError(f"{name} not found (looked next to hipcc, under $ROCM_PATH, /opt/rocm and on PATH)")
    return found

BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\b")
ENDS = re.compile(r"^(s_endpgm|s_setpc_b64|s_swappc_b64)\b")
EXEC_RESTORE = re.compile(r"^s_or_b64 exec, exec, ")
# other writes of EXEC inside a join block, ahead of its restore:
#  * `s_and_saveexec / s_andn2_saveexec / s_or_saveexec`: the block is the `else` of an if / else (or holds a nested `if` whose skip
#    branch was removed): what FOLLOWS runs under the mask it sets, by design -- but what PRECEDES it at the head of the block is as
#    misplaced as in front of a plain restore;
#  * `s_mov_b64 exec, ..` and the like: the set / restore bracket of an SGPR spill to memory (-amdgpu-spill-sgpr-to-vgpr=0 builds);
#    it excuses nothing -- the instructions around it still run under the incoming edge's mask.
EXEC_SAVE = re.compile(r"^s_\w+saveexec\w* ")
EXEC_WRITE = re.compile(r"^(s_\w+ exec(_lo|_hi)?,|v_cmpx_)")
# instructions whose effect does not depend on EXEC: the scalar unit, and the two cross-lane moves the SGPR spills are made of
LANE_AGNOSTIC = re.compile(r"^(s_\w+|v_readlane_b32|v_writelane_b32|v_readfirstlane_b32)\b")


def code_objects(path, tmp):
    """-> gfx950 code object files for `path` (a code object itself, or a host object / library with an offload bundle)"""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":  # e_machine = EM_AMDGPU
        return [path]
    link = os.path.join(tmp, os.path.basename(path))
    if not os.path.exists(link):
        os.symlink(os.path.abspath(path), link)
    subprocess.run([llvm_tool("llvm-objdump"), "--offloading", link], capture_output=True, text=True)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(os.path.basename(path) + ".") and f.endswith("gfx950"))


def kernels(co):
    """-> {symbol: [(address, text, branch target address or None)]} from the disassembly of one code object"""
    out = subprocess.run([llvm_tool("llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    syms, cur = {}, None
    base = {}
    for ln in out.split("\n"):
        m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", ln)
        if m:
            cur = m.group(2)
            syms[cur] = []
            base[cur] = int(m.group(1), 16)
            continue
        m = re.match(r"^\t(.+?)\s*// ([0-9A-F]{12}): [0-9A-F ]+(?:<(.+)\+0x([0-9a-f]+)>|<(.+)>)?\s*$", ln)
        if m and cur is not None:
            text, addr = re.sub(r"\s+", " ", m.group(1).strip()), int(m.group(2), 16)
            tgt = None
            if BRANCH.match(text):
                if m.group(3):
                    tgt = base.get(m.group(3), 0) + int(m.group(4), 16)
                elif m.group(5):
                    tgt = base.get(m.group(5))
            syms[cur].append((addr, text, tgt))
    return syms


def scan_kernel(ins):
    """-> [(address of the EXEC restore, text, [instructions ahead of it in its block that depend on EXEC])]
    Only JOIN blocks count: a block reached with EXEC = 0 -- the target of an `s_cbranch_execz` (the edge that skips an `if` body or a
    loop) or the fall-through of an `s_cbranch_execnz` (a loop's exit).  Nothing the compiler means to happen can sit between the head
    of such a block and its EXEC restore: on that edge it would have no effect at all.  (A block entered by `s_cbranch_execnz` is the
    body of an `if`: its instructions run under the narrowed mask by design and its trailing restore is not a finding.)"""
    starts = {ins[0][0]: set()} if ins else {}
    for i, (addr, text, tgt) in enumerate(ins):
        if tgt is not None:
            starts.setdefault(tgt, set()).add("execz-target" if text.startswith("s_cbranch_execz") else "target")
        if (BRANCH.match(text) or ENDS.match(text)) and i + 1 < len(ins):
            starts.setdefault(ins[i + 1][0], set()).add("execnz-fallthrough" if text.startswith("s_cbranch_execnz") else "fallthrough")
    found = []
    for i, (addr, text, _) in enumerate(ins):
        if not EXEC_RESTORE.match(text):
            continue
        ahead, j = [], i
        while ins[j][0] not in starts and j > 0:
            j -= 1
            a, t, _ = ins[j]
            if EXEC_RESTORE.match(t):
                ahead = []  # what lies between two restores runs under the inner join's mask by design (an `if` nested in an
                break       # `if` whose skip branch was removed); the earlier restore has its own entry
            if EXEC_SAVE.match(t):
                ahead = []
                continue
            if EXEC_WRITE.match(t):
                continue
            if not LANE_AGNOSTIC.match(t):
                ahead.append((a, t))
        if ahead and starts.get(ins[j][0], set()) & {"execz-target", "execnz-fallthrough"}:
            found.append((addr, text, ahead[::-1]))
    return found


def scan_paths(paths):
    findings = []
    with tempfile.TemporaryDirectory(prefix="lr_scan_") as tmp:
        files = []
        for p in paths:
            if os.path.isdir(p):
                files += [os.path.join(p, f) for f in sorted(os.listdir(p)) if f.endswith((".o", ".so", ".hsaco", ".co"))]
            else:
                files.append(p)
        for f in files:
            for co in code_objects(f, tmp):
                for sym, ins in kernels(co).items():
                    for addr, text, ahead in scan_kernel(ins):
                        findings.append({"unit": os.path.basename(f), "kernel": sym, "addr": addr, "restore": text, "ahead": ahead})
    return findings
