#!/usr/bin/env python3
"""Scan gfx950 code objects for the miscompile behind round 4's wrong-result kernel (profiles/r6_f64_p32_bisect.txt).

The pattern: a block that re-enables lanes with `s_or_b64 exec, exec, s[..]` (the end of an `if` or of a loop with a per-lane trip
count) must do so BEFORE any instruction that depends on EXEC.  ROCm 7.2's register allocator places live-range-split copies and
VGPR spills (`v_mov_b64`, `v_accvgpr_write_b32`, `scratch_store`) at the very top of such a block when the block's prologue also holds
SGPR spills (`v_writelane_b32`) -- ahead of the EXEC restore -- so they run for the lanes of the incoming edge only, and the lanes
the `s_or_b64` switches back on read stale registers afterwards.  It needs register pressure at a divergent join: exactly what
`-amdgpu-sched-strategy=max-ilp` produces in a kernel that already spills.

    python tools/exec_prologue_scan.py <code object | object file with a gfx950 bundle | directory of them> ...

Prints one line per finding (kernel, address of the EXEC restore, the instructions ahead of it); exit code 1 if there are any.
`logreg_amd/build.py` runs the same scan over every unit of the library and fails the build on a finding."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from logreg_amd.isa_gate import llvm_tool, scan_paths  # noqa: E402


def demangle(name):
    try:
        r = subprocess.run([llvm_tool("llvm-cxxfilt"), name], capture_output=True, text=True)
        return r.stdout.strip() or name
    except OSError:
        return name


def main():
    findings = scan_paths(sys.argv[1:])
    for f in findings:
        print("%s  %s\n    %#x  %s   <- %d EXEC-dependent instruction(s) ahead of the restore in its block:" % (
            f["unit"], demangle(f["kernel"]), f["addr"], f["restore"], len(f["ahead"])))
        for a, t in f["ahead"][:12]:
            print("        %#x  %s" % (a, t))
        if len(f["ahead"]) > 12:
            print("        ... %d more" % (len(f["ahead"]) - 12))
    print("%d finding(s)" % len(findings))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
