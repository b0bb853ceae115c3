#!/usr/bin/env python3
"""Device assembly of one instantiation unit, with the library's own flags:  python tools/isa.py f64 8 [-o out.s] [extra hipcc flags]
(units: f32|f64 x 4|8|16|32, and f32 64|128 for the wide unit)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from logreg_amd import build as B

dt, p = sys.argv[1], int(sys.argv[2])
rest = sys.argv[3:]
out = f"/tmp/lr_{dt}_p{p}.s"
if rest[:1] == ["-o"]:
    out, rest = rest[1], rest[2:]
if p > 32:
    cmd = [B._hipcc(), *B.COMMON, f"-DLR_P={p}", f"-DLR_SFX={dt}_p{p}", f"-DLR_DTYPE={0 if dt == 'f32' else 1}", os.path.join(B.CSRC, "lr_inst_wide.hip")]
else:
    cmd = [B._hipcc(), *B.COMMON, f"-DLR_T={'float' if dt == 'f32' else 'double'}", f"-DLR_P={p}", f"-DLR_SFX={dt}_p{p}",
           f"-DLR_DTYPE={0 if dt == 'f32' else 1}", os.path.join(B.CSRC, "lr_inst.hip")]
subprocess.run(cmd + ["-S", "--cuda-device-only", "-o", out] + rest, check=True)
print(out)
