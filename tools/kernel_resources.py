#!/usr/bin/env python3
"""Register / scratch / spill table of every kernel in the built library's gfx950 code objects.

    python tools/kernel_resources.py [--all] [--json]

Reads the AMDGPU metadata notes (`.private_segment_fixed_size`, `.sgpr_spill_count`, `.vgpr_spill_count`, register counts) of
each object under logreg_amd/lib/obj.  Default: only kernels with scratch or spills.  `logreg_amd.build` runs the same check
as a gate after compiling (logreg_amd/build.py: resource_gate)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from logreg_amd import build as B

if __name__ == "__main__":
    rows = B.kernel_resources()
    show = rows if "--all" in sys.argv else [r for r in rows if r["scratch"] or r["sgpr_spills"] or r["vgpr_spills"]]
    if "--json" in sys.argv:
        print(json.dumps(show, indent=1))
    else:
        for r in show:
            print(f"{r['unit']:18s} scratch={r['scratch']:5d} sgpr_spill={r['sgpr_spills']:4d} vgpr_spill={r['vgpr_spills']:4d} "
                  f"vgpr={r['vgprs']:3d} agpr={r['agprs']:3d} sgpr={r['sgprs']:3d} lds={r['lds']:6d}  {r['name']}")
        print(f"{len(rows)} kernels, {sum(1 for r in rows if r['scratch'])} with scratch, "
              f"{sum(1 for r in rows if r['vgpr_spills'])} with VGPR spills, {sum(1 for r in rows if r['sgpr_spills'])} with SGPR spills")
