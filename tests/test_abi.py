"""The C-ABI library loads without a GPU and exports exactly what include/logreg_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    txt = open(os.path.join(REPO, "include", "logreg_hip.h")).read()
    return sorted(set(re.findall(r"LR_API\s+[\w\s\*]+?\b(lr_\w+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("lr_model_create", "lr_eval", "lr_run_rwmh", "lr_run_mala", "lr_run_ul", "lr_run_hmc", "lr_last_error"):
        assert must in syms
    assert len(syms) >= 20


def test_library_builds_loads_and_exports_every_declared_symbol():
    from logreg_amd import _lib, build
    build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} declared in logreg_hip.h but not exported"
    # the ctypes binding covers the same set: no drift between header and binding
    assert sorted(_lib.SYMBOLS) == declared_symbols()
    _lib.load()


def test_reference_citations_in_header():
    txt = open(os.path.join(REPO, "include", "logreg_hip.h")).read()
    for cite in ("fit-np-hmc.py:23-24", ":44-47 (glp)", "fit-np-mala.py:61-78", "fit-numpy.py:53-62",
                 "fit-np-hmc.py:56-87", "fit-np-ul.py:61-68"):
        assert cite in txt


def test_no_gpu_means_loud_failure_not_fallback(pima):
    import logreg_amd as la
    if la.device_count() > 0:
        pytest.skip("GPU present")
    X, y = pima
    with pytest.raises(la.LogregHipError, match="no CPU fallback"):
        la.LogReg(X, y, 1.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "logreg_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "lr_oracle" not in src and "liblr_oracle" not in src, f


def test_plain_c_client_compiles_against_the_header(tmp_path):
    """examples/fit_bayes.c (the reference's C/fit-bayes.c as a client of the C ABI) builds with gcc."""
    import subprocess
    from logreg_amd import _lib, build
    build.build(verbose=False)
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "fit_bayes"
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "examples", "fit_bayes.c"), "-L", lib_dir, "-llogreg_hip",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    assert exe.exists()


def test_no_kernel_of_the_library_uses_scratch_memory():
    """The build gate (logreg_amd/build.py resource_gate) on the objects the library was linked from: every gfx950 code object's
    `.private_segment_fixed_size` is 0 -- no whitelist.  A register spill to scratch is a large silent slowdown, and the one kernel
    family ever found computing wrong results (round 4, float64 at padded p = 32) was one that spilled."""
    from logreg_amd import build as b
    b.build(verbose=False)
    rows = b.kernel_resources()
    assert len(rows) > 400 and all(r["name"] for r in rows)
    assert {r["unit"] for r in rows} >= {f"lr_inst_{dt}_p{p}" for dt in ("f32", "f64") for p in (4, 8, 16, 32, 64, 128)}
    bad = [(r["unit"], r["name"], r["scratch"]) for r in rows if r["scratch"]]
    assert not bad, bad
    b.resource_gate(strict=True, verbose=False)


def test_no_join_block_runs_instructions_ahead_of_its_exec_restore():
    """The second build gate (logreg_amd/build.py exec_prologue_gate, logreg_amd/isa_gate.py) on the objects both libraries are linked
    from: no block entered with EXEC = 0 (target of s_cbranch_execz / fall-through of s_cbranch_execnz) has an EXEC-dependent
    instruction ahead of its `s_or_b64 exec, exec, ...`.  That placement -- a register-allocator copy in front of the restore -- is what
    made round 4's float64 p = 32 MALA kernel wrong in every chain under -amdgpu-sched-strategy=max-ilp (profiles/r6_f64_p32_bisect.txt)."""
    from logreg_amd import build as b, isa_gate
    for alt in (False, True):
        b.build(verbose=False, alt=alt)
        objs = b.unit_objects(alt)
        assert len(objs) == 13 and all(os.path.exists(o) for o in objs)
        assert isa_gate.scan_paths(objs) == []
        b.exec_prologue_gate(strict=True, verbose=False, alt=alt)


def test_the_exec_restore_scan_recognises_the_round_4_pattern():
    """isa_gate.scan_kernel on hand-written instruction lists: the miscompiled shape (copies between the SGPR spills at the head of a
    join and its restore), the same block compiled correctly, the body of an `if` entered by s_cbranch_execnz (runs under the narrowed
    mask by design), an `if` without a skip branch, and a loop exit."""
    from logreg_amd.isa_gate import scan_kernel

    def prog(*rows):
        out, addr = [], 0x100
        labels = {r[1:]: None for r in rows if r.startswith(":")}
        for r in rows:  # first pass: addresses of the labels
            if r.startswith(":"):
                labels[r[1:]] = addr
            else:
                addr += 4
        addr = 0x100
        for r in rows:
            if r.startswith(":"):
                continue
            text, tgt = r, None
            if "->" in r:
                text, lab = [x.strip() for x in r.split("->")]
                tgt = labels[lab]
            out.append((addr, text, tgt))
            addr += 4
        return out

    bad = prog("v_cmp_gt_i64_e32 vcc, s[36:37], v[2:3]", "s_and_saveexec_b64 s[4:5], vcc", "s_cbranch_execz 12 -> join",
               "v_fmac_f64_e32 v[30:31], v[132:133], v[136:137]",
               ":join", "v_writelane_b32 v255, s92, 17", "v_accvgpr_write_b32 a8, v120", "v_mov_b64_e32 v[70:71], v[30:31]",
               "v_writelane_b32 v255, s31, 21", "s_or_b64 exec, exec, s[4:5]", "v_add_f64 v[14:15], v[88:89], v[68:69]", "s_endpgm")
    found = scan_kernel(bad)
    assert len(found) == 1 and [t for _, t in found[0][2]] == ["v_accvgpr_write_b32 a8, v120", "v_mov_b64_e32 v[70:71], v[30:31]"]
    good = prog("s_and_saveexec_b64 s[4:5], vcc", "s_cbranch_execz 12 -> join", "v_fmac_f64_e32 v[30:31], v[132:133], v[136:137]",
                ":join", "v_writelane_b32 v255, s92, 17", "s_or_b64 exec, exec, s[4:5]", "v_mov_b64_e32 v[70:71], v[30:31]", "s_endpgm")
    assert scan_kernel(good) == []
    body = prog("s_and_saveexec_b64 s[48:49], s[10:11]", "s_cbranch_execnz 45 -> body", ":back", "s_or_b64 exec, exec, s[48:49]", "s_endpgm",
                ":body", "v_mov_b32_e32 v225, v1", "global_store_dword v[4:5], v143, off", "s_branch 3 -> back")
    assert scan_kernel(body) == []
    no_skip = prog("s_and_saveexec_b64 s[2:3], vcc", "global_store_dword v[4:5], v143, off", "s_or_b64 exec, exec, s[2:3]", "s_endpgm")
    assert scan_kernel(no_skip) == []
    if_else = prog("s_and_saveexec_b64 s[12:13], s[10:11]", "s_cbranch_execz 19 -> else_", "v_mul_lo_u32 v137, v141, s77",
                   ":else_", "s_andn2_saveexec_b64 s[10:11], s[10:11]", "v_mul_lo_u32 v136, v138, s76", "s_or_b64 exec, exec, s[10:11]", "s_endpgm")
    assert scan_kernel(if_else) == []  # (the `else` body runs under the flipped mask by design)
    bad_else = prog("s_and_saveexec_b64 s[12:13], s[10:11]", "s_cbranch_execz 19 -> else_", "v_mul_lo_u32 v137, v141, s77",
                    ":else_", "v_mov_b64_e32 v[70:71], v[30:31]", "s_andn2_saveexec_b64 s[10:11], s[10:11]", "v_mul_lo_u32 v136, v138, s76",
                    "s_or_b64 exec, exec, s[10:11]", "s_endpgm")
    assert len(scan_kernel(bad_else)) == 1 and scan_kernel(bad_else)[0][2][0][1] == "v_mov_b64_e32 v[70:71], v[30:31]"
    loop_exit = prog(":loop", "v_fmac_f64_e32 v[0:1], v[2:3], v[4:5]", "s_andn2_b64 exec, exec, s[54:55]", "s_cbranch_execnz 9 -> loop",
                     "v_writelane_b32 v254, s10, 57", "v_mov_b64_e32 v[178:179], v[168:169]", "s_or_b64 exec, exec, s[54:55]", "s_endpgm")
    found = scan_kernel(loop_exit)
    assert len(found) == 1 and found[0][2][0][1] == "v_mov_b64_e32 v[178:179], v[168:169]"


def test_second_build_exports_the_same_abi():
    """logreg_amd/lib_alt/liblogreg_hip.so -- the same sources under the compiler's default scheduler, SLP on (tests/altlib.py,
    tests/test_gpu_builds.py) -- exports every symbol of the header and carries its own build id."""
    import ctypes as C
    from logreg_amd import _lib, build
    build.build(verbose=False, alt=True)
    L = C.CDLL(build.ALT_LIB)
    for name in _lib.SYMBOLS:
        assert hasattr(L, name), name
    L.lr_build_id.restype = C.c_char_p
    assert L.lr_build_id().decode() == build.source_hash(alt=True) != build.source_hash()
    assert set(build.TUNING) & set(build.COMMON) and not set(build.TUNING) & set(build.ALT_COMMON)


def test_isa_symbolic_execution_tool_on_a_small_reduction(tmp_path):
    """tools/isa_symexec.py (the analyser that cleared the dataflow of round 4's wrong kernel, profiles/r6_f64_p32_bisect.txt) on a
    hand-written butterfly: two float64 values each reduced over the wave's halves by a v_permlane32_swap pair -- both final sums depend
    on exactly their own input pair and have the same shape; a third, whose high half was taken from another value's register, is the odd one."""
    import subprocess
    import sys
    src = "\n".join([
        "\tv_mov_b32_e32 v10, v0", "\tv_mov_b32_e32 v11, v1", "\ts_nop 1", "\tv_permlane32_swap_b32 v0, v10", "\tv_permlane32_swap_b32 v1, v11", "\ts_nop 0",
        "\tv_add_f64 v[20:21], v[0:1], v[10:11]",
        "\tv_mov_b32_e32 v12, v2", "\tv_mov_b32_e32 v13, v3", "\ts_nop 1", "\tv_permlane32_swap_b32 v2, v12", "\tv_permlane32_swap_b32 v3, v13", "\ts_nop 0",
        "\tv_add_f64 v[22:23], v[2:3], v[12:13]",
        "\tv_mov_b32_e32 v14, v4", "\tv_mov_b32_e32 v15, v3", "\ts_nop 1", "\tv_permlane32_swap_b32 v4, v14", "\tv_permlane32_swap_b32 v5, v15", "\ts_nop 0",
        "\tv_add_f64 v[24:25], v[4:5], v[14:15]", ""])
    f = tmp_path / "k.s"
    f.write_text(src)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "isa_symexec.py"), str(f), "1", "22", "--sums"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [ln for ln in r.stdout.splitlines() if "v_add_f64" in ln]
    assert len(rows) == 3 and "leaves(2): v0 v1" in rows[0] and "leaves(2): v2 v3" in rows[1] and "ODD" not in rows[0] + rows[1]
    assert "ODD SHAPE" in rows[2] or "leaves(3)" in rows[2], rows[2]
