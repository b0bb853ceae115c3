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
