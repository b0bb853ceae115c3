"""Load the SECOND build of the product's own sources (logreg_amd/lib_alt/liblogreg_hip.so: the compiler's default scheduler,
SLP vectorisation on, no loop alignment, AGPR-form MFMA) -- TEST INFRASTRUCTURE ONLY.

`install()` binds it with the product's ctypes signatures and puts it where `logreg_amd._lib.load()` keeps its handle, so that
the whole Python face runs on it; `uninstall()` restores the production handle.  Objects keep the handle of the library that made
them.  tests/test_gpu_builds.py runs the same fuzz cases through both builds and demands bit-identical results: the flag sets
differ only in scheduling and register allocation, which may not change a result (profiles/r6_f64_p32_bisect.txt: once they did).
The product has no switch that would load this library."""
import ctypes as C

_state = {"lib": None, "saved": None}


def load():
    """-> the bound CDLL of the second build (built here if missing or stale and hipcc is present)"""
    if _state["lib"] is None:
        from logreg_amd import _lib, build
        if build.needs_build(alt=True):
            build.build(alt=True, verbose=False)
        L = C.CDLL(build.ALT_LIB)
        for name, (res, args) in _lib.SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        have, want = L.lr_build_id().decode(), build.source_hash(alt=True)
        assert have == want, f"{build.ALT_LIB} was built from other sources ({have} != {want})"
        assert L.lr_sizeof_run_opts() == C.sizeof(_lib.RunOpts)
        _state["lib"] = L
    return _state["lib"]


def install():
    from logreg_amd import _lib
    _lib.load()  # (the production handle first: it is what uninstall() restores)
    L = load()
    if _state["saved"] is None:
        _state["saved"] = (_lib._lib,)
    _lib._lib = L
    return L


def uninstall():
    from logreg_amd import _lib
    if _state["saved"] is not None:
        _lib._lib = _state["saved"][0]
        _state["saved"] = None
