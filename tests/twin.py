"""Build and inject the CPU test double of the C ABI (tests/host/lr_cpu_twin.c) -- TEST INFRASTRUCTURE ONLY.

`install()` compiles the twin with gcc into a temporary directory, binds it with the product's own ctypes signatures
(logreg_amd/_lib.py SYMBOLS) and puts it where `logreg_amd._lib.load()` keeps its handle, so that everything above the C ABI
-- the Python face and the plain-C client -- runs against it in the GPU-less container; `uninstall()` removes it again.  The
product has no switch that would load this library: the injection lives here, under tests/.
"""
import ctypes as C
import os
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "host", "lr_cpu_twin.c")
CFLAGS = ["-O2", "-std=gnu11", "-fPIC", "-fopenmp", "-Wall", "-Wextra", "-fno-fast-math", "-ffp-contract=off"]
_state = {"dir": None, "path": None, "saved": None}


def build() -> str:
    """-> path of the twin's shared library (built once per process)"""
    if _state["path"] is None:
        _state["dir"] = tempfile.TemporaryDirectory(prefix="lr_twin_")
        path = os.path.join(_state["dir"].name, "liblogreg_twin.so")
        subprocess.run(["gcc", *CFLAGS, "-shared", SRC, "-o", path, "-lm", "-lrt", "-lpthread"], check=True, capture_output=True)
        _state["path"] = path
    return _state["path"]


def install():
    """Make `logreg_amd._lib.load()` return the twin.  Returns the bound CDLL."""
    from logreg_amd import _lib
    L = C.CDLL(build())
    for name, (res, args) in _lib.SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the twin and the binding drift apart
        fn.restype = res
        fn.argtypes = args
    assert L.lr_sizeof_run_opts() == C.sizeof(_lib.RunOpts)
    if _state["saved"] is None:
        _state["saved"] = (_lib._lib,)
    _lib._lib = L
    return L


def uninstall():
    """Restore the product's own handle.  Objects made on the twin (models, device arrays) keep the handle of the library that made
    them and release themselves through it, whatever is current; the garbage is collected here only so that they do not
    outlive the test module (exception tracebacks keep frames alive in cycles)."""
    import gc
    from logreg_amd import _lib
    gc.collect()
    if _state["saved"] is not None:
        _lib._lib = _state["saved"][0]
        _state["saved"] = None
