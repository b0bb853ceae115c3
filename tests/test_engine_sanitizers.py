"""The HOST ENGINE of liblogreg_hip.so under the sanitizers, in the GPU-less container (SURVEY section 5 "sanitizers").

lr_api.hip / lr_engine.h / lr_model.h / lr_plan.h -- the 1 200 lines that own model handles, device images, workspaces, per-stream
side slots, the fork / join events of two-part plans and every argument check -- are compiled with -fsanitize=address,undefined (and,
separately, -fsanitize=thread) and linked, with the library's own instantiation objects, against tests/host/hip_stub.cpp: the 31 HIP
runtime calls the library makes, on the host heap, launches as validated no-ops.  tests/host/engine_harness.cpp then drives the C ABI:
13 model shapes x 2 dtypes (every image kind) x all four kernel families x both precision policies, host and device buffers,
statistics, planned shards, the Hessian, every forced (mode, group), two chain sets on two streams + a wide model on alternating
streams, 40 error returns, a failing device allocation at EVERY allocation of model creation and of a run (LR_ERR_NOMEM, nothing
leaked, the model still usable), and a model, its stream, buffers and runs on device 1 of two (what a rank > 0 of the multi-GPU job does:
not one allocation, creation or launch may happen with device 0 current, none may touch another device's stream or event).  A sanitizer report, a leak, a wait on an unrecorded event or a bad launch configuration fails the test.
(First run, round 6: found that a failed hipMalloc left HIP's sticky error for the next launch check to trip over -- lr_model.h fail().)"""
import os
import subprocess

import pytest

from conftest import REPO

LLVM = "/opt/rocm/lib/llvm/bin"


def _build(tmp, san, name):
    from logreg_amd import build as b
    b.build(verbose=False)
    cxx = os.path.join(LLVM, "clang++")
    if not os.path.exists(cxx):
        pytest.skip("ROCm's clang++ not found")
    host = ["-O1", "-g", "-std=c++17", *san]
    inc = ["-I", os.path.join(REPO, "logreg_amd", "csrc"), "-I", os.path.join(REPO, "include")]
    o = {k: str(tmp / f"{k}_{name}.o") for k in ("stub", "harness", "api")}
    subprocess.run([cxx, *host, "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-c", os.path.join(REPO, "tests", "host", "hip_stub.cpp"), "-o", o["stub"]],
                   check=True, capture_output=True)
    subprocess.run([cxx, *host, *inc, "-c", os.path.join(REPO, "tests", "host", "engine_harness.cpp"), "-o", o["harness"]], check=True, capture_output=True)
    xh = [a for f in san for a in ("-Xarch_host", f)]
    r = subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", *xh, *inc, '-DLR_BUILD_ID="sanitizer-harness"', "-c",
                        os.path.join(REPO, "logreg_amd", "csrc", "lr_api.hip"), "-o", o["api"]], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    insts = [p for p in b.unit_objects() if os.path.basename(p).startswith("lr_inst_")]
    assert len(insts) == 12
    exe = str(tmp / f"engine_harness_{name}")
    r = subprocess.run([cxx, *san, *o.values(), *insts, "-ldl", "-lpthread", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_host_engine_under_asan_and_ubsan(tmp_path):
    exe = _build(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "asan")
    r = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("engine harness (all):") and last.endswith(" 0 failures"), tail
    launches = int(last.split(":")[1].split("kernel launches")[0])
    assert launches > 3000  # (every engine did launch: chain kernels, the tall and the wide stepwise kernels)
    for fam in ("k_chain*", "k_tall*", "k_wide*"):
        assert int(last.split(fam)[0].split()[-1].strip("(,")) > 100, last


def test_host_engine_two_threads_under_tsan(tmp_path):
    """Two host threads on DIFFERENT handles (a float32 model with two-part launches on its own stream; a float64 wide model on the
    stepwise engine on another): the library's only shared state is per-thread (the error message) or per-handle."""
    exe = _build(tmp_path, ["-fsanitize=thread"], "tsan")
    r = subprocess.run([exe, "threads"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "WARNING: ThreadSanitizer" not in r.stderr, tail
    assert r.stdout.strip().splitlines()[-1].endswith(" 0 failures"), tail
