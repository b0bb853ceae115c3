"""ctypes binding of liblogreg_hip.so (the C ABI in include/logreg_hip.h).

There is no CPU fallback: if the library cannot be loaded, or no MI355X is visible, every entry
point raises.  Nothing in this package imports the test oracle under oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "liblogreg_hip.so")

LR_F32, LR_F64 = 0, 1
STATS_ROWS = 7  # LR_STATS_ROWS
PREC_BY_NAME = {"auto": 0, "full": 1, "bf16": 2}  # LR_PREC_*
KIND_BY_NAME = {"rwmh": 0, "mala": 1, "hmc": 2, "ul": 3}  # LR_KIND_*
MODE_AUTO, MODE_REG, MODE_LDS, MODE_GLOBAL, MODE_MFMA, MODE_STEPWISE, MODE_MIXED = -1, 0, 1, 2, 3, 4, 5
MODE_NAMES = {MODE_REG: "reg", MODE_LDS: "lds", MODE_GLOBAL: "global", MODE_MFMA: "mfma", MODE_STEPWISE: "stepwise", MODE_MIXED: "mixed"}
MODE_BY_NAME = {"auto": MODE_AUTO, "reg": MODE_REG, "lds": MODE_LDS, "global": MODE_GLOBAL, "mfma": MODE_MFMA, "stepwise": MODE_STEPWISE, "mixed": MODE_MIXED}


class LogregHipError(RuntimeError):
    pass


class RunOpts(C.Structure):
    _fields_ = [("n_chains", C.c_int64), ("chain_offset", C.c_int64), ("thin", C.c_int64), ("iters", C.c_int64),
                ("iter_offset", C.c_int64), ("seed", C.c_uint64), ("group", C.c_int32), ("mode", C.c_int32),
                ("on_device", C.c_int32), ("stream", C.c_void_p),
                ("stats", C.c_void_p), ("stats_batch", C.c_int64), ("stats_first", C.c_int64), ("stats_slots", C.c_int64),
                ("precision", C.c_int32), ("plan_chains", C.c_int32), ("plan_first", C.c_int64)]


class PlanInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("group", C.c_int32), ("rows", C.c_int32), ("tail_group", C.c_int32), ("tail_rows", C.c_int32),
                ("split", C.c_int64), ("tail_mode", C.c_int32), ("reserved", C.c_int32)]


# name -> (restype, argtypes); every symbol include/logreg_hip.h declares
_vp, _dp, _i32, _i64, _u64 = C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int64, C.c_uint64
_op = C.POINTER(RunOpts)
SYMBOLS = {
    "lr_last_error": (C.c_char_p, []),
    "lr_build_id": (C.c_char_p, []),
    "lr_sizeof_run_opts": (C.c_int, []),
    "lr_device_count": (C.c_int, []),
    "lr_device_cus": (C.c_int, [C.c_int]),
    "lr_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "lr_model_create": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, C.POINTER(_vp)]),
    "lr_model_destroy": (None, [_vp]),
    "lr_model_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "lr_model_debug_opts": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "lr_model_interior_format": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "lr_eval": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _op]),
    "lr_run_rwmh": (C.c_int, [_vp, _vp, _vp, _vp, _op, _vp, _vp]),
    "lr_run_mala": (C.c_int, [_vp, _vp, _vp, C.c_double, _vp, _op, _vp, _vp]),
    "lr_run_ul": (C.c_int, [_vp, _vp, C.c_double, _vp, _op, _vp, _vp]),
    "lr_run_hmc": (C.c_int, [_vp, _vp, C.c_double, _i32, _vp, _op, _vp, _vp]),
    "lr_hessian": (C.c_int, [_vp, _vp, _dp, _vp, _vp, _vp]),
    "lr_stats_reduce": (C.c_int, [C.c_int, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp]),
    "lr_plan_run": (C.c_int, [_vp, _i32, _op, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "lr_plan_run_info": (C.c_int, [_vp, _i32, _op, C.c_void_p]),
    "lr_plan": (C.c_int, [_vp, _i64, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "lr_malloc": (C.c_int, [C.c_int, _u64, C.POINTER(_vp)]),
    "lr_free": (C.c_int, [C.c_int, _vp]),
    "lr_memcpy_h2d": (C.c_int, [C.c_int, _vp, _vp, _u64, _vp]),
    "lr_memcpy_d2h": (C.c_int, [C.c_int, _vp, _vp, _u64, _vp]),
    "lr_memset": (C.c_int, [C.c_int, _vp, C.c_int, _u64, _vp]),
    "lr_stream_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lr_stream_destroy": (C.c_int, [C.c_int, _vp]),
    "lr_stream_sync": (C.c_int, [C.c_int, _vp]),
    "lr_event_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lr_event_destroy": (C.c_int, [C.c_int, _vp]),
    "lr_event_record": (C.c_int, [C.c_int, _vp, _vp]),
    "lr_event_elapsed_ms": (C.c_int, [C.c_int, _vp, _vp, C.POINTER(C.c_float)]),
    # the C-level exchange (RCCL, loaded at first use); the Python face itself goes through torch.distributed
    "lr_comm_unique_id": (C.c_int, [_vp]),
    "lr_comm_create": (C.c_int, [_vp, _i32, _i32, C.c_int, C.POINTER(_vp)]),
    "lr_comm_destroy": (C.c_int, [_vp]),
    "lr_gather": (C.c_int, [_vp, _vp, _vp, _u64, _i32, _vp]),
    "lr_allreduce_sum_f64": (C.c_int, [_vp, _vp, _u64, _vp]),
}

_lib = None


def load():
    """Load the HIP library (building it first if the sources are newer and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build
    if _build.built_extra() and _build.EXTRA_ENV not in os.environ:
        # a development build (e.g. -DLR_STAMPS: clock reads and printf in hot loops) left behind by a profiling tool must
        # never be what an ordinary run times or tests: replace it with the production build, or refuse
        try:
            _build.build(force=True, verbose=False)
        except Exception as e:
            raise LogregHipError(
                f"{LIB_PATH} is a development build (flags {_build.built_extra()!r}) and the production library could not be "
                f"rebuilt: {e}.  Run `python -m logreg_amd.build --force`, or set {_build.EXTRA_ENV} to use it on purpose.") from e
    if _build.needs_build():  # missing, or older than a kernel source / the header: never run stale kernels
        try:
            _build.build(verbose=False)
        except Exception as e:  # no hipcc / compile error: loud, no fallback
            raise LogregHipError(
                f"liblogreg_hip.so is missing or stale ({LIB_PATH}) and could not be built: {e}. "
                "Run `python -m logreg_amd.build`. There is no CPU fallback.") from e
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise LogregHipError(f"cannot load {LIB_PATH}: {e}. There is no CPU fallback.") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the ABI and the binding drift apart
        fn.restype = res
        fn.argtypes = args
    have, want = L.lr_build_id().decode(), _build.source_hash()
    if have != want:
        raise LogregHipError(f"{LIB_PATH} was built from other sources (build id {have}, sources {want}); "
                             "run `python -m logreg_amd.build --force`")
    if L.lr_sizeof_run_opts() != C.sizeof(RunOpts):
        raise LogregHipError(f"lr_run_opts is {L.lr_sizeof_run_opts()} bytes in the library, {C.sizeof(RunOpts)} in the binding")
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        msg = load().lr_last_error()
        raise LogregHipError(f"liblogreg_hip error {rc}: {msg.decode() if msg else '?'}")


def device_count() -> int:
    return int(load().lr_device_count())


def device_info(device: int = 0) -> str:
    """"pci=... uuid=... name=... cus=..." of `device` (lr_device_info)."""
    buf = C.create_string_buffer(256)
    check(load().lr_device_info(int(device), buf, 256))
    return buf.value.decode()


def require_gpu():
    n = device_count()
    if n <= 0:
        raise LogregHipError("no AMD GPU visible to HIP: logreg_amd has no CPU fallback "
                             "(the CPU restatement under oracle/ is test infrastructure only)")
    return n
