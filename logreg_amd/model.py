"""Model closures of the reference -- `ll`, `lprior`, `lpost`, `glp` -- backed by the HIP library.

In the reference these are module-level closures over the globals `X, y, pscale`
(Python/fit-np-hmc.py:23-47).  Here `LogReg(X, y, pscale)` owns the device copy of the data and
hands back callables with the same signatures:

    model = LogReg(X, y, pscale)
    ll, lprior, lpost, glp = model.ll, model.lprior, model.lpost, model.glp
    lpost(beta)   # float          for beta of shape [p]   (as in the reference)
    glp(beta)     # ndarray [p]
    lpost(B)      # ndarray [C]    for B of shape [C, p]   (extension: many parameter vectors at once)

Every call runs the `k_eval` HIP kernel; there is no CPU implementation in this package.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import LR_F32, LR_F64, RunOpts, check

_DTYPES = {"float32": (LR_F32, np.float32), "f32": (LR_F32, np.float32), np.float32: (LR_F32, np.float32),
           "float64": (LR_F64, np.float64), "f64": (LR_F64, np.float64), np.float64: (LR_F64, np.float64)}


class DeviceArray:
    """A device allocation made through the C ABI (lr_malloc); exposes __cuda_array_interface__
    so `torch.as_tensor(arr, device="cuda")` can view it without a copy."""

    def __init__(self, device: int, shape, dtype):
        self.device = int(device)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        self._L = _lib.load()  # an allocation is used and released through the library handle that made it
        check(self._L.lr_malloc(self.device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_host(cls, device, arr, dtype=None, stream=None):
        a = np.ascontiguousarray(arr, dtype=dtype)
        d = cls(device, a.shape, a.dtype)
        d.copy_from(a, stream)
        return d

    def copy_from(self, arr, stream=None):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        check(self._L.lr_memcpy_h2d(self.device, self.ptr, a.ctypes.data, self.nbytes, stream))

    def to_host(self, stream=None) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        check(self._L.lr_memcpy_d2h(self.device, out.ctypes.data, self.ptr, self.nbytes, stream))
        return out

    def rows(self, i0: int, i1: int) -> "DeviceArray":
        """Non-owning view of rows [i0, i1) along the first axis."""
        v = DeviceArray.__new__(DeviceArray)
        v.device, v.dtype, v._L = self.device, self.dtype, self._L
        v.shape = (i1 - i0,) + self.shape[1:]
        row_bytes = self.nbytes // max(self.shape[0], 1)
        v.nbytes = (i1 - i0) * row_bytes
        v.ptr = self.ptr + i0 * row_bytes
        v._view_of = self  # keeps the owner alive
        return v

    def zero_(self, stream=None):
        check(self._L.lr_memset(self.device, self.ptr, 0, self.nbytes, stream))

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2,
                "strides": None}

    def free(self):
        if getattr(self, "_view_of", None) is not None:
            self.ptr = None
            return
        if getattr(self, "ptr", None):
            try:
                self._L.lr_free(self.device, self.ptr)
            finally:
                self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ModelFn:
    """One of the reference's model closures; callable like `ll(beta)`."""

    def __init__(self, model: "LogReg", kind: str):
        self.model = model
        self.kind = kind  # "ll" | "lprior" | "lpost" | "glp"
        self.__name__ = kind

    def __call__(self, beta):
        return self.model.eval(beta, (self.kind,))[self.kind]

    def __repr__(self):
        return f"<{self.kind} of {self.model!r}>"


class LogReg:
    """Bayesian logistic regression with independent N(0, pscale^2) priors, resident on one GPU.

    X [n,p] (intercept column included, as the reference builds it, fit-np-hmc.py:18-19),
    y [n] in {0,1}, pscale [p] or scalar (fit-np-hmc.py:31).  dtype = arithmetic type of the
    device path ("float32" default; "float64" = the reference's own arithmetic: every kernel family at every width up to p = 128,
    HMC under the default precision policy with float64 state and end points and a cheaper force inside the trajectory).
    """

    def __init__(self, X, y, pscale, dtype="float32", device: int = 0):
        L = _lib.load()
        _lib.require_gpu()
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if X.ndim != 2 or y.shape != (X.shape[0],):
            raise ValueError(f"X must be [n,p] and y [n]; got {X.shape} and {y.shape}")
        self.n, self.p = X.shape
        self.pscale = np.ascontiguousarray(np.broadcast_to(np.asarray(pscale, dtype=np.float64), (self.p,)))
        try:
            self.dtype_id, self.np_dtype = _DTYPES[dtype]
        except KeyError:
            raise ValueError(f"dtype must be float32 or float64, got {dtype!r}") from None
        self.device = int(device)
        import hashlib
        self.data_hash = hashlib.sha256(X.tobytes() + y.tobytes() + self.pscale.tobytes()).hexdigest()[:16]
        h = C.c_void_p()
        check(L.lr_model_create(X.ctypes.data, y.ctypes.data, self.n, self.p, self.pscale.ctypes.data,
                                self.dtype_id, self.device, C.byref(h)))
        self._h, self._L = h, L  # (the handle belongs to the library that made it)
        self.ll = ModelFn(self, "ll")
        self.lprior = ModelFn(self, "lprior")
        self.lpost = ModelFn(self, "lpost")
        self.glp = ModelFn(self, "glp")

    # ------------------------------------------------------------------------------------------
    @property
    def handle(self):
        if self._h is None:
            raise _lib.LogregHipError("model was closed")
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.lr_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return f"LogReg(n={self.n}, p={self.p}, dtype={np.dtype(self.np_dtype).name}, device={self.device})"

    def debug_opts(self) -> str:
        """The A/B switches this model was created with (LOGREG_DEBUG_OPTS, read once at creation): "" = the defaults."""
        buf = C.create_string_buffer(128)
        check(self._L.lr_model_debug_opts(self.handle, buf, 128))
        return buf.value.decode()

    def interior_format(self) -> str:
        """Operand format of HMC's reduced-precision interior steps under precision="auto" where such a kernel is planned for this
        model: "none", "bf16" (bf16 operand pieces) or "f16" (wide models whose design fits IEEE half precision)."""
        f = C.c_int32()
        check(self._L.lr_model_interior_format(self.handle, C.byref(f)))
        return ("none", "bf16", "f16")[f.value]

    def plan(self, chains: int, group: int = 0, mode: str = "auto") -> dict:
        """Kernel variant the library will launch for `chains` chains."""
        m, g, r = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._L.lr_plan(self.handle, int(chains), int(group), _lib.MODE_BY_NAME[mode], C.byref(m),
                                  C.byref(g), C.byref(r)))
        return {"mode": _lib.MODE_NAMES[m.value], "group": g.value, "rows_per_lane": r.value}

    def hessian(self, beta):
        """(lpost, glp [p], H [p,p]) at one beta, H = -d2 lpost / d beta2 = X^T W X + diag(1/pscale^2), computed on
        the device in float64 in one pass over the rows (`lr_hessian`)."""
        b = np.ascontiguousarray(beta, dtype=np.float64)
        if b.shape != (self.p,):
            raise ValueError(f"beta must have shape ({self.p},); got {b.shape}")
        lp = C.c_double()
        g = np.empty(self.p)
        H = np.empty((self.p, self.p))
        check(self._L.lr_hessian(self.handle, b.ctypes.data, C.byref(lp), g.ctypes.data, H.ctypes.data, None))
        return lp.value, g, H

    # ------------------------------------------------------------------------------------------
    def eval(self, beta, want=("ll", "lprior", "lpost", "glp"), group: int = 0, mode: str = "auto") -> dict:
        """Evaluate the requested closures at beta [p] or [C,p] in one kernel launch."""
        b = np.asarray(beta, dtype=np.float64)
        single = b.ndim == 1
        b2 = np.ascontiguousarray(np.atleast_2d(b), dtype=self.np_dtype)
        if b2.ndim != 2 or b2.shape[1] != self.p:
            raise ValueError(f"beta must have trailing dimension p={self.p}; got shape {b.shape}")
        Cn = b2.shape[0]
        bufs = {k: np.empty((Cn, self.p) if k == "glp" else (Cn,), dtype=self.np_dtype) for k in want}

        def ptr(k):
            return bufs[k].ctypes.data if k in bufs else None
        opts = RunOpts(n_chains=Cn, group=group, mode=_lib.MODE_BY_NAME[mode], on_device=0)
        check(self._L.lr_eval(self.handle, b2.ctypes.data, ptr("ll"), ptr("lprior"), ptr("lpost"), ptr("glp"),
                                  C.byref(opts)))
        out = {}
        for k, v in bufs.items():
            v = v.astype(np.float64)
            out[k] = (v[0] if k == "glp" else float(v[0])) if single else v
        return out
