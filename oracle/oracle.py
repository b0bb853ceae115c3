"""ctypes front-end of oracle/lr_oracle.c -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package `logreg_amd` must never import this module (tests/test_no_oracle_in_product.py
enforces it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liblr_oracle.so")

RWMH, MALA, HMC, UL = 0, 1, 2, 3
KINDS = {"rwmh": RWMH, "mala": MALA, "hmc": HMC, "ul": UL}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)  # builds the float32 timing build too
    return _LIB_PATH


class _Model(C.Structure):
    _fields_ = [("n", C.c_int64), ("p", C.c_int32), ("X", C.c_void_p), ("y", C.c_void_p),
                ("pscale", C.c_void_p)]


class _Kernel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("step", C.c_double), ("l", C.c_int32), ("scale", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_ll.restype = C.c_double
        L.orc_lprior.restype = C.c_double
        L.orc_lpost.restype = C.c_double
        L.orc_alpi.restype = C.c_double
        L.orc_glp.restype = None
        L.orc_run.restype = C.c_int
        L.orc_max_threads.restype = C.c_int32
        L.orc_hmc_step.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_double, dp]
        L.orc_rwmh_step.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_void_p, C.c_void_p, C.c_double, dp]
        L.orc_mala_step.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_double, C.c_void_p, C.c_void_p,
                                    C.c_double, dp]
        L.orc_ul_step.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_leapfrog.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.orc_alpi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p, dp]
        L.orc_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                              C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_int32]
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def draws(seed, chain, it, p):
    z = np.zeros(p)
    u = C.c_double()
    lib().orc_draws(int(seed), int(chain), int(it), int(p), _ptr(z), C.byref(u))
    return z, u.value


class OracleModel:
    """float64 restatement of the model closures (fit-np-hmc.py:23-47) and kernels."""

    def __init__(self, X, y, pscale):
        self.X = _f64(X)
        self.y = _f64(y)
        self.n, self.p = self.X.shape
        self.pscale = _f64(np.broadcast_to(np.asarray(pscale, dtype=np.float64), (self.p,)))
        self._m = _Model(self.n, self.p, self.X.ctypes.data, self.y.ctypes.data, self.pscale.ctypes.data)
        self._mp = C.cast(C.pointer(self._m), C.c_void_p)

    # -- model --------------------------------------------------------------------------------
    def _scalar(self, fn, beta):
        b = _f64(beta)
        if b.ndim == 1:
            return fn(self._mp, _ptr(b))
        return np.array([fn(self._mp, _ptr(_f64(r))) for r in b])

    def ll(self, beta):
        return self._scalar(lib().orc_ll, beta)

    def lprior(self, beta):
        return self._scalar(lib().orc_lprior, beta)

    def lpost(self, beta):
        return self._scalar(lib().orc_lpost, beta)

    def glp(self, beta):
        b = _f64(beta)
        if b.ndim == 1:
            g = np.zeros(self.p)
            lib().orc_glp(self._mp, _ptr(b), _ptr(g))
            return g
        return np.array([self.glp(r) for r in b])

    # -- single steps with explicit draws ------------------------------------------------------
    def leapfrog(self, q, p, eps, l, dmm):
        q = _f64(q).copy(); p = _f64(p).copy(); d = self._vec(dmm)
        lib().orc_leapfrog(self._mp, _ptr(q), _ptr(p), float(eps), int(l), _ptr(d))
        return q, p

    def alpi(self, q, p, dmm):
        q = _f64(q); p = _f64(p); d = self._vec(dmm)
        return lib().orc_alpi(self._mp, _ptr(q), _ptr(p), _ptr(d))

    def _vec(self, v):
        return _f64(np.broadcast_to(np.asarray(v, dtype=np.float64), (self.p,))).copy()

    def hmc_step(self, q, eps, l, dmm, z, u):
        q = _f64(q).copy(); d = self._vec(dmm); z = _f64(z); m = C.c_double()
        acc = lib().orc_hmc_step(self._mp, _ptr(q), float(eps), int(l), _ptr(d), _ptr(z), float(u), C.byref(m))
        return q, acc, m.value

    def mala_step(self, x, ll, dt, pre, z, u):
        x = _f64(x).copy(); d = self._vec(pre); z = _f64(z); m = C.c_double(); l_ = C.c_double(ll)
        acc = lib().orc_mala_step(self._mp, _ptr(x), C.byref(l_), float(dt), _ptr(d), _ptr(z), float(u), C.byref(m))
        return x, l_.value, acc, m.value

    def rwmh_step(self, x, ll, prop_sd, z, u):
        x = _f64(x).copy(); d = self._vec(prop_sd); z = _f64(z); m = C.c_double(); l_ = C.c_double(ll)
        acc = lib().orc_rwmh_step(self._mp, _ptr(x), C.byref(l_), _ptr(d), _ptr(z), float(u), C.byref(m))
        return x, l_.value, acc, m.value

    def ul_step(self, x, dt, pre, z):
        x = _f64(x).copy(); d = self._vec(pre); z = _f64(z)
        lib().orc_ul_step(self._mp, _ptr(x), float(dt), _ptr(d), _ptr(z))
        return x

    # -- chain driver ---------------------------------------------------------------------------
    def run(self, kind, state, *, step=0.0, l=0, scale=1.0, thin=1, iters=1, seed=0, chain_offset=0,
            iter_offset=0, ll_state=None, ext_z=None, ext_u=None, keep=True, threads=1):
        """mcmc() for C chains.  Returns dict(out[iters,C,p], state[C,p], ll[C], accepts[C], margin[C])."""
        st = _f64(state)
        single = st.ndim == 1
        st = np.atleast_2d(st).copy()
        Cn = st.shape[0]
        sc = self._vec(scale)
        k = _Kernel(KINDS[kind], float(step), int(l), sc.ctypes.data)
        ll = np.full(Cn, -np.inf) if ll_state is None else _f64(np.broadcast_to(ll_state, (Cn,))).copy()
        out = np.zeros((iters, Cn, self.p)) if keep else None
        acc = np.zeros(Cn, dtype=np.uint64)
        mar = np.zeros(Cn)
        ez = None if ext_z is None else _f64(ext_z).reshape(iters * thin, Cn, self.p)
        eu = None if ext_u is None else _f64(ext_u).reshape(iters * thin, Cn)
        rc = lib().orc_run(self._mp, C.cast(C.pointer(k), C.c_void_p), _ptr(st), _ptr(ll), Cn, int(chain_offset),
                           int(thin), int(iters), int(iter_offset), int(seed),
                           None if ez is None else _ptr(ez), None if eu is None else _ptr(eu),
                           None if out is None else _ptr(out), _ptr(acc), _ptr(mar), int(threads))
        if rc != 0:
            raise RuntimeError("orc_run failed")
        if single and out is not None:
            out = out[:, 0, :]
        return {"out": out, "state": st[0] if single else st, "ll": ll, "accepts": acc, "margin": mar}


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ---- the same C source compiled in IEEE float32 (`make f32`): a TIMING build for bench.py's cpu_baseline fp32 row
# (SURVEY.md section 8(d)(ii)); no test uses it as a checker -- the parity oracle is the float64 build above.
_LIB32_PATH = os.path.join(_HERE, "_build", "liblr_oracle_f32.so")
_lib32 = None


class _Kernel32(C.Structure):
    _fields_ = [("kind", C.c_int32), ("step", C.c_float), ("l", C.c_int32), ("scale", C.c_void_p)]


def lib32():
    global _lib32
    if _lib32 is None:
        build()
        if not os.path.exists(_LIB32_PATH):
            subprocess.run(["make", "-C", _HERE, "-s", "f32"], check=True)
        L = C.CDLL(_LIB32_PATH)
        L.orc_sizeof_real.restype = C.c_int32
        assert L.orc_sizeof_real() == 4
        L.orc_run.restype = C.c_int
        L.orc_run.argtypes = lib().orc_run.argtypes
        _lib32 = L
    return _lib32


def run_f32(X, y, pscale, kind, state, *, step, l=0, scale=1.0, thin=1, iters=1, seed=0, threads=1):
    """orc_run of the float32 build on C chains (no samples kept) -> dict(state[C,p] float32, accepts[C])."""
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    X, y = f32(X), f32(y)
    n, p = X.shape
    ps = f32(np.broadcast_to(np.asarray(pscale, dtype=np.float32), (p,)))
    m = _Model(n, p, X.ctypes.data, y.ctypes.data, ps.ctypes.data)
    sc = f32(np.broadcast_to(np.asarray(scale, dtype=np.float32), (p,))).copy()
    k = _Kernel32(KINDS[kind], float(step), int(l), sc.ctypes.data)
    st = np.atleast_2d(f32(state)).copy()
    Cn = st.shape[0]
    ll = np.full(Cn, -np.inf, dtype=np.float32)
    acc = np.zeros(Cn, dtype=np.uint64)
    rc = lib32().orc_run(C.cast(C.pointer(m), C.c_void_p), C.cast(C.pointer(k), C.c_void_p), _ptr(st), _ptr(ll), Cn, 0,
                         int(thin), int(iters), 0, int(seed), None, None, None, _ptr(acc), None, int(threads))
    if rc != 0:
        raise RuntimeError("orc_run (float32 build) failed")
    return {"state": st, "accepts": acc}
