#!/usr/bin/env python3
"""Phase breakdown of the persistent wide trajectory kernel (lr_wide_persist.h) at config 5; needs a -DLR_STAMPS build."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
L = _lib.load()
rd = L.lr_debug_read_stamps
rd.restype = C.c_int
rd.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
ns, nw = C.c_int(), C.c_int()
rd(None, C.byref(ns), C.byref(nw))
buf = np.zeros((ns.value, nw.value, 16, 16), dtype=np.uint64)
fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg5.json")))
n, p, Ctot = fix["n"], fix["p"], 1024
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
m = la.LogReg(X, y, np.array(fix["pscale"]))
k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((Ctot, p))
cs = la.ChainSet(k, q0, seed=3)
cs.advance(4, 1, keep=False)
cs.sync()
rd(buf.ctypes.data, None, None)
t = buf[0, :256, :8, :6].astype(np.float64) * 0.01 / (fix["l"] - 1)  # us per step
names = ["operand build + barrier", "row loop", "in-workgroup reduction", "publish (stores, drain, barrier, flag)", "poll + barrier", "gather + update + barrier"]
print("persistent kernel, config 5: us per step and phase (median over waves | min | max), total", round(float(np.median(t.sum(axis=2))), 2))
for i, nm in enumerate(names):
    print(f"  {nm:42s} {np.median(t[:, :, i]):6.2f} | {t[:, :, i].min():6.2f} | {t[:, :, i].max():6.2f}")
