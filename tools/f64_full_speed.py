#!/usr/bin/env python3
"""The headline workload (HMC L=50, n=200, p=8, 4096 chains) on the float64 model with every evaluation float64 (precision="full"):
chain-iterations/s and the fraction of the fp64 vector peak (bench.py extra.f64's figure), plus MALA."""
import ctypes as Ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
import bench
L = _lib.load()
stream = Ct.c_void_p(); _lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)
X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7), dtype="float64")
q0 = bench.headline_init(0, 4096)
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=50, dmm=np.ones(8))
for C, mode, g in ((4096, "auto", 0), (4096, "lds", 16), (4096, "lds", 8), (4096, "reg", 32), (8192, "auto", 0), (8192, "lds", 16), (16384, "auto", 0)):
    cs = la.ChainSet(k, np.tile(q0, (C // 4096, 1)), seed=42, stream=stream, precision="full", mode=mode, group=g)
    ms = bench._timed_chainset(la, timer, cs, 10, 20, repeats=3)
    its = C * 200 / (ms * 1e-3)
    print(f"HMC all-float64 {C} chains {mode}/{g}: {its:.4g} it/s, {its * 50 * bench.flops_per_grad_eval(200, 8) / 1e12 / 78.6:.3f} of the fp64 vector peak, plan {cs.plan()}, accept {cs.get_accepts().sum() / (C * 620):.4f}", flush=True)
km = la.malaKernel(m.lpost, m.glp, dt=1e-3, pre=np.ones(8))
cs = la.ChainSet(km, np.tile(q0, (2, 1)), seed=42, stream=stream)
ms = bench._timed_chainset(la, timer, cs, 2, 500, repeats=3)
print(f"MALA all-float64 8192 chains: {8192 * 1000 / (ms * 1e-3):.4g} it/s, plan {cs.plan()}")
