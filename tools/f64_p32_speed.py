#!/usr/bin/env python3
"""float64 models at 17 <= p <= 32: the fused distributed-state kernel (k_chain_dist) against the stepwise engine that ran these
models in round 4.  HMC L=20 and MALA, n = 200, p = 24; chain-iterations/s (HIP events)."""
import ctypes as Ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
import bench
L = _lib.load()
stream = Ct.c_void_p(); _lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)
for n, p in ((200, 24), (200, 32), (500, 20)):
    X, y, _ = la.synthetic_logreg(n, p, seed=7, beta_sd=0.3 / np.sqrt(p))
    m = la.LogReg(X, y, np.ones(p), dtype="float64")
    sc = 1.0 / np.sqrt(n)
    kerns = {"hmc L=20": (la.hmcKernel(m.lpost, m.glp, eps=0.3 * sc, l=20, dmm=np.ones(p)), 2, 5),
             "mala": (la.malaKernel(m.lpost, m.glp, dt=0.05 * sc * sc, pre=np.ones(p)), 2, 50)}
    for C in (256, 1024, 4096, 16384):
        q0 = 0.3 * sc * np.random.default_rng(1).standard_normal((C, p))
        for name, (k, iters, thin) in kerns.items():
            row = [f"n={n} p={p} chains={C} {name}:"]
            for mode, g in (("auto", 0), ("lds", 64), ("lds", 16), ("stepwise", 0)):
                try:
                    cs = la.ChainSet(k, q0, seed=3, stream=stream, precision="full", mode=mode, group=g)
                    ms = bench._timed_chainset(la, timer, cs, iters, thin, repeats=2)
                    pl = cs.plan()
                    row.append(f"{mode}/{g} [{pl['mode']} {pl['group']}] {C * iters * thin / (ms * 1e-3):.3g} it/s")
                except la.LogregHipError as e:
                    row.append(f"{mode}/{g} ERR {str(e)[:40]}")
            print("  ".join(row), flush=True)
