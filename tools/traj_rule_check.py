#!/usr/bin/env python3
"""Where does the one-tile trajectory kernel beat the launch-per-step interior kernels below one chain tile per CU?  us per evaluation,
wide_traj=1 (forced) against wide_traj=0, by image size (n x P x 2 bytes) and chain count."""
import ctypes as Ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la
from logreg_amd import _lib
import bench
L = _lib.load()
stream = Ct.c_void_p()
_lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)

def timed(opt, n, p, C, l=50):
    os.environ["LOGREG_DEBUG_OPTS"] = opt
    X, y, _ = la.synthetic_logreg(n, p, seed=1, beta_sd=0.3 / np.sqrt(p))
    m = la.LogReg(X, y, np.full(p, 2.0))
    k = la.hmcKernel(m.lpost, m.glp, eps=0.4 / np.sqrt(n), l=l, dmm=np.ones(p))
    q0 = (0.3 / np.sqrt(n)) * np.random.default_rng(3).standard_normal((C, p))
    cs = la.ChainSet(k, q0, seed=5, stream=stream)
    return bench._timed_chainset(la, timer, cs, 8, 1) * 1e3 / (8 * l)

for n, p in ((1000, 128), (1500, 128), (2000, 128), (2500, 128), (3000, 128), (4096, 128), (3000, 64), (5000, 64), (8000, 64)):
    P = 64 if p <= 64 else 128
    for C in (256, 1024, 2048, 3072):
        t1, t0 = timed("wide_traj=1", n, p, C), timed("wide_traj=0", n, p, C)
        print(f"n={n} p={p} image {n * P * 2 // 1024} KB chains={C}: trajectory {t1:.2f} us | launch per step {t0:.2f} us  -> {'trajectory' if t1 < t0 else 'launch per step'}", flush=True)
