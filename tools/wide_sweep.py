#!/usr/bin/env python3
"""Config-5-shaped throughput (HMC L=50, n=4096, p=128) versus chains per GPU (tools only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
n, p = 4096, 128
X, y, _ = la.synthetic_logreg(n, p, seed=20240005, beta_sd=0.1)
m = la.LogReg(X, y, np.ones(p))
k = la.hmcKernel(m.lpost, m.glp, eps=5e-3, l=50, dmm=np.ones(p))
fg = 4 * n * p + 5 * n + 2 * p
for c in [int(a) for a in sys.argv[1:]] or (256, 1024, 4096, 8192):
    cs = la.ChainSet(k, np.zeros((c, p)), seed=5)
    cs.advance(2, 1, keep=False); cs.sync()
    t0 = time.perf_counter(); cs.advance(10, 1, keep=False); cs.sync(); dt = time.perf_counter() - t0
    print(c, cs.plan(), "it/s %.4g" % (c * 10 / dt), "TF %.1f" % (c * 10 * 50 * fg / dt / 1e12), "ms/step %.3f" % (dt / 500 * 1e3), flush=True)
