#!/usr/bin/env python3
"""HMC throughput versus parameter count at n = 200 (padded widths 4..128), AUTO plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
n, C, L = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 50
for p in (3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128):
    X, y, _ = la.synthetic_logreg(n, p, seed=p, beta_sd=0.3)
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=0.05, l=L, dmm=np.ones(p))
    cs = la.ChainSet(k, np.zeros((C, p)), seed=5)
    cs.advance(1, 2, keep=False); cs.sync()
    t0 = time.perf_counter(); cs.advance(4, 5, keep=False); cs.sync(); dt = time.perf_counter() - t0
    ev = C * 20 * L / dt
    fg = 4 * n * p + 5 * n + 2 * p
    print(p, cs.plan(), "evals/s %.3g" % ev, "TF(unpadded) %.1f" % (ev * fg / 1e12), flush=True)
