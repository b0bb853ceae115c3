#!/usr/bin/env python3
"""Wide partial kernel: time per leapfrog step versus rows per workgroup (fixed overhead vs per-block cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
p, c = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for n in (512, 1024, 2048, 4096, 8192, 16384):
    X, y, _ = la.synthetic_logreg(n, p, seed=n, beta_sd=0.05)
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=50, dmm=np.ones(p))
    cs = la.ChainSet(k, np.zeros((c, p)), seed=5)
    cs.advance(2, 1, keep=False); cs.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); cs.advance(6, 1, keep=False); cs.sync(); best = min(best, time.perf_counter() - t0)
    plan = cs.plan()
    print(n, plan, "us/step %.2f" % (best / 300 * 1e6), "rows/WG", plan["rows_per_lane"], flush=True)
