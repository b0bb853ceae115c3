#!/usr/bin/env python3
"""Mid-size n (between the register and the tall regime): what AUTO picks and how the engines compare."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
p, C, L = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 20
for n in (300, 500, 1000, 2000, 4000, 6000, 8000, 12000):
    X, y, _ = la.synthetic_logreg(n, p, seed=n)
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=0.5 / np.sqrt(n), l=L, dmm=np.ones(p))
    res = []
    for mode, group in (("auto", 0), ("lds", 8), ("lds", 64), ("global", 64), ("stepwise", 0)):
        try:
            cs = la.ChainSet(k, np.zeros((C, p)), seed=5, mode=mode, group=group)
            cs.advance(1, 1, keep=False); cs.sync()
            t0 = time.perf_counter(); cs.advance(4, 1, keep=False); cs.sync(); dt = time.perf_counter() - t0
            pl = cs.plan()
            res.append("%s/%d->%s%d: %.3g" % (mode, group, pl["mode"][:4], pl["group"], C * 4 * L * n / dt))
        except la.LogregHipError as e:
            res.append("%s/%d: n/a" % (mode, group))
    print(n, " | ".join(res), "(chain-rows/s)", flush=True)
