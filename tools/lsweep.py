#!/usr/bin/env python3
"""Per-iteration fixed cost against per-leapfrog-step cost of the headline workload (n=200, p=8, 4096 chains):
launches of 20 iterations at several L; time = a + b L."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q0 = 0.05 * np.random.default_rng(1).standard_normal((C, 8))
for prec in ("full", "auto"):
    pts = []
    for L in (1, 2, 10, 25, 50, 100):
        k = la.hmcKernel(m.lpost, m.glp, eps=0.1 * 50 / max(L, 50) / 10, l=L, dmm=np.ones(8))
        cs = la.ChainSet(k, q0, seed=5, precision=prec)
        cs.advance(1, 20, keep=False); cs.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); cs.advance(1, 20, keep=False); cs.sync(); best = min(best, time.perf_counter() - t0)
        pts.append((L, best / 20 * 1e6))
        print(prec, cs.plan(), "L=%d: %.3f us per iteration of all chains" % (L, best / 20 * 1e6), flush=True)
    (L1, t1), (L2, t2) = pts[2], pts[-1]
    b = (t2 - t1) / (L2 - L1)
    print(prec, "per leapfrog step %.4f us, fixed per iteration %.3f us (= %.1f steps)" % (b, t2 - b * L2, (t2 - b * L2) / b))
