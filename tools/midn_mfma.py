#!/usr/bin/env python3
"""Usage: midn_mfma.py [n ...].  Mid-size data (256 < n <= 1024, p = 8), HMC L=20: the planner's register variant against the fused matrix-core
kernel with the rows split over 4 waves (mode="mfma", group=4) -- chain-iterations/s and acceptance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
for n in ([int(a) for a in sys.argv[1:]] or [400, 512, 700, 1000]):
    X, y, _ = la.synthetic_logreg(n, 8, seed=n)
    m = la.LogReg(X, y, np.ones(8))
    bmap, info = la.find_map(m)
    eps = 0.9 / np.sqrt(np.max(np.linalg.eigvalsh(info["hessian"]))) / 8 ** 0.25
    k = la.hmcKernel(m.lpost, m.glp, eps=eps, l=20, dmm=np.ones(8))
    for C in (1024, 2048, 4096, 8192, 16384):
        q0 = bmap + info["sd"] * np.random.default_rng(1).standard_normal((C, 8))
        row = [n, C]
        for mode, group, prec in (("reg", 0, "full"), ("mfma", 4, "auto")):
            cs = la.ChainSet(k, q0, seed=5, mode=mode, group=group, precision=prec)
            cs.advance(1, 5, keep=False); cs.sync()
            a0 = cs.get_accepts().sum()
            t0 = time.perf_counter(); cs.advance(1, 40, keep=False); cs.sync(); dt = time.perf_counter() - t0
            row += [cs.plan()["mode"], cs.plan()["group"], "it/s %.3e" % (C * 40 / dt), "acc %.3f" % ((cs.get_accepts().sum() - a0) / (40 * C))]
        print(*row, flush=True)
