#!/usr/bin/env python3
"""Planner crossovers re-measured at sustained clocks (0.2 s of load first), HMC L=50: the all-fp32 choice against the
matrix-core chain kernel the default policy would pick.  usage: planner_check.py [nxp ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
L = 50
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(200, 12), (200, 16), (200, 24), (200, 32), (500, 16), (200, 8), (400, 8)]
for n, p in shapes:
    X, y, _ = la.synthetic_logreg(n, p, seed=n + p, beta_sd=0.5 / np.sqrt(p))
    m = la.LogReg(X, y, np.ones(p))
    bmap, info = la.find_map(m)
    eps = 0.9 / np.sqrt(np.max(np.linalg.eigvalsh(info["hessian"]))) / p ** 0.25
    k = la.hmcKernel(m.lpost, m.glp, eps=eps, l=L, dmm=np.ones(p))
    for C in (1024, 2048, 4096):
        q0 = bmap + info["sd"] * np.random.default_rng(1).standard_normal((C, p))
        row = ["n=%d p=%d C=%d" % (n, p, C)]
        for prec in ("full", "auto"):
            cs = la.ChainSet(k, q0, seed=5, precision=prec)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.2:
                for _ in range(5): cs.advance(1, 20, keep=False)
                cs.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(5): cs.advance(1, 20, keep=False)
                cs.sync(); best = min(best, time.perf_counter() - t0)
            pl = cs.plan()
            row.append("%s: %s%d/%d %.3e it/s |" % (prec, pl["mode"], pl["group"], pl["rows_per_lane"], C * 100 / best))
        print(*row, flush=True)
