#!/usr/bin/env python3
"""Tall models with p = 8..32 on the stepwise engine: time per HMC leapfrog step, exact fp32 interior gradients
("full") against the bf16 matrix-pipe interior kernel of lr_tall_mx.h ("auto"), and the acceptance rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for n, p in ((100000, 8), (20000, 12), (20000, 24), (50000, 32), (5000, 30), (2000, 20)):
    X, y, _ = la.synthetic_logreg(n, p, seed=n + p, beta_sd=0.5 / np.sqrt(p))
    m = la.LogReg(X, y, np.ones(p))
    bmap, info = la.find_map(m)
    eps = 0.9 / np.sqrt(np.max(np.linalg.eigvalsh(info["hessian"]))) / p ** 0.25
    k = la.hmcKernel(m.lpost, m.glp, eps=eps, l=20, dmm=np.ones(p))
    q0 = bmap + info["sd"] * np.random.default_rng(1).standard_normal((C, p))
    row = [n, p]
    for prec in ("full", "auto"):
        cs = la.ChainSet(k, q0, seed=5, mode="stepwise", precision=prec)
        cs.advance(2, 1, keep=False); cs.sync()
        a0 = cs.get_accepts().sum()
        t0 = time.perf_counter(); cs.advance(6, 1, keep=False); cs.sync(); dt = time.perf_counter() - t0
        row += [prec, "us/step %.2f" % (dt / 120 * 1e6), "accept %.3f" % ((cs.get_accepts().sum() - a0) / (6 * C))]
    print(*row, cs.plan(), flush=True)
