#!/usr/bin/env python3
"""Run a few HMC launches of one shape (for rocprofv3):  run_shape.py n p chains mode group [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
n, p, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode, group = sys.argv[4], int(sys.argv[5])
L = int(sys.argv[6]) if len(sys.argv) > 6 else 20
X, y, _ = la.synthetic_logreg(n, p, seed=n + p, beta_sd=0.5 / np.sqrt(p))
m = la.LogReg(X, y, np.ones(p))
k = la.hmcKernel(m.lpost, m.glp, eps=0.3 / np.sqrt(n), l=L, dmm=np.ones(p))
cs = la.ChainSet(k, 0.02 * np.random.default_rng(1).standard_normal((C, p)), seed=5, mode=mode, group=group)
for _ in range(4):
    cs.advance(1, 5, keep=False)
cs.sync()
print(cs.plan(), cs.get_accepts().mean() / 20)
