#!/usr/bin/env python3
"""Configs 1 and 3 quick timing: RWMH 1 chain x 2e6 iterations; MALA thin 1000, 8192 chains x 4 kept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
X, y = la.load_pima()
m = la.LogReg(X, y, np.array([10.0, 1, 1, 1, 1, 1, 1, 1]))
MAP = np.array([-9.19131622, 0.09705401, 0.03112265, -0.00564495, -0.00062272, 0.0814371, 1.26032561, 0.03939102])
k = la.mhKernel(m.lpost, la.rwProposal(0.02 * np.array([10.0, 1, 1, 1, 1, 1, 5, 1])))
km = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=np.array([100.0, 1, 1, 1, 1, 1, 25, 1]))
cs = la.ChainSet(k, MAP, seed=1)
cs.advance(1, 1000, keep=False); cs.sync()
t0 = time.perf_counter(); cs.advance(2, 1000000, keep=False); cs.sync(); dt = time.perf_counter() - t0
print("config 1 RWMH 1 chain: it/s %.3e" % (2e6 / dt), cs.plan(), "accept", cs.get_accepts()[0] / 2.001e6, flush=True)
for C in (8192, 65536):
    for kern, name in ((km, "mala"), (k, "rwmh")):
        cs = la.ChainSet(kern, np.tile(MAP, (C, 1)), seed=3)
        cs.advance(1, 1000, keep=False); cs.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); cs.advance(2, 1000, keep=False); cs.sync(); best = min(best, time.perf_counter() - t0)
        print(f"{name} {C} chains: chain-it/s %.3e" % (C * 2000 / best), cs.plan(), "accept %.4f" % (cs.get_accepts().sum() / (C * 7000)), flush=True)
