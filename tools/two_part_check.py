#!/usr/bin/env python3
"""Two-part plans against the single launch they replace, at launch lengths a run actually uses (HMC L=50 thin 20; MALA / RWMH thin
1000): chain-iterations/s of AUTO and of the forced head variant alone, per kernel family and chain count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la

X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7))
kernels = {"hmc": (la.hmcKernel(m.lpost, m.glp, eps=0.1, l=50, dmm=np.ones(8)), 20),
           "mala": (la.malaKernel(m.lpost, m.glp, dt=2e-3, pre=np.ones(8)), 1000),
           "rwmh": (la.mhKernel(m.lpost, la.rwProposal(0.05 * np.ones(8))), 1000)}


def rate(cs, C, thin):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        cs.advance(1, thin, keep=False); cs.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            cs.advance(1, thin, keep=False)
        cs.sync()
        best = min(best, (time.perf_counter() - t0) / 5)
    return C * thin / best, best


print("# tools/two_part_check.py: AUTO (two-part where planned) | forced reg 16 single launch; it/s and ms per launch")
for kind, (k, thin) in kernels.items():
    for C in (4608, 5120, 6144, 9216, 10240, 13312):
        q0 = 0.017 * np.random.default_rng(1).standard_normal((C, 8))
        a = la.ChainSet(k, q0, seed=5, precision="full")
        f = la.ChainSet(k, q0, seed=5, precision="full", mode="reg", group=16)
        ra, ta = rate(a, C, thin)
        rf, tf = rate(f, C, thin)
        pl = a.plan()
        print(f"{kind:5s} {C:6d}: AUTO {pl['group']}/{pl['rows_per_lane']}{'+' + pl['tail']['mode'] + str(pl['tail']['group']) + '@' + str(pl['tail']['from']) if 'tail' in pl else '':13s} "
              f"{ra:.3e} ({ta * 1e3:.3f} ms) | reg16 {rf:.3e} ({tf * 1e3:.3f} ms)  ratio {ra / rf:.3f}", flush=True)
