#!/usr/bin/env python3
"""Headline workload (HMC L=50, n=200, p=8, thin 20) across chain counts: the planner's choice and chain-iterations/s under both
precision policies -- where the wave quantisation of the register kernel shows and what AUTO does about it (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la

n, p, L, thin = 200, 8, 50, 20
X, y, _ = la.synthetic_logreg(n, p, seed=20240001)
dtype = "float32"
if sys.argv[1:2] == ["f64"]:  # python tools/chain_grid.py f64 [chain counts]: the float64 model
    dtype = "float64"
    del sys.argv[1]
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7), dtype=dtype)
init = np.array([-0.65920504, -0.18123564, -0.64985465, -0.19187958, -0.11223836, -0.51230749, -0.10401207, -0.8432688])
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=L, dmm=np.ones(p))
print(f"# tools/chain_grid.py on one MI355X, {dtype} model: chains, per policy the planned variant, ms per launch of 20 iterations, chain-iterations/s")
for C in [int(a) for a in sys.argv[1:]] or [1024, 2048, 2560, 3072, 4096, 5120, 6144, 8192, 10240, 12288, 16384, 24576, 32768, 65536]:
    q0 = init + 0.017 * np.random.default_rng(1).standard_normal((C, p))
    row = f"{C:6d}"
    for prec in ("full", "auto"):
        cs = la.ChainSet(k, q0, seed=5, precision=prec)
        t0 = time.perf_counter()
        nw = 0
        while time.perf_counter() - t0 < 0.1:
            cs.advance(1, thin, keep=False); cs.sync(); nw += 1
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                cs.advance(1, thin, keep=False)
            cs.sync()
            best = min(best, (time.perf_counter() - t0) / 10)
        pl = cs.plan()
        tail = f"+{pl['tail']['mode']}{pl['tail']['group']}/{pl['tail']['rows_per_lane']}@{pl['tail']['from']}" if "tail" in pl else ""
        row += f"   {prec}: {pl['mode']}{pl['group']}/{pl['rows_per_lane']}{tail:<12s} {best * 1e3:7.3f} ms {C * thin / best:9.3e}"
    print(row, flush=True)
