#!/usr/bin/env python3
"""9 <= p <= 32, register-resident n, HMC L=20: the planner's choice with precision="full" (register / LDS kernels on the
vector ALU) against the fused matrix-core kernel with bf16 interior steps (mode="mfma", rows split over 4 waves; 16 chains
per wave where that variant exists) -- chain-iterations/s, algorithmic TFLOP/s, acceptance."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
L = 20
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(200, 12), (200, 16), (200, 24), (200, 32), (500, 16), (500, 32), (1000, 12), (1000, 16)]
for n, p in shapes:
    X, y, _ = la.synthetic_logreg(n, p, seed=n + p, beta_sd=0.5 / np.sqrt(p))
    m = la.LogReg(X, y, np.ones(p))
    bmap, info = la.find_map(m)
    eps = 0.9 / np.sqrt(np.max(np.linalg.eigvalsh(info["hessian"]))) / p ** 0.25
    k = la.hmcKernel(m.lpost, m.glp, eps=eps, l=L, dmm=np.ones(p))
    fg = 4 * n * p + 5 * n + 2 * p
    for C in (1024, 4096, 16384):
        q0 = bmap + info["sd"] * np.random.default_rng(1).standard_normal((C, p))
        row = ["n=%d p=%d C=%d" % (n, p, C)]
        for mode, group, prec in (("auto", 0, "full"), ("mfma", 4, "auto"), ("mfma", 1, "auto")):
            try:
                cs = la.ChainSet(k, q0, seed=5, mode=mode, group=group, precision=prec)
                cs.advance(1, 3, keep=False); cs.sync()
            except Exception:  # no such variant for this width
                continue
            a0 = cs.get_accepts().sum()
            its = 20
            t0 = time.perf_counter(); cs.advance(1, its, keep=False); cs.sync(); dt = time.perf_counter() - t0
            pl = cs.plan()
            row.append("%s%d/%d %.3e it/s %.0f TF acc %.3f |" % (pl["mode"], pl["group"], pl["rows_per_lane"], C * its / dt, C * its * L * fg / dt / 1e12,
                                                              (cs.get_accepts().sum() - a0) / (its * C)))
        print(*row, flush=True)
