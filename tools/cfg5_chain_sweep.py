#!/usr/bin/env python3
"""Config 5 design (n = 4096, p = 128) at growing chain counts on one GPU: time per evaluation of all chains and
algorithmic TFLOP/s (F_g = 4np + 5n + 2p per chain and evaluation), default interior-gradient policy vs "full"."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg5.json")))
n, p = fix["n"], fix["p"]
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
m = la.LogReg(X, y, np.array(fix["pscale"]))
k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
fg = 4 * n * p + 5 * n + 2 * p
for C in (1024, 2048, 4096, 8192, 16384):
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((C, p))
    row = [C]
    for prec in ("auto", "full"):
        cs = la.ChainSet(k, q0, seed=3, precision=prec)
        cs.advance(1, 1, keep=False); cs.sync()
        a0 = cs.get_accepts().sum()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); cs.advance(2, 1, keep=False); cs.sync(); best = min(best, time.perf_counter() - t0)
        per_eval = best / (2 * fix["l"])
        row += [prec, "us/eval %.2f" % (per_eval * 1e6), "TF %.0f" % (C * fg / per_eval / 1e12), "accept %.3f" % ((cs.get_accepts().sum() - a0) / (4 * C))]
    print(*row, flush=True)
