#!/usr/bin/env python3
"""BASELINE config 4's shape (n = 100 000, p = 8, HMC L = 50, 1024 chains) on a float64 model: precision="full" (every evaluation
on lr_tall.h's float64 kernel) against the default policy (interior gradients on the bf16 matrix pipe, lr_tall_mx.h; position, momentum
and end points float64).  Development tool: us per evaluation of all chains."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la

fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg4.json")))
n, p, C = fix["n"], fix["p"], 1024
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix.get("beta_sd", 1.0))
for dtype in ("float64", "float32"):
    m = la.LogReg(X, y, np.array(fix["pscale"]), dtype=dtype)
    k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((C, p))
    for prec in ("full", "auto"):
        cs = la.ChainSet(k, q0, seed=5, precision=prec)
        cs.advance(1, 2, keep=False); cs.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); cs.advance(1, 4, keep=False); cs.sync(); best = min(best, (time.perf_counter() - t0) / 4)
        print(f"{dtype} {prec}: plan {cs.plan()}  {best * 1e6 / (fix['l'] + 1):8.2f} us per evaluation  {C / best:9.3e} it/s  accept {cs.get_accepts().sum() / (C * 14):.3f}", flush=True)
