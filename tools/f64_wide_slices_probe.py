#!/usr/bin/env python3
"""Config 5's shape on a float64 model, every evaluation on the f64 matrix pipe (precision="full"): row slices per evaluation (the
planner's choice is one workgroup per CU).  Development tool: us per evaluation of all chains by forced slice count."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la

fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg5.json")))
n, p = fix["n"], fix["p"]
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
m = la.LogReg(X, y, np.array(fix["pscale"]), dtype="float64")
k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
for C in (1024, 4096):
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((C, p))
    for prec in ("full", "auto"):
        for group in (0, 8, 16, 24, 32, 48, 64):
            cs = la.ChainSet(k, q0, seed=5, precision=prec, mode="stepwise" if group else "auto", group=group)
            cs.advance(1, 1, keep=False); cs.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); cs.advance(1, 2, keep=False); cs.sync(); best = min(best, (time.perf_counter() - t0) / 2)
            print(f"C={C} {prec} group={group}: plan {cs.plan()}  {best * 1e6 / (fix['l'] + 1):8.2f} us per evaluation", flush=True)
