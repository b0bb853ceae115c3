#!/usr/bin/env python3
"""k_wide_traj2_bf16 (two chain tiles per workgroup) against k_wide_traj_bf16: the same chains bit for bit (ragged chain counts
included), and the time per evaluation of both at config 5's design."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la

fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg5.json")))
n, p = fix["n"], fix["p"]
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
outs = {}
for opt in ("wide_traj=1", "wide_traj=2"):
    os.environ["LOGREG_DEBUG_OPTS"] = opt
    m = la.LogReg(X, y, np.array(fix["pscale"]))
    k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    for C in (100, 1000):
        rng = np.random.Generator(np.random.Philox(4005))
        q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * rng.standard_normal((C, p))
        out, info = la.mcmc(q0, k, thin=1, iters=3, verb=False, seed=5, return_info=True)
        outs[(opt, C)] = (out, info["accepts"].copy())
        print(opt, C, info["plan"], "accept", info["accepts"].mean() / 3, flush=True)
for C in (100, 1000):
    a, b = outs[("wide_traj=1", C)], outs[("wide_traj=2", C)]
    print("chains", C, "bit-identical:", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])), "max|diff|", float(np.max(np.abs(a[0] - b[0]))))
