#!/usr/bin/env python3
"""The reference's Python/fit-np-hmc.py, end to end, on the drop-in (needs an MI355X).

Same stages as the reference script: data block -> model closures -> MAP warm start -> HMC with
eps=1e-3, l=50, dmm=1/pre, thin=20 -> parquet b0..b7 -> summary.  `--chains C` runs C chains.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from logreg_amd import (LogReg, find_map, hmcKernel, load_pima, mcmc, print_summary, summarise,  # noqa: E402
                        write_parquet)

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=1)
ap.add_argument("--iters", type=int, default=10000)
ap.add_argument("--out", default="fit-np-hmc.parquet")
a = ap.parse_args()

X, y = load_pima()                                   # fit-np-hmc.py:12-19
n, p = X.shape
pscale = np.array([10., 1., 1., 1., 1., 1., 1., 1.])  # :31
model = LogReg(X, y, pscale)
ll, lprior, lpost, glp = model.ll, model.lprior, model.lpost, model.glp   # :23-47

init = np.random.randn(p) * 0.1                      # :26
print("MAP:")
res_x, info = find_map(LogReg(X, y, pscale, dtype="float64"), init)      # :49 (Newton on the device instead of SciPy BFGS)
print(res_x)
print(ll(res_x))
print(glp(res_x))

print("HMC:")
pre = np.array([100., 1., 1., 1., 1., 1., 25., 1.])   # :105
start = res_x if a.chains == 1 else np.tile(res_x, (a.chains, 1))
out = mcmc(start, hmcKernel(lpost, glp, eps=1e-3, l=50, dmm=1 / pre), thin=20, iters=a.iters)   # :107-108

print(out)
write_parquet(out, a.out)                            # :111-112
print_summary(out)                                   # :113-117
s = summarise(out)
print("ESS:", np.round(s["ess"]), "MCSE:", s["mcse"])
