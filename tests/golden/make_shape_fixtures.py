#!/usr/bin/env python3
"""Reference-generated vectors at shapes OTHER than Pima's (n = 200, p = 8): BASELINE configs 4 and 5 at full size
and one mid shape.  They pin the oracle (and, through the C ABI, the device kernels) at those sizes with numbers that
came out of the reference's own closures, not out of the oracle "by extension".

How.  The reference's model closures read the module globals X, y, pscale (Python/fit-np-hmc.py:23-47), and
hmcKernel's leapf reads nothing but its arguments and glpi (fit-np-hmc.py:67-75).  So the head of the script is
exec-ed exactly as tests/golden/make_fixtures.py does (load_reference), and then ns["X"], ns["y"], ns["pscale"] are
replaced by a synthetic design (logreg_amd.data.synthetic_logreg: NumPy Philox, no GPU): ll / lprior / lpost / glp
and leapf then run the reference's code on the new data.  Only numbers are written; no reference source is stored.

    python tests/golden/make_shape_fixtures.py          # needs /root/reference (build container only), ~20 s

Writes tests/golden/shape_<name>.json with
    beta [8, p]; ll, lprior, lpost [8]; glp [8, p]                      (F1-type, fit-np-hmc.py:23-24, 33-34, 36-37, 44-47)
    leap: q0, p0 -> q1, p1_negated, alpi0, alpi1 with eps, l, dmm        (F3-type, fit-np-hmc.py:66-78)
The designs are regenerated from (n, p, data_seed, beta_sd) by the tests; they are not stored.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from logreg_amd.data import synthetic_logreg  # noqa: E402  (NumPy only)
from make_fixtures import REF, closure_of, jdump, load_reference  # noqa: E402

SHAPES = {
    # name: design + HMC settings.  cfg4 / cfg5 take the design, prior, step size and expansion point (MAP, Laplace sd) of
    # the committed full-size fixtures, so the points are where the samplers actually work
    "cfg4": dict(fullsize=4),
    "cfg5": dict(fullsize=5),
    "mid": dict(n=1000, p=20, data_seed=20240020, beta_sd=0.3, pscale=[10.0] + [1.0] * 19, eps=0.02, l=50),
}


def newton_map(X, y, pscale, iters=50):
    b = np.zeros(X.shape[1])
    iv = 1.0 / np.asarray(pscale) ** 2
    H = None
    for _ in range(iters):
        mu = 1.0 / (1.0 + np.exp(-(X @ b)))
        g = X.T @ (y - mu) - b * iv
        H = (X * (mu * (1 - mu))[:, None]).T @ X + np.diag(iv)
        step = np.linalg.solve(H, g)
        b = b + step
        if np.max(np.abs(step)) < 1e-13:
            break
    return b, np.sqrt(np.diag(np.linalg.inv(H)))


def make(name: str):
    c = dict(SHAPES[name])
    if "fullsize" in c:
        fix = json.load(open(os.path.join(HERE, f"fullsize_cfg{c['fullsize']}.json")))
        c = dict(n=fix["n"], p=fix["p"], data_seed=fix["data_seed"], beta_sd=fix["beta_sd"], pscale=fix["pscale"],
                 eps=fix["eps"], l=fix["l"], center=np.array(fix["map"]), spread=np.array(fix["laplace_sd"]))
    X, y, _ = synthetic_logreg(c["n"], c["p"], seed=c["data_seed"], beta_sd=c["beta_sd"])
    pscale = np.asarray(c["pscale"], dtype=np.float64)
    if "center" not in c:
        c["center"], c["spread"] = newton_map(X, y, pscale)
    ns = load_reference("hmc", 1)
    # the reference's data block, replaced: X float64 [n, p] with the intercept column, y float32 in {0, 1} as
    # fit-np-hmc.py:17 has it, pscale float64 [p]
    ns["X"], ns["y"], ns["pscale"] = X, y.astype(np.float32), pscale
    ns["n"], ns["p"] = X.shape
    p = c["p"]
    rng = np.random.Generator(np.random.Philox(7000 + len(name) + p))
    betas = [c["center"].copy(), np.zeros(p)]
    for k in range(6):
        betas.append(c["center"] + (1.0 if k < 3 else 4.0) * c["spread"] * rng.standard_normal(p))
    betas = np.array(betas)
    rows = {"ll": [], "lprior": [], "lpost": [], "glp": []}
    for b in betas:
        for nm in rows:
            rows[nm].append(ns[nm](b))
    dmm = np.ones(p)
    kern = ns["hmcKernel"](ns["lpost"], ns["glp"], eps=c["eps"], l=c["l"], dmm=dmm)
    mhk = closure_of(kern, "mhk")
    alpi = closure_of(mhk, "lpost")
    leapf = closure_of(closure_of(mhk, "rprop"), "leapf")
    q0 = c["center"] + c["spread"] * rng.standard_normal(p)
    p0 = rng.standard_normal(p) * np.sqrt(dmm)
    q1, p1 = leapf(q0, p0)
    jdump(f"shape_{name}.json", {
        "source": "ll/lprior/lpost/glp closures (Python/fit-np-hmc.py:23-24,33-34,36-37,44-47) and the leapf/alpi closure cells of "
                  "hmcKernel (fit-np-hmc.py:65-78), run on a synthetic design put in place of the script's globals X, y, pscale; "
                  "tests/golden/make_shape_fixtures.py",
        "n": c["n"], "p": p, "data_seed": c["data_seed"], "beta_sd": c["beta_sd"], "pscale": pscale,
        "beta": betas, **{k: np.array(v) for k, v in rows.items()},
        "leap": {"eps": c["eps"], "l": c["l"], "dmm": dmm, "q0": q0, "p0": p0, "q1": q1, "p1_negated": p1,
                 "alpi0": alpi((q0, p0)), "alpi1": alpi((q1, p1))},
    })


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    for nm in (sys.argv[1:] or list(SHAPES)):
        make(nm)
