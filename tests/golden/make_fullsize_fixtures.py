#!/usr/bin/env python3
"""Golden posteriors for BASELINE.json configs 4 and 5 at their FULL sizes (SURVEY.md section 8(d) table:
"means/SDs vs fp64 CPU restatement").

Unlike make_fixtures.py (which imports the reference's NumPy scripts), the numbers here come from the
float64 C oracle (oracle/lr_oracle.c) -- itself pinned against the reference by tests/test_oracle.py --
because the reference ships no data at these sizes: the designs are the synthetic ones SURVEY.md section
8(d) defines (generator: logreg_amd.data.synthetic_logreg, NumPy Philox, no GPU involved).  The long
oracle runs take minutes on 8 cores, so they are done once here and committed as numbers; the GPU tests
(tests/test_gpu_fullsize.py) re-run the oracle only on a 64-chain subset for step-level parity.

    python tests/golden/make_fullsize_fixtures.py [2] [4] [5]

Writes tests/golden/fullsize_cfg2.json (the synthetic n = 200, p = 8 design of bench.py's headline, at the bench's own
eps = 0.1, L = 50) / fullsize_cfg4.json / fullsize_cfg5.json:
    map, laplace_sd           Newton MAP (float64 NumPy) and sqrt(diag((-H)^-1))
    eps, l, dmm               HMC settings, eps tuned to an acceptance rate inside 0.6-0.95
    accept, accept_se         acceptance rate of the long oracle run
    mean, sd                  pooled posterior summary of the long oracle run (after `dropped` warm-up iterations)
    mcse, se_sd               their standard errors, from the spread between the independent chains
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from logreg_amd.data import synthetic_logreg  # noqa: E402  (NumPy only)
from logreg_amd.diagnostics import summarise  # noqa: E402  (NumPy only)
from oracle.oracle import OracleModel, max_threads  # noqa: E402

CONFIGS = {
    # the synthetic design bench.py's `value` is measured on (BASELINE.json configs[1]: n = 200, p = 8, seed 20240001), at the
    # bench's own settings: eps = 0.1 (fixed, not tuned), L = 50, unit mass
    2: dict(n=200, p=8, seed=20240001, beta_sd=0.5, pscale=[10.0] + [1.0] * 7, l=50, chains=256, iters=4100, drop=100,
            eps_grid=[0.1]),
    4: dict(n=100000, p=8, seed=20240004, beta_sd=0.5, pscale=[10.0] + [1.0] * 7, l=50, chains=128, iters=250, drop=50,
            eps_grid=[0.004, 0.006, 0.008, 0.010]),
    5: dict(n=4096, p=128, seed=20240005, beta_sd=0.1, pscale=[1.0] * 128, l=50, chains=128, iters=400, drop=50,
            eps_grid=[0.008, 0.012, 0.016, 0.020]),
}


def newton_map(X, y, pscale, iters=50):
    n, p = X.shape
    b = np.zeros(p)
    iv = 1.0 / np.asarray(pscale) ** 2
    H = None
    for _ in range(iters):
        eta = X @ b
        mu = 1.0 / (1.0 + np.exp(-eta))
        g = X.T @ (y - mu) - b * iv
        H = (X * (mu * (1 - mu))[:, None]).T @ X + np.diag(iv)
        step = np.linalg.solve(H, g)
        b = b + step
        if np.max(np.abs(step)) < 1e-13:
            break
    return b, np.sqrt(np.diag(np.linalg.inv(H)))


def make(cfg_id: int):
    c = CONFIGS[cfg_id]
    X, y, _ = synthetic_logreg(c["n"], c["p"], seed=c["seed"], beta_sd=c["beta_sd"])
    ps = np.asarray(c["pscale"])
    orc = OracleModel(X, y, ps)
    bmap, lsd = newton_map(X, y, ps)
    p, thr = c["p"], max_threads()
    dmm = np.ones(p)
    rng = np.random.Generator(np.random.Philox(1000 + cfg_id))
    init = bmap + lsd * rng.standard_normal((c["chains"], p))
    # tune eps on a short run: the largest grid value whose acceptance stays >= 0.75
    best = None
    for eps in c["eps_grid"]:
        r = orc.run("hmc", init[:2 * thr], step=eps, l=c["l"], scale=dmm, thin=1, iters=12, seed=7, keep=False, threads=thr)
        acc = r["accepts"].sum() / (2 * thr * 12)
        print(f"cfg {cfg_id}: eps={eps} accept={acc:.3f}", flush=True)
        if acc >= 0.75 or len(c["eps_grid"]) == 1:
            best = eps
    eps = best if best is not None else c["eps_grid"][0]
    t0 = time.time()
    r = orc.run("hmc", init, step=eps, l=c["l"], scale=dmm, thin=1, iters=c["iters"], seed=20240000 + cfg_id, threads=thr)
    dt = time.time() - t0
    out = r["out"][c["drop"]:]  # starts are Laplace draws about the MAP; the posterior mean sits up to 0.2 sd away
    s = summarise(out, max_chains=None)
    # Standard errors from the spread BETWEEN the independent chains (no autocorrelation estimate involved: Geyer's
    # truncation on chains of a few hundred draws overstates the ESS): per-chain means m_c and second moments
    # v_c = mean((x - pooled mean)^2);  se(mean) = sd_c(m_c)/sqrt(C),  se(sd) = sd_c(v_c)/sqrt(C) / (2 sd).
    C = out.shape[1]
    m_c = out.mean(axis=0)
    v_c = ((out - s["mean"]) ** 2).mean(axis=0)
    mcse = m_c.std(axis=0, ddof=1) / np.sqrt(C)
    se_sd = v_c.std(axis=0, ddof=1) / np.sqrt(C) / (2 * s["sd"])
    ndraw = c["chains"] * c["iters"]
    acc = float(r["accepts"].sum() / ndraw)
    fix = {"config": cfg_id, "n": c["n"], "p": p, "data_seed": c["seed"], "beta_sd": c["beta_sd"], "pscale": c["pscale"],
           "eps": eps, "l": c["l"], "dmm": dmm.tolist(), "map": bmap.tolist(), "laplace_sd": lsd.tolist(),
           "oracle_chains": c["chains"], "oracle_iters": c["iters"], "oracle_seed": 20240000 + cfg_id,
           "accept": acc, "accept_se": float(np.sqrt(acc * (1 - acc) / ndraw)), "mean": s["mean"].tolist(),
           "sd": s["sd"].tolist(), "ess_geyer": s["ess"].tolist(), "mcse": mcse.tolist(), "se_sd": se_sd.tolist(), "dropped": c["drop"],
           "source": "oracle/lr_oracle.c float64, orc_run HMC on the Philox stream; tests/golden/make_fullsize_fixtures.py",
           "oracle_wall_s": dt}
    path = os.path.join(HERE, f"fullsize_cfg{cfg_id}.json")
    with open(path, "w") as f:
        json.dump(fix, f, indent=1)
    print(f"cfg {cfg_id}: eps={eps} accept={acc:.3f} min ESS={s['ess'].min():.0f} of {out.shape[0] * out.shape[1]} "
          f"draws, {dt:.0f} s -> {path}", flush=True)


if __name__ == "__main__":
    for cid in ([int(a) for a in sys.argv[1:]] or [4, 5]):
        make(cid)
