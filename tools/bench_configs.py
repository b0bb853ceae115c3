#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on ONE MI355X (they are parity-test cases, not
bench.py lines; numbers land in DESIGN.md / profiles/).

    python tools/bench_configs.py [1 3 4]

config 1: RWMH, 1 chain, 10000 kept x thin 1000 on Pima   (the reference's fit-numpy.py run)
config 3: MALA dt=1e-5, pre=[100,1,..,25,1], thin 1000, 8192 chains (= one GPU's shard of 65536)
config 4: HMC L=50 on synthetic n=100000, p=8, 1024 chains (tall data, rows streamed)
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logreg_amd as la  # noqa: E402

PSCALE = np.array([10.0, 1, 1, 1, 1, 1, 1, 1])
PRE = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
MAP = np.array([-9.19131622, 0.09705401, 0.03112265, -0.00564495, -0.00062272, 0.0814371, 1.26032561, 0.03939102])


def timed(cs, iters, thin, warm=1, repeats=1):
    """Wall time of `iters` kept iterations; repeats > 1: the best of that many (short measurements)."""
    for _ in range(warm):
        cs.advance(1, thin, keep=False)
    cs.sync()
    best, out = None, None
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = cs.advance(iters, thin)
        cs.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out


def config1():
    X, y = la.load_pima()
    m = la.LogReg(X, y, PSCALE)
    k = la.mhKernel(m.lpost, la.rwProposal(0.02 * np.array([10.0, 1, 1, 1, 1, 1, 5, 1])))
    t0 = time.perf_counter()
    out, info = la.mcmc(MAP, k, thin=1000, iters=10000, verb=False, seed=1, return_info=True)
    dt = time.perf_counter() - t0
    s = la.summarise(out)
    return {"config": 1, "what": "RWMH 1 chain x 1e7 iterations (fit-numpy.py:86)", "wall_s": dt, "it_per_s": 1e7 / dt,
            "plan": info["plan"], "accept": float(info["accepts"][0]) / 1e7, "mean": s["mean"].round(4).tolist(),
            "min_ess_per_s": float(s["ess"].min() / dt), "reference_wall_s": 1541.8}


def config3(chains=8192):
    X, y = la.load_pima()
    m = la.LogReg(X, y, PSCALE)
    k = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE)
    res = []
    for mode, group in (("auto", 0), ("reg", 16), ("reg", 32), ("mfma", 1), ("mfma", 4)):
        cs = la.ChainSet(k, np.tile(MAP, (chains, 1)), seed=3, group=group, mode=mode)
        dt, _ = timed(cs, 4, 1000)
        acc = cs.get_accepts().sum() / (chains * 5000)
        res.append({"plan": cs.plan(), "it_per_s": chains * 4000 / dt, "accept": float(acc)})
    return {"config": 3, "what": f"MALA thin 1000, {chains} chains (one GPU's shard)", "variants": res}


def config4(chains=1024, n=100000):
    X, y, _ = la.synthetic_logreg(n, 8, seed=20240004)
    m = la.LogReg(X, y, PSCALE)
    init = np.zeros(8)
    k = la.hmcKernel(m.lpost, m.glp, eps=1e-3 / 22, l=50, dmm=np.ones(8) * (n / 200.0))
    cs = la.ChainSet(k, np.tile(init, (chains, 1)), seed=4)
    dt, _ = timed(cs, 2, 1, warm=1, repeats=3)
    its = chains * 2
    return {"config": 4, "what": f"HMC L=50, n={n}, p=8, {chains} chains", "plan": cs.plan(), "it_per_s": its / dt,
            "grad_evals_per_s": its * 50 / dt, "x_pass_GBps_per_eval_stream": its * 50 * n * 8 * 4 / dt / 1e9,
            "accept": float(cs.get_accepts().sum() / (chains * 7))}


def config5(chains=1024, n=4096, p=128):
    X, y, _ = la.synthetic_logreg(n, p, seed=20240005, beta_sd=0.1)
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=5e-3, l=50, dmm=np.ones(p))
    cs = la.ChainSet(k, np.zeros((chains, p)), seed=5)
    dt, _ = timed(cs, 4, 1, warm=1, repeats=3)
    its = chains * 4
    fg = 4 * n * p + 5 * n + 2 * p
    return {"config": 5, "what": f"HMC L=50, n={n}, p={p}, {chains} chains (one GPU's shard of 8192)", "plan": cs.plan(),
            "it_per_s": its / dt, "grad_evals_per_s": its * 50 / dt, "tflops_algorithmic": its * 50 * fg / dt / 1e12,
            "accept": float(cs.get_accepts().sum() / (chains * 13))}


if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or [1, 3, 4, 5]
    for c in which:
        print(json.dumps({1: config1, 3: config3, 4: config4, 5: config5}[c]()), flush=True)
