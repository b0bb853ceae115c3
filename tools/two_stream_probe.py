#!/usr/bin/env python3
"""Does splitting the chains of a stepwise-engine run over K streams hide the per-evaluation launch/ramp cost?
Config 5 (n=4096, p=128) and config 4 (n=100000, p=8), 1024 chains in all, K = 1, 2, 4 concurrent ChainSets."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
L = _lib.load()
for cfg in (5, 4):
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"fullsize_cfg{cfg}.json")))
    n, p, Ctot = fix["n"], fix["p"], 1024
    X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
    m = la.LogReg(X, y, np.array(fix["pscale"]))
    k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((Ctot, p))
    for K in (1, 2, 4):
        sets = []
        for i in range(K):
            s = C.c_void_p(); _lib.check(L.lr_stream_create(0, C.byref(s)))
            lo, hi = i * Ctot // K, (i + 1) * Ctot // K
            sets.append(la.ChainSet(k, q0[lo:hi], seed=3, chain_offset=lo, stream=s))
        for cs in sets: cs.advance(1, 1, keep=False)
        for cs in sets: cs.sync()
        t0 = time.perf_counter()
        for cs in sets: cs.advance(4, 1, keep=False)
        for cs in sets: cs.sync()
        dt = time.perf_counter() - t0
        print(f"cfg {cfg}: {K} stream(s) x {Ctot // K} chains: {dt / (4 * fix['l']) * 1e6:.2f} us per evaluation of all {Ctot} chains", sets[0].plan(), flush=True)
