#!/usr/bin/env python3
"""Per-launch time distribution of the headline workload's chain kernels (development tool): does a variant's launch time depend on
what ran before it, or vary between launches?   usage: launch_jitter.py [chains ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la

n, p, L, thin = 200, 8, 50, 20
X, y, _ = la.synthetic_logreg(n, p, seed=1234, beta_sd=0.5 / np.sqrt(p))
m = la.LogReg(X, y, np.ones(p))
bmap, info = la.find_map(m)
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=L, dmm=np.ones(p))
stream = None
if "--stream" in sys.argv:  # a created stream, as bench.py uses
    import ctypes as Ct
    from logreg_amd import _lib
    stream = Ct.c_void_p()
    _lib.check(_lib.load().lr_stream_create(0, Ct.byref(stream)))
for C in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4096]:
    q0 = bmap + 0.1 * 0.17 * np.random.default_rng(1).standard_normal((C, p))
    sets = {prec: la.ChainSet(k, q0, seed=5, precision=prec, stream=stream) for prec in ("auto", "full")}
    quick = "--quick" in sys.argv
    if quick:
        sets.pop("full")
    for rnd in range(1 if quick else 3):
        for prec, cs in sets.items():
            ts = []
            for b in range(40):
                t0 = time.perf_counter()
                for _ in range(10):
                    cs.advance(1, thin, keep=False)
                cs.sync()
                ts.append((time.perf_counter() - t0) / 10 * 1e3)
            ts = np.array(ts)
            print(f"C={C} round {rnd} {prec:4s} {cs.plan()}: ms per launch over 40 batches of 10: min {ts.min():.3f} median {np.median(ts):.3f} "
                  f"p90 {np.percentile(ts, 90):.3f} max {ts.max():.3f}   first 5 batches {np.round(ts[:5], 3)}", flush=True)
