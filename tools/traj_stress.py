#!/usr/bin/env python3
"""Stress of one scenario: fresh float64 wide model -> seeded run -> the same run in chunks; counts mismatches (tools/traj_determinism.py's
little brother, for a rare event: one chunk mismatch was seen once in ~26 runs of tests/test_gpu_parity.py's float64 trajectory test).
    python3 tools/traj_stress.py [repeats] [dtype] [p] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["LOGREG_DEBUG_OPTS"] = os.environ.get("LOGREG_DEBUG_OPTS", "wide_traj=2")
import logreg_amd as la

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dtype = sys.argv[2] if len(sys.argv) > 2 else "float64"
p, n = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 500)
X, y, _ = la.synthetic_logreg(n, p, seed=905 + p, beta_sd=0.1)
b = 0.1 * np.random.default_rng(p + 1).standard_normal((600, p))
kw = dict(thin=1, iters=2, verb=False, seed=12)
first, bad = None, []
for rep in range(reps):
    m = la.LogReg(X, y, np.full(p, 1.5), dtype=dtype)
    k = la.hmcKernel(m.lpost, m.glp, eps=0.02, l=9, dmm=np.ones(p))
    a1 = la.mcmc(b, k, **kw)
    a2 = la.mcmc(b, k, chunk=1, **kw)
    first = a1 if first is None else first
    for name, x in (("run vs first model's run", a1), ("chunked vs first model's run", a2)):
        d = np.abs(x - first)
        if d.max() > 0:
            bad.append((rep, name, float(d.max()), np.flatnonzero(d.max(axis=(0, 2)) > 0)[:12].tolist(), np.flatnonzero(d.max(axis=(1, 2)) > 0).tolist()))
print(f"{dtype} p={p} n={n} {os.environ['LOGREG_DEBUG_OPTS']}: {len(bad)} mismatches in {2 * reps} runs", bad[:6], flush=True)
