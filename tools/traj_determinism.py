#!/usr/bin/env python3
"""Are the trajectory kernels deterministic, and do they read memory nobody wrote?  Repeats the same seeded run (and its chunked form) and
reports the largest difference; before every repeat the device allocator's free memory is POISONED (blocks of several sizes filled with
0xFF bytes = NaNs and freed again), so that a kernel reading an uninitialised workspace gives itself away.
    python3 tools/traj_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def poison(la, fill):
    blocks = [la.DeviceArray.from_host(0, np.full(nbytes // 4, fill, dtype=np.uint32)) for nbytes in (1 << 16, 1 << 20, 3 << 20, 1 << 24, 1 << 26, 3 << 26)]
    for blk in blocks:
        blk.free()


for dtype in ("float64", "float32"):
    for p, n in ((64, 500), (128, 900)):
        for opt in ("wide_traj=2", "wide_traj=1"):
            os.environ["LOGREG_DEBUG_OPTS"] = opt
            import logreg_amd as la
            X, y, _ = la.synthetic_logreg(n, p, seed=905 + p, beta_sd=0.1)
            m = la.LogReg(X, y, np.full(p, 1.5), dtype=dtype)
            k = la.hmcKernel(m.lpost, m.glp, eps=0.02, l=9, dmm=np.ones(p))
            b = 0.1 * np.random.default_rng(p + 1).standard_normal((600, p))
            kw = dict(thin=1, iters=2, verb=False, seed=12)
            ref = la.mcmc(b, k, **kw)
            worst, nbad, where = 0.0, 0, None
            for rep in range(6):
                poison(la, 0xFFFFFFFF if rep % 3 else 0x7F7F7F7F)
                m = la.LogReg(X, y, np.full(p, 1.5), dtype=dtype)  # (a fresh model: fresh workspaces out of the poisoned pool)
                k = la.hmcKernel(m.lpost, m.glp, eps=0.02, l=9, dmm=np.ones(p))
                out = la.mcmc(b, k, chunk=1 if rep % 2 else None, **kw)
                d = np.abs(out - ref)
                if d.max() > 0:
                    nbad += 1
                    if d.max() > worst:
                        worst = d.max()
                        it, ch, j = np.unravel_index(np.argmax(d), d.shape)
                        where = (int(it), int(ch), int(j), int((d.max(axis=(0, 2)) > 0).sum()))
            print(f"{dtype} p={p} {opt}: {nbad} of 6 repeats differ, max |diff| {worst:.3e}, (iteration, chain, coordinate, chains affected) {where}", flush=True)
