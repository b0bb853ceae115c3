#!/usr/bin/env python3
"""AUTO against every forced alternative (tests/test_gpu_planner.py runs the same function as a test).

For each shape (n, p, chains, kind, precision): chain-iterations/s of the planner's own choice and of each forced (mode, group)
that accepts the shape, at sustained clocks (a short untimed load first), best of 3.  Prints one line per candidate and the
ratio AUTO / best.   usage: planner_bench.py [--quick] [n,p,C,kind,precision ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import logreg_amd as la  # noqa: E402

# shapes on both sides of the planner's boundaries (lr_plan.h: kMfmaRules rows, the register-family group rule, the
# LDS / stepwise rule); 256 CUs: 4 / 8 / 16 / 40 / 64 / 96 chains per CU = 1024 / 2048 / 4096 / 10 240 / 16 384 / 24 576 chains
SHAPES = [
    (200, 8, 2048, "hmc", "auto"), (200, 8, 4096, "hmc", "auto"), (200, 8, 8192, "hmc", "auto"), (200, 8, 10240, "hmc", "auto"),
    (200, 8, 2560, "hmc", "full"), (200, 8, 4096, "hmc", "full"), (200, 8, 5120, "hmc", "full"), (200, 8, 6144, "hmc", "full"),
    (400, 8, 2048, "hmc", "auto"), (400, 8, 4096, "hmc", "auto"),
    (2000, 8, 1024, "hmc", "auto"), (2000, 8, 2048, "hmc", "auto"), (3000, 8, 2048, "hmc", "auto"), (3000, 8, 4096, "hmc", "auto"),
    (200, 12, 1024, "hmc", "auto"), (500, 16, 1024, "hmc", "auto"), (500, 16, 2048, "hmc", "auto"), (200, 24, 1024, "hmc", "auto"),
    (200, 8, 8192, "mala", "auto"), (200, 8, 3072, "mala", "auto"),
    # round 4: all-fp32 families where the register variant is one chain per wave
    (500, 16, 4096, "mala", "auto"), (500, 16, 8192, "hmc", "full"), (300, 12, 16384, "mala", "auto"), (600, 8, 16384, "mala", "auto"),
]
# forced alternatives tried for every shape (those the library rejects for the shape are skipped)
ALTERNATIVES = [("reg", 16), ("reg", 32), ("reg", 64), ("lds", 1), ("lds", 8), ("lds", 16), ("lds", 64), ("global", 1), ("mfma", 1), ("mfma", 4), ("mfma", 8),
                ("mixed", 16), ("stepwise", 0)]


def rate(cs, C, thin, seconds=0.08):
    """chain-iterations/s of a ChainSet: untimed load until the clocks hold, then best of 3 batches"""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds or n < 2:
        cs.advance(1, thin, keep=False)
        cs.sync()
        n += 1
    per = (time.perf_counter() - t0) / n
    batch = max(2, int(0.01 / per))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(batch):
            cs.advance(1, thin, keep=False)
        cs.sync()
        best = min(best, (time.perf_counter() - t0) / batch)
    return C * thin / best


def candidates(n, p, C, kind, precision, L=20, dtype=None):
    X, y, _ = la.synthetic_logreg(n, p, seed=n + p, beta_sd=0.5 / np.sqrt(p))
    m = la.LogReg(X, y, np.ones(p), dtype=dtype or os.environ.get("PLANNER_BENCH_DTYPE", "float32"))  # (the tool's own switch, not the library's)
    bmap, info = la.find_map(m)
    eps = 0.9 / np.sqrt(np.max(np.linalg.eigvalsh(info["hessian"]))) / p ** 0.25
    if kind == "hmc":
        k, thin = la.hmcKernel(m.lpost, m.glp, eps=eps, l=L, dmm=np.ones(p)), 5
    elif kind == "rwmh":
        k, thin = la.mhKernel(m.lpost, la.rwProposal(0.3 * eps * np.ones(p))), 100
    else:
        k, thin = la.malaKernel(m.lpost, m.glp, dt=eps * eps, pre=np.ones(p)), 100
    q0 = bmap + info["sd"] * np.random.default_rng(1).standard_normal((C, p))
    out = []
    for mode, group in [("auto", 0)] + ALTERNATIVES:
        try:
            cs = la.ChainSet(k, q0, seed=5, mode=mode, group=group, precision=precision)
            plan = cs.plan()
            if mode != "auto" and any(o[1] == plan for o in out):
                continue  # the same variant as one already timed (AUTO's own, usually)
            out.append((mode, plan, rate(cs, C, thin)))
        except la.LogregHipError:
            continue
    return out


def fmt(plan):
    return f"{plan['mode']}{plan['group']}/{plan['rows_per_lane']}"


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    shapes = [tuple(int(v) if v.isdigit() else v for v in a.split(",")) for a in args] or SHAPES
    if "--quick" in sys.argv:
        shapes = shapes[::3]
    print("# tools/planner_bench.py on one MI355X: chain-iterations/s of AUTO and of every forced alternative; ratio = AUTO / best")
    for n, p, C, kind, prec in shapes:
        res = candidates(n, p, C, kind, prec)
        auto, best = res[0], max(res, key=lambda r: r[2])
        line = " | ".join(f"{fmt(pl)} {r:.3e}" for _, pl, r in res[1:])
        print(f"n={n} p={p} C={C} {kind} {prec}: AUTO {fmt(auto[1])} {auto[2]:.3e}  ratio {auto[2] / best[2]:.2f} (best {fmt(best[1])}) || {line}", flush=True)
