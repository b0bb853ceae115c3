#!/usr/bin/env python3
"""Build-differential fuzz (GPU): every random case runs through the production library and through the second build of the same
sources (tests/altlib.py) and must give BIT-IDENTICAL samples, accept counts, threaded log-densities and closure values.
usage: fuzz_builds.py [cases] [seed]   -- prints differences and a summary; exit code 1 on any difference.

The cases cover both dtypes, both precision policies, every kernel family and every engine the planner can be forced onto (register /
LDS / global rows at every lane-group width, matrix-core chain kernels, mixed, distributed-state, stepwise tall and wide, row-split and
trajectory kernels).  No oracle here: tests/fuzz_parity.py compares with the reference arithmetic; this one compares two compilations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import logreg_amd as la
import altlib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
diffs, done, skipped = [], 0, 0
engines = {}
t0 = time.time()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1)


def one(X, y, ps, dtype, kind, kw, q0, ll0, thin, iters, seed, mode, group, prec):
    m = la.LogReg(X, y, ps, dtype=dtype)
    if kind == "hmc":
        kern = la.hmcKernel(m.lpost, m.glp, eps=kw["step"], l=kw["l"], dmm=kw["scale"])
    elif kind == "mala":
        kern = la.malaKernel(m.lpost, m.glp, dt=kw["step"], pre=kw["scale"])
    elif kind == "ul":
        kern = la.ulKernel(m.glp, dt=kw["step"], pre=kw["scale"])
    else:
        kern = la.mhKernel(m.lpost, la.rwProposal(kw["scale"]))
    cs = la.ChainSet(kern, q0, seed=seed, ll=ll0, mode=mode, group=group, precision=prec)
    out = cs.advance(iters, thin).to_host()
    res = {"out": out, "accepts": cs.get_accepts(), "state": cs.get_state(), "ll": cs.get_ll(), "plan": cs.plan()}
    ev = m.eval(q0[:64], mode=mode if mode not in ("stepwise", "mixed") else "auto", group=group if mode not in ("stepwise", "mixed") else 0)
    res["lpost"], res["glp"] = ev["lpost"], ev["glp"]
    m.close()
    return res


for case in range(cases):
    dtype = str(rng.choice(["float32", "float64"]))
    prec = str(rng.choice(["full", "auto"]))
    p = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 32, 33, 40, 64, 100, 128]))
    n = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 199, 200, 201, 208, 209, 255, 256, 257, 400, 513, 1000, 1024, 1025, 1450, 2390, 2401, 2500,
                      5001, 8191, 8193, 9001, 20000]))
    if p > 32:
        n = min(n, 2500)
    C = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 130, 300, 1024]))
    if rng.random() < 0.2:  # many chains: two-part plans, matrix-core kernels at full occupancy, trajectory kernels with one and two tiles
        C = int(rng.choice([1031, 4096, 4097, 5120, 8192, 9000, 17000]))
        n = min(n, 256 if p <= 32 else 1000)
    kind = str(rng.choice(["hmc", "hmc", "mala", "rwmh", "ul"]))
    X, y, _ = la.synthetic_logreg(n, p, seed=5000 + case, beta_sd=0.3 / np.sqrt(p))
    ps = rng.uniform(0.5, 3.0, p)
    modes = [("auto", 0)] * 2
    probe = la.LogReg(X, y, ps, dtype=dtype)
    if p <= 32:
        for md in ("reg", "lds", "global", "mfma", "mixed"):
            for g in (1, 2, 4, 8, 16, 32, 64):
                try:
                    probe.plan(C, g, md); modes.append((md, g))
                except la.LogregHipError:
                    pass
        modes.append(("stepwise", 0))
    probe.close()
    mode, group = modes[int(rng.integers(len(modes)))]
    sc = 1.0 / np.sqrt(max(n, 4))
    q0 = 0.3 * sc * rng.standard_normal((C, p))
    scale = rng.uniform(0.5, 2.0, p)
    if kind == "hmc":
        kw = dict(step=0.3 * sc, l=int(rng.integers(1, 9)), scale=scale)
    elif kind in ("mala", "ul"):
        kw = dict(step=0.05 * sc * sc, scale=scale)
    else:
        kw = dict(scale=0.3 * sc * scale)
    ll0 = None
    if kind in ("mala", "rwmh") and rng.random() < 0.5:
        ll0 = -0.7 * n + rng.standard_normal(C)
    thin, iters = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    tag = f"case {case}: {dtype} {prec} n={n} p={p} C={C} {kind} {mode}/{group} thin={thin} iters={iters}"
    try:
        a = one(X, y, ps, dtype, kind, kw, q0, ll0, thin, iters, case, mode, group, prec)
    except la.LogregHipError as e:
        skipped += 1
        print("SKIP", tag, "->", str(e)[:80]); continue
    altlib.install()
    try:
        b = one(X, y, ps, dtype, kind, kw, q0, ll0, thin, iters, case, mode, group, prec)
    finally:
        altlib.uninstall()
    done += 1
    key = str(a["plan"].get("mode")) + "/" + str(a["plan"].get("group"))
    engines[key] = engines.get(key, 0) + 1
    errs = []
    if a["plan"] != b["plan"]:
        errs.append(f"plans differ: {a['plan']} vs {b['plan']}")
    for k in ("out", "accepts", "state", "ll", "lpost", "glp"):
        if a[k].shape != b[k].shape or not np.array_equal(bits(a[k]), bits(b[k])):
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            errs.append("%s differs in %d of %d entries (max %.3g)" % (k, int((d != 0).sum()) if d.size else -1, d.size, d.max() if d.size else 0))
    if errs:
        diffs.append(tag + f" plan={a['plan']} :: " + "; ".join(errs)); print("DIFF", diffs[-1], flush=True)
print("libraries:", la._lib.load().lr_build_id().decode(), "(production) vs", altlib.load().lr_build_id().decode(), "(second build)")
print("engines:", " ".join(f"{k}x{v}" for k, v in sorted(engines.items())))
print(f"build-differential fuzz: {done} cases run, {skipped} skipped, {len(diffs)} differ, {time.time() - t0:.0f}s")
sys.exit(1 if diffs else 0)
