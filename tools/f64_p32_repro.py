import sys, numpy as np
sys.path.insert(0, '.')
import logreg_amd as la
from oracle.oracle import OracleModel
for (n, p, C) in ((1, 17, 15), (16, 24, 64), (255, 32, 16), (255, 32, 64), (100, 20, 16), (1000, 30, 130)):
    X, y, _ = la.synthetic_logreg(n, p, seed=1071, beta_sd=0.3 / np.sqrt(p))
    rng = np.random.default_rng(5)
    ps = rng.uniform(0.5, 3.0, p)
    orc = OracleModel(X, y, ps)
    sc = 1.0 / np.sqrt(max(n, 4))
    q0 = 0.3 * sc * rng.standard_normal((C, p))
    scale = rng.uniform(0.5, 2.0, p)
    dt = 0.05 * sc * sc
    ll0 = orc.lpost(q0)
    for kind in ("mala", "rwmh", "hmc", "ul"):
        if kind == "mala": kw = dict(step=dt, scale=scale)
        elif kind == "ul": kw = dict(step=dt, scale=scale)
        elif kind == "hmc": kw = dict(step=0.3 * sc, l=3, scale=scale)
        else: kw = dict(scale=0.3 * sc * scale)
        ref = orc.run(kind, q0, thin=2, iters=2, seed=71, ll_state=ll0 if kind in ("mala", "rwmh") else None, threads=0, **kw)
        for dtype in ("float64", "float32"):
            m = la.LogReg(X, y, ps, dtype=dtype)
            kern = {"mala": lambda: la.malaKernel(m.lpost, m.glp, dt=dt, pre=scale), "ul": lambda: la.ulKernel(m.glp, dt=dt, pre=scale),
                    "hmc": lambda: la.hmcKernel(m.lpost, m.glp, eps=0.3 * sc, l=3, dmm=scale), "rwmh": lambda: la.mhKernel(m.lpost, la.rwProposal(0.3 * sc * scale))}[kind]()
            for mode, g in (("auto", 0), ("lds", 64), ("lds", 8), ("lds", 1), ("global", 64), ("global", 1)):
                try:
                    out, info = la.mcmc(q0, kern, thin=2, iters=2, verb=False, seed=71, ll=ll0 if kind in ("mala", "rwmh") else None, mode=mode, group=g, return_info=True, precision="full")
                except la.LogregHipError as e:
                    continue
                err = np.max(np.abs(out - ref["out"]))
                flag = "  <-- BAD" if err > 1e-2 * sc else ""
                if flag or mode == "auto":
                    print(f"n={n} p={p} {kind} {dtype} {mode}/{g} plan={info['plan']} err={err:.3g} acc_diff={(info['accepts'] != ref['accepts']).sum()}{flag}", flush=True)
