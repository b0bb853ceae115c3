import sys, numpy as np
sys.path.insert(0, '.')
import logreg_amd as la
from oracle.oracle import OracleModel
for n, p in ((8193, 1), (8193, 8), (8193, 3), (2500, 1)):
    X, y, _ = la.synthetic_logreg(n, p, seed=1038, beta_sd=0.3 / np.sqrt(p))
    ps = np.full(p, 1.7)
    orc = OracleModel(X, y, ps)
    m = la.LogReg(X, y, ps)
    q0 = 0.3 / np.sqrt(n) * np.random.default_rng(3).standard_normal((130, p))
    ref = orc.lpost(q0); refg = orc.glp(q0)
    for mode, g in (("lds", 1), ("lds", 8), ("lds", 64), ("global", 1), ("global", 64), ("stepwise", 0)):
        try:
            r = m.eval(q0, mode=mode if mode != "stepwise" else "auto", group=g)
        except la.LogregHipError as e:
            print(n, p, mode, g, "n/a"); continue
        print(f"n={n} p={p} {mode}/{g}: lpost |err| max {np.max(np.abs(r['lpost'] - ref)):.3g} mean {np.mean(r['lpost'] - ref):+.3g} (|lpost| ~ {np.abs(ref).mean():.0f}); glp err max {np.max(np.abs(r['glp'] - refg)):.3g} (|glp| max {np.abs(refg).max():.0f})")
