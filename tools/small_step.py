import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, logreg_amd as la
for n, p, C in ((2000, 20, 1024), (1000, 50, 1024), (4096, 128, 1024)):
    X, y, _ = la.synthetic_logreg(n, p, seed=1, beta_sd=0.1)
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=0.3 / np.sqrt(n), l=20, dmm=np.ones(p))
    cs = la.ChainSet(k, np.zeros((C, p)), seed=5, mode="stepwise")
    cs.advance(1, 1, keep=False); cs.sync()
    t0 = time.perf_counter(); cs.advance(10, 1, keep=False); cs.sync(); dt = time.perf_counter() - t0
    print(n, p, C, cs.plan(), "us per leapfrog step %.2f" % (dt / 200 * 1e6), flush=True)
