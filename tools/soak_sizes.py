import numpy as np, time, sys
sys.path.insert(0, "/root/repo")
import logreg_amd as la
X, y = la.load_pima()
ps = np.array([10.0, 1, 1, 1, 1, 1, 1, 1]); PRE = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
m = la.LogReg(X, y, ps)
MAP = np.array([-9.19131622, 0.09705401, 0.03112265, -0.00564495, -0.00062272, 0.0814371, 1.26032561, 0.03939102])
k = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=50, dmm=1 / PRE)
for C in (1, 3, 65, 4097, 1 << 20):
    t0 = time.time()
    out, info = la.mcmc(np.tile(MAP, (C, 1)), k, thin=2, iters=3, verb=False, seed=1, return_info=True)
    print(C, info["plan"], out.shape, np.isfinite(out).all(), float(info["accepts"].mean()) / 6, "%.2fs" % (time.time() - t0), flush=True)
    # chain c of a big run == chain c run alone (global chain id in the Philox counter)
    if C > 1:
        solo, _ = la.mcmc(np.tile(MAP, (1, 1)), k, thin=2, iters=3, verb=False, seed=1, return_info=True, chain_offset=C - 1, group=info["plan"]["group"], mode=info["plan"]["mode"])
        print("  last chain solo == in-batch:", np.array_equal(solo[:, 0], out[:, C - 1]))
