#!/usr/bin/env python3
"""HMC on a float64 model with every evaluation float64: step-for-step parity of the 16-lanes-per-chain kernel (k_chain_f64x, lr_f64x.h:
state distributed over the group, group across the DPP rows) with the float64 oracle, bit-exact chunk / shard reruns, and the headline
workload's rate on every lane-group width."""
import ctypes as Ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
from oracle.oracle import OracleModel
import bench

for (n, p, C) in ((200, 8, 70), (37, 5, 16), (255, 7, 33), (500, 8, 130), (16, 6, 5)):
    X, y, _ = la.synthetic_logreg(n, p, seed=900 + n, beta_sd=0.3 / np.sqrt(p))
    ps = np.linspace(1.0, 3.0, p)
    orc = OracleModel(X, y, ps)
    m = la.LogReg(X, y, ps, dtype="float64")
    rng = np.random.default_rng(n)
    sc = 1 / np.sqrt(n)
    q0 = 0.3 * sc * rng.standard_normal((C, p))
    scale = rng.uniform(0.5, 2.0, p)
    k = la.hmcKernel(m.lpost, m.glp, eps=0.3 * sc, l=5, dmm=scale)
    ref = orc.run("hmc", q0, step=0.3 * sc, l=5, scale=scale, thin=2, iters=3, seed=11, threads=0)
    for md, g in (("lds", 16), ("lds", 8), ("lds", 64), ("reg", 32), ("reg", 64)):
        try:
            out, info = la.mcmc(q0, k, thin=2, iters=3, verb=False, seed=11, mode=md, group=g, precision="full", return_info=True)
        except la.LogregHipError as e:
            print(f"n={n} p={p} C={C} {md}/{g}: {str(e)[:90]}"); continue
        ok = ref["margin"] > 1e-8
        err = np.max(np.abs(out[:, ok] - ref["out"][:, ok]))
        accd = int((info["accepts"][ok] != ref["accepts"][ok]).sum())
        ch = la.mcmc(q0, k, thin=2, iters=3, verb=False, seed=11, mode=md, group=g, precision="full", chunk=1)
        h = C // 2
        sh = np.concatenate([la.mcmc(q0[:h], k, thin=2, iters=3, verb=False, seed=11, mode=md, group=g, precision="full"),
                             la.mcmc(q0[h:], k, thin=2, iters=3, verb=False, seed=11, mode=md, group=g, precision="full", chain_offset=h)], axis=1)
        print(f"n={n} p={p} C={C} {md}/{g} plan={info['plan']}: max|d state| vs oracle {err:.3g}, accept diffs {accd}, chunk bit-exact {np.array_equal(ch, out)}, "
              f"shards bit-exact {np.array_equal(sh, out)}, accept rate {info['accepts'].mean() / 6:.3f}", flush=True)

L = _lib.load()
stream = Ct.c_void_p(); _lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)
X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7), dtype="float64")
q0 = bench.headline_init(0, 4096)
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=50, dmm=np.ones(8))
for C in (1024, 2048, 4096, 8192, 16384):
    for mode, g in (("auto", 0), ("lds", 16), ("lds", 8), ("reg", 32), ("reg", 64), ("lds", 64)):
        init = np.tile(q0, ((C + 4095) // 4096, 1))[:C]
        try:
            cs = la.ChainSet(k, init, seed=42, stream=stream, precision="full", mode=mode, group=g)
            ms = bench._timed_chainset(la, timer, cs, 10, 20, repeats=3)
        except la.LogregHipError as e:
            print(f"{C} chains {mode}/{g}: {str(e)[:80]}"); continue
        its = C * 200 / (ms * 1e-3)
        print(f"HMC all-float64 {C} chains {mode}/{g}: {its:.4g} it/s, {its * 50 * bench.flops_per_grad_eval(200, 8) / 1e12 / 78.6:.3f} of the fp64 vector peak, "
              f"{ms / 1000 * 1e3:.3f} us per evaluation, plan {cs.plan()}, accept {cs.get_accepts().sum() / (C * 620):.4f}", flush=True)
