import ctypes as Ct, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, logreg_amd as la
from logreg_amd import _lib
import bench
L = _lib.load()
stream = Ct.c_void_p(); _lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)
X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7), dtype="float64")
q0 = bench.headline_init(0, 4096)
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=50, dmm=np.ones(8))
for C in (32768, 65536, 131072, 262144):
    for mode, g in (("auto", 0), ("lds", 8), ("global", 1), ("lds", 1)):
        init = np.tile(q0, (C // 4096, 1))
        try:
            cs = la.ChainSet(k, init, seed=42, stream=stream, precision="full", mode=mode, group=g)
            ms = bench._timed_chainset(la, timer, cs, 2, 20, repeats=2)
        except la.LogregHipError as e:
            print(C, mode, g, str(e)[:80]); continue
        its = C * 40 / (ms * 1e-3)
        print(f"HMC all-float64 {C} chains {mode}/{g}: {its:.4g} it/s, {its * 50 * bench.flops_per_grad_eval(200, 8) / 1e12 / 78.6:.3f} of the fp64 vector peak, plan {cs.plan()}", flush=True)
