#!/usr/bin/env python3
"""Default precision policy between exactly-filled chain counts: would a matrix-core head + a register-kernel remainder, co-resident,
beat the single matrix-core launch?  (HMC L=50, n=200, p=8, thin 20.)  Head: mfma S=4 on stream 1 (model with the residency cap);
remainder: reg 64 / 32 on stream 2 (a second model created with residency_cap=0, as the library launches remainders)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
L = _lib.load()
X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
ps = np.array([10.0] + [1.0] * 7)
m1 = la.LogReg(X, y, ps)
os.environ["LOGREG_DEBUG_OPTS"] = "residency_cap=0"
m2 = la.LogReg(X, y, ps)
del os.environ["LOGREG_DEBUG_OPTS"]
k1 = la.hmcKernel(m1.lpost, m1.glp, eps=0.1, l=50, dmm=np.ones(8))
k2 = la.hmcKernel(m2.lpost, m2.glp, eps=0.1, l=50, dmm=np.ones(8))
thin = 20


def stream():
    s = C.c_void_p(); _lib.check(L.lr_stream_create(0, C.byref(s))); return s


def timed(sets, reps=20):
    for _ in range(30):
        for cs in sets: cs.advance(1, thin, keep=False)
    for cs in sets: cs.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            for cs in sets: cs.advance(1, thin, keep=False)
        for cs in sets: cs.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


print("# tools/two_part_auto_probe.py: ms per step of 20 iterations, default precision policy")
for Ctot, head, hg, tg in ((5120, 4096, 4, 64), (6144, 4096, 4, 32), (9216, 8192, 4, 64), (10240, 8192, 4, 32), (20480, 16384, 1, 16)):
    q0 = 0.017 * np.random.default_rng(1).standard_normal((Ctot, 8))
    s1, s2 = stream(), stream()
    single = [la.ChainSet(k1, q0, seed=5, stream=s1)]
    pair = [la.ChainSet(k1, q0[:head], seed=5, mode="mfma", group=hg, stream=s1),
            la.ChainSet(k2, q0[head:], seed=5, mode="reg", group=tg, chain_offset=head, stream=s2, precision="full")]
    t1, t2 = timed(single), timed(pair)
    print(f"{Ctot} chains: single {single[0].plan()} {t1 * 1e3:.3f} ms ({Ctot * thin / t1:.3e} it/s) | {head} on mfma S={hg} + {Ctot - head} on reg {tg}, "
          f"two streams {t2 * 1e3:.3f} ms ({Ctot * thin / t2:.3e} it/s)", flush=True)
