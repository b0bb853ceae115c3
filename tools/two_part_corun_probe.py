#!/usr/bin/env python3
"""Would the two launches of a two-part plan (lr_plan.h: exactly-filled head on 16 lanes per chain + remainder on wider groups) gain
from running CO-RESIDENT instead of back to back?  Experiment outside the library: the head and the tail as two ChainSets with
forced variants on two streams (residency cap off: LOGREG_DEBUG_OPTS=residency_cap=0 set here before the model is created, so
that a head and a tail workgroup can share a CU), timed against the same two launches on one stream and against the library's
own two-part run.  Headline workload (HMC L=50, n=200, p=8, thin 20), all fp32."""
import ctypes as C, os, sys, time
os.environ["LOGREG_DEBUG_OPTS"] = "residency_cap=0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
L = _lib.load()
n, p, thin = 200, 8, 20
X, y, _ = la.synthetic_logreg(n, p, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7))
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=50, dmm=np.ones(p))


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


print("# tools/two_part_corun_probe.py: ms per step of 20 iterations, residency cap off")
for Ctot, head, tg in ((5120, 4096, 64), (6144, 4096, 32), (10240, 8192, 32)):
    q0 = 0.017 * np.random.default_rng(1).standard_normal((Ctot, p))
    s1, s2 = stream(), stream()
    one = [la.ChainSet(k, q0[:head], seed=5, mode="reg", group=16, stream=s1, precision="full"),
           la.ChainSet(k, q0[head:], seed=5, mode="reg", group=tg, chain_offset=head, stream=s1, precision="full")]
    two = [la.ChainSet(k, q0[:head], seed=5, mode="reg", group=16, stream=s1, precision="full"),
           la.ChainSet(k, q0[head:], seed=5, mode="reg", group=tg, chain_offset=head, stream=s2, precision="full")]
    lib = [la.ChainSet(k, q0, seed=5, stream=s1, precision="full")]
    t1, t2, t3 = timed(one), timed(two), timed(lib)
    print(f"{Ctot} chains = {head} on reg 16 + {Ctot - head} on reg {tg}: one stream {t1 * 1e3:.3f}  two streams {t2 * 1e3:.3f}  library {t3 * 1e3:.3f} ms "
          f"({Ctot * thin / t1:.3e} | {Ctot * thin / t2:.3e} | {Ctot * thin / t3:.3e} it/s)  plan {lib[0].plan()}", flush=True)
