#!/usr/bin/env python3
"""Where does a leapfrog step of the wide TRAJECTORY kernel go?  (development tool: LOGREG_HIPCC_FLAGS=-DLR_STAMPS build)
    LOGREG_DEBUG_OPTS=wide_traj=2 python3 tools/stamps_traj.py [chains]
Shader cycles of a wave per phase, summed over the launch (lr_stamps.h LR_TRAJ_PHASE): 0 operand build, 1 DMA issue,
2 waiting for the block's DMA, [2 -> next 1] = the block's LDS reads + MFMAs + sigmoid (accounted to phase 1 of the next trip), 3 reduction."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
L = _lib.load()
rd = L.lr_debug_read_stamps
rd.restype = C.c_int
rd.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
ns, nw = C.c_int(), C.c_int()
rd(None, C.byref(ns), C.byref(nw))
buf = np.zeros((ns.value, nw.value, 16, 16), dtype=np.uint64)
fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg5.json")))
n, p = fix["n"], fix["p"]
Ctot = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
m = la.LogReg(X, y, np.array(fix["pscale"]))
k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((Ctot, p))
cs = la.ChainSet(k, q0, seed=3)
cs.advance(3, 1, keep=False); cs.sync()
rd(buf.ctypes.data, None, None)
cs.advance(2, 1, keep=False); cs.sync()
rd(buf.ctypes.data, None, None)
wgs = (Ctot + 31) // 32 if "wide_traj=2" in m.debug_opts() else (Ctot + 15) // 16
# (the device indexes the buffer as [launch][workgroups of the grid][16 waves][16]: lr_stamps.h LR_STAMP_AT)
t = buf.reshape(-1)[wgs * 256:2 * wgs * 256].reshape(wgs, 16, 16).astype(np.float64)  # second stamped launch
live = t[:, 0, 7] > 0
t = t[live][:, :8]
steps = fix["l"] - 1
print(f"chains {Ctot}, opts {m.debug_opts()!r}, {int(live.sum())} workgroups, {steps} steps; shader cycles per step and wave (median over waves | p90):")
for name, k_ in (("operand build (beta pieces)", 8), ("block arithmetic (LDS reads, MFMAs, sigmoid)", 12), ("DMA issue (8 x 1 KB per block)", 9), ("waiting for the block's DMA", 10), ("reduction over the waves + update", 11), ("whole step", 7)):
    d = t[:, :, k_] / steps
    print(f"  {name:58s} {np.median(d):9.0f} | {np.percentile(d, 90):9.0f}")
