#!/usr/bin/env python3
"""Where does the time of one interior leapfrog step go?  (development tool)

Needs a library built with time stamps:   LOGREG_HIPCC_FLAGS=-DLR_STAMPS python -m logreg_amd.build --force
Runs BASELINE configs 4 and 5 (1024 chains, default precision policy), reads the per-wave stamps of the first
interior-step launches (lr_tall.h LR_STAMP: 100 MHz wall clock at 0 entry, 1 before / 2 after the prologue barrier,
3 loop start, 4 loop end, 5 after the reduction barrier, 6 exit; shader clock at loop start / end) and prints, per
configuration, medians over the workgroups of a steady-state launch and the gap between consecutive launches."""
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

for cfg in [int(x) for x in sys.argv[1:]] or [4, 5]:
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"fullsize_cfg{cfg}.json")))
    n, p, Ctot = fix["n"], fix["p"], 1024
    X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
    m = la.LogReg(X, y, np.array(fix["pscale"]))
    k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((Ctot, p))
    cs = la.ChainSet(k, q0, seed=3)
    cs.advance(4, 1, keep=False)  # warm: clocks, caches
    cs.sync()
    rd(buf.ctypes.data, None, None)  # discard, restart the slot counter
    cs.advance(1, 1, keep=False)
    cs.sync()
    rd(buf.ctypes.data, None, None)
    used = [s for s in range(ns.value) if buf[s, 0, 0, 0] != 0]
    print(f"config {cfg}: plan {cs.plan()}, {len(used)} stamped launches")
    t = buf[used].astype(np.int64)                      # [launch][wg][wave][k]
    live = (t[:, :, 0, 0] != 0).all(axis=0)             # workgroups that exist
    t = t[:, live]
    waves = (t[5, :, :, 0] != 0).all(axis=0)            # waves that exist
    t = t[:, :, waves]
    tick = 0.01  # us per 100 MHz tick
    names = ["entry->prologue loads issued", "prologue barrier", "operand build", "row loop", "reduction barrier", "stores issued"]
    for s in (5, 20):
        if s >= len(used):
            continue
        x = t[s]
        t0 = x[:, :, 0].min()
        print(f"  launch {s}: first entry 0, last entry {(x[:, :, 0].max() - t0) * tick:.2f} us, first exit {(x[:, :, 6].min() - t0) * tick:.2f}, "
              f"last exit {(x[:, :, 6].max() - t0) * tick:.2f}; gap to next launch's first entry {(t[s + 1][:, :, 0].min() - x[:, :, 6].max()) * tick:.2f} us")
        for k_ in range(6):
            d = (x[:, :, k_ + 1] - x[:, :, k_]) * tick
            print(f"    {names[k_]:32s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
        if x[:, :, 14].max() > 0:  # 16-wave tall kernel: shader cycles of a wave inside the chunk barriers / issuing the chunk loads
            loop = (x[:, :, 9] - x[:, :, 8]).astype(float)
            for nm, k_ in (("  of it in chunk barriers", 14), ("  of it issuing chunk loads", 15)):
                d = x[:, :, k_] / loop * 100
                print(f"    {nm:32s} median {np.median(d):6.1f}  p10 {np.percentile(d, 10):6.1f}  p90 {np.percentile(d, 90):6.1f}  max {d.max():6.1f} % of the loop cycles")
        if (x[:, 0, 10] != 0).all():  # finer prologue stamps (wide row-split kernel, waves that do the fused update)
            w = x[:, :, 10] != 0
            for nm, a_, b_ in (("  entry -> address setup done", 0, 10), ("  loads issued -> all returned", 10, 11), ("  first DMA blocks issued", 11, 12), ("  update arithmetic + LDS + stores", 12, 1)):
                d = ((x[:, :, b_] - x[:, :, a_]) * tick)[w]
                print(f"    {nm:32s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
        if x[:, 0, 13].max() > 0:  # XCC id per workgroup (wide kernel): which XCD do the row slices of one chain tile run on?
            live_idx = np.nonzero(live)[0]
            xcc = {int(w): int(x[i, 0, 13]) for i, w in enumerate(live_idx)}
            print("    XCC id of workgroups 0..15 (linear id x + 64 y):", [xcc.get(w) for w in range(16)], " slices y = 0..3 of tile 5:", [xcc.get(5 + 64 * y_) for y_ in range(4)])
        clk = (x[:, :, 9] - x[:, :, 8]) / np.maximum((x[:, :, 4] - x[:, :, 3]) * tick, 1e-9)
        print(f"    shader clock inside the row loop: median {np.median(clk) / 1e3:.2f} GHz; loop cycles median {np.median(x[:, :, 9] - x[:, :, 8])}")
        wg_span = (x[:, :, 6].max(axis=1) - x[:, :, 0].min(axis=1)) * tick
        print(f"    workgroup lifetime median {np.median(wg_span):.2f} us, max {wg_span.max():.2f}")
    per = np.diff(t[:, :, :, 0].min(axis=(1, 2))) * tick
    print(f"  launch-to-launch period: median {np.median(per[2:]):.2f} us")
