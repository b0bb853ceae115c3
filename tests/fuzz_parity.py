#!/usr/bin/env python3
"""Randomised parity fuzz (GPU): random n, p, chain counts, kernels and engines against the CPU oracle.
usage: fuzz_parity.py [cases] [seed] [full|auto] [float32|float64]   -- prints failures and a summary; exit code 1 on any failure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la
from oracle.oracle import OracleModel

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
# third argument "auto": HMC cases run with the default interior-gradient policy (bf16 matrix-pipe interior steps
# where a kernel exists) and are compared with the oracle at the looser tolerances such trajectories allow
PREC = sys.argv[3] if len(sys.argv) > 3 else "full"
# fourth argument: the model's dtype ("float64": the all-float64 kernels, and with "auto" the float64 models' default policy --
# float32 / bf16 force inside HMC trajectories under a float64 state)
DTYPE = sys.argv[4] if len(sys.argv) > 4 else "float32"
fails, done, skipped = [], 0, 0
t0 = time.time()
for case in range(cases):
    # (FUZZ_P="9,12,16,17,20,24,31,32": a campaign on chosen widths)
    p = int(rng.choice([int(v) for v in os.environ["FUZZ_P"].split(",")] if os.environ.get("FUZZ_P") else
                       [1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 32, 33, 40, 64, 100, 128]))
    n = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 199, 200, 201, 208, 209, 255, 256, 257, 400, 513, 1000, 1024, 1025, 1450, 2390, 2401, 2500,
                      5001, 8191, 8193, 9001]))
    if p > 32:
        n = min(n, 1000)
    C = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 130, 300]))
    big = rng.random() < float(os.environ.get("FUZZ_BIG", "0.12"))  # (FUZZ_BIG=0.6: a campaign on many-chain cases)  many chains (two-part plans, matrix-core kernels at full occupancy; wide models: the trajectory kernels with one
    if big:                    # and two chain tiles per workgroup): few rows, the oracle replays the first 64 chains and 64 behind the largest
        C = int(rng.choice([1031, 4097, 5120, 9000, 10240, 17000]))  # exactly-filled count
        n = min(n, 256)
    kind = str(rng.choice(["hmc", "mala", "rwmh", "ul"]))
    X, y, _ = la.synthetic_logreg(n, p, seed=1000 + case, beta_sd=0.3 / np.sqrt(p))
    ps = rng.uniform(0.5, 3.0, p)
    orc = OracleModel(X, y, ps)
    m = la.LogReg(X, y, ps, dtype=DTYPE)
    modes = [("auto", 0)]
    if DTYPE == "float64" and kind == "hmc" and p <= 16:
        modes += [("mixed", g) for g in (16, 32, 64)]  # (rejected where the rows do not fit: skipped)
    if DTYPE == "float64" and 16 < p <= 32:  # (float64 at 17 <= p <= 32: the distributed-state kernel, 16 or 64 lanes per chain)
        modes += [("lds", 16), ("lds", 64), ("global", 16), ("global", 64), ("stepwise", 0)]
    if p <= 32 and not (DTYPE == "float64" and p > 16):
        modes += [("lds", 8), ("lds", 64), ("global", 64), ("global", 1), ("stepwise", 0)]
        for g in (16, 32, 64):
            try:
                m.plan(C, g, "reg"); modes.append(("reg", g))
            except la.LogregHipError:
                pass
        for g in (1, 4):  # matrix-core chain kernel: operands in registers, LDS or device memory as n grows (twice the weight)
            try:
                m.plan(C, g, "mfma"); modes += [("mfma", g)] * 2
            except la.LogregHipError:
                pass
    mode, group = modes[int(rng.integers(len(modes)))]
    sc = 1.0 / np.sqrt(max(n, 4))
    q0 = 0.3 * sc * rng.standard_normal((C, p))
    scale = rng.uniform(0.5, 2.0, p)
    if kind == "hmc":
        L = int(rng.integers(1, 8)); eps = 0.3 * sc
        kern = la.hmcKernel(m.lpost, m.glp, eps=eps, l=L, dmm=scale); kw = dict(step=eps, l=L, scale=scale)
    elif kind == "mala":
        dt = 0.05 * sc * sc
        kern = la.malaKernel(m.lpost, m.glp, dt=dt, pre=scale); kw = dict(step=dt, scale=scale)
    elif kind == "ul":
        dt = 0.05 * sc * sc
        kern = la.ulKernel(m.glp, dt=dt, pre=scale); kw = dict(step=dt, scale=scale)
    else:
        sd = 0.3 * sc * scale
        kern = la.mhKernel(m.lpost, la.rwProposal(sd)); kw = dict(scale=sd)
    ll0 = orc.lpost(q0) if kind in ("mala", "rwmh") else None
    thin, iters = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    tag = f"case {case}: n={n} p={p} C={C} {kind} {mode}/{group} thin={thin} iters={iters}"
    sel = np.arange(C)
    if big:
        mode, group = "auto", 0
        t0c = (C // 4096) * 4096 if C > 4096 else 0
        sel = np.unique(np.concatenate([np.arange(64), np.arange(t0c, min(C, t0c + 64))]))
    try:
        if big:  # two oracle runs with the global chain ids of the two blocks
            blocks = [sel[sel < 64], sel[sel >= 64]]
            parts = [orc.run(kind, q0[b], thin=thin, iters=iters, seed=case, ll_state=None if ll0 is None else ll0[b], threads=0, chain_offset=int(b[0]), **kw)
                     for b in blocks if len(b)]
            ref = {k2: np.concatenate([pp[k2] for pp in parts], axis=1 if k2 == "out" else 0) for k2 in ("out", "accepts", "margin")}
        else:
            ref = orc.run(kind, q0, thin=thin, iters=iters, seed=case, ll_state=ll0, threads=0, **kw)
        out_all, info = la.mcmc(q0, kern, thin=thin, iters=iters, verb=False, seed=case, ll=ll0, mode=mode, group=group,
                                return_info=True, precision=PREC)
        out = out_all[:, sel]
        info = dict(info, accepts=info["accepts"][sel])
        r = m.eval(q0[sel], mode=mode if mode != "stepwise" else "auto", group=group if mode != "stepwise" else 0)
    except la.LogregHipError as e:
        skipped += 1
        print("SKIP", tag, "->", str(e)[:80]); continue
    done += 1
    # decisions are compared where the oracle's |a - log u| clears the fp32 resolution of the log-density
    # (|ll| ~ 0.7 n: ulp-level sums of n terms; 2e-3 up to n = 1000, growing with n)
    ok = ref["margin"] > 2e-3 * max(1.0, n / 1000.0)
    if p == 1 and DTYPE == "float32":
        # intercept-only design: every row is +-1, so the fp32 rounding errors of the n value terms and of their running sum are all
        # the same error, not a random walk (n = 8193 on 1 - 8 lanes per chain: |lpost error| up to 0.033 where p = 3 has 0.0015;
        # tools/f32_value_sum_probe.py) -- a forced narrow lane group there resolves decisions to ~0.1
        ok = ref["margin"] > 1.5e-2 * max(1.0, n / 1000.0)
    loose = PREC != "full" and kind == "hmc"
    if loose:
        ok = ref["margin"] > 0.25
    lp_ref = orc.lpost(q0[sel])
    errs = []
    if not np.allclose(r["lpost"], lp_ref, rtol=3e-5, atol=3e-5 * n ** 0.5):
        errs.append("lpost %.3g" % np.max(np.abs(r["lpost"] - lp_ref)))
    gtol = 2e-4 * np.sqrt(n) * max(1.0, np.abs(X).max())
    if np.max(np.abs(r["glp"] - orc.glp(q0[sel]))) > gtol:
        errs.append("glp %.3g" % np.max(np.abs(r["glp"] - orc.glp(q0[sel]))))
    if ok.any():
        if not np.array_equal(info["accepts"][ok], ref["accepts"][ok].astype(np.uint32)):
            errs.append("accepts differ in %d chains" % int((info["accepts"][ok] != ref["accepts"][ok]).sum()))
        d = np.max(np.abs(out[:, ok] - ref["out"][:, ok]))
        if not d < (2e-3 if not loose else 6e-2) * sc * 3 + 1e-5:
            errs.append("states %.3g" % d)
    if not np.isfinite(out).all():
        errs.append("non-finite output")
    if rng.random() < 0.4:  # chunk and shard invariance: bit-exact (global chain id and iteration in the Philox counter)
        again = la.mcmc(q0, kern, thin=thin, iters=iters, verb=False, seed=case, ll=ll0, mode=mode, group=group, chunk=1,
                        precision=PREC)
        if not np.array_equal(again, out_all):
            errs.append("chunk=1 differs")
        pl = info["plan"]
        if C > 1 and pl["mode"] != "stepwise" and not big:  # (stepwise slicing depends on the chain count by design)
            h = C // 2
            a = la.mcmc(q0[:h], kern, thin=thin, iters=iters, verb=False, seed=case, ll=None if ll0 is None else ll0[:h],
                        mode=pl["mode"], group=pl["group"], precision=PREC)
            b = la.mcmc(q0[h:], kern, thin=thin, iters=iters, verb=False, seed=case, ll=None if ll0 is None else ll0[h:],
                        mode=pl["mode"], group=pl["group"], chain_offset=h, precision=PREC)
            if not np.array_equal(np.concatenate([a, b], axis=1), out):
                errs.append("shards differ")
        if big:  # a shard of the planned run, straddling the split of a two-part plan if there is one
            lo = max(0, t0c - 50)
            sh = la.mcmc(q0[lo:lo + 120], kern, thin=thin, iters=iters, verb=False, seed=case, ll=None if ll0 is None else ll0[lo:lo + 120],
                         chain_offset=lo, plan_chains=C, plan_first=0, precision=PREC)
            if not np.array_equal(sh, out_all[:, lo:lo + 120]):
                errs.append("planned shard differs")
    if rng.random() < 0.3:  # on-device streaming statistics of the same engine against NumPy on its own kept samples
        try:
            cs = la.ChainSet(kern, q0, seed=case, ll=ll0, mode=mode, group=group, precision=PREC)
            cs.enable_stats(2, 3)
            smp = np.concatenate([cs.advance(k_, thin).to_host() for k_ in (1, 3, 2)]).astype(np.float64)
            got = cs.stats_summary()
            flat = smp.reshape(-1, p)
            tol = 1e-9 if DTYPE == "float64" else 1e-6
            if not np.allclose(got["mean"], flat.mean(0), rtol=tol, atol=tol * (np.abs(flat).max() + 1e-30)):
                errs.append("stats mean %.3g" % np.max(np.abs(got["mean"] - flat.mean(0))))
            if C * 6 > 1 and not np.allclose(got["sd"], flat.std(0, ddof=1), rtol=1e-5, atol=1e-7 * (np.abs(flat).max() + 1e-30)):
                errs.append("stats sd %.3g" % np.max(np.abs(got["sd"] - flat.std(0, ddof=1))))
        except la.LogregHipError as e:
            errs.append("stats run: " + str(e)[:80])
    if rng.random() < 0.15:  # checkpoint in the middle of a run, resume in another object: bit-equal to the uninterrupted run
        try:
            one = la.ChainSet(kern, q0, seed=case, ll=ll0, mode=mode, group=group, precision=PREC)
            whole = one.advance(4, thin).to_host()
            two = la.ChainSet(kern, q0, seed=case, ll=ll0, mode=mode, group=group, precision=PREC)
            first = two.advance(1, thin).to_host()
            three = la.ChainSet.resume(kern, two.checkpoint(), group=group, mode=mode, precision=PREC)
            rest = three.advance(3, thin).to_host()
            if not np.array_equal(np.concatenate([first, rest]), whole):
                errs.append("checkpoint / resume differs")
            if not np.array_equal(one.get_accepts(), three.get_accepts()):
                errs.append("accept counts after resume differ")
        except la.LogregHipError as e:
            errs.append("checkpoint run: " + str(e)[:80])
    if errs:
        fails.append(tag + " :: " + "; ".join(errs)); print("FAIL", fails[-1], flush=True)
        # the same chains on every engine that takes the shape, and on the other dtype: which variants disagree with the oracle?
        for dt2 in (() if big else ("float64", "float32")):
            m2 = la.LogReg(X, y, ps, dtype=dt2)
            k2 = {"hmc": lambda: la.hmcKernel(m2.lpost, m2.glp, eps=kw.get("step"), l=kw.get("l", 1), dmm=scale),
                  "mala": lambda: la.malaKernel(m2.lpost, m2.glp, dt=kw.get("step"), pre=scale),
                  "ul": lambda: la.ulKernel(m2.glp, dt=kw.get("step"), pre=scale),
                  "rwmh": lambda: la.mhKernel(m2.lpost, la.rwProposal(kw["scale"]))}[kind]()
            for md, g in [("auto", 0), ("reg", 16), ("reg", 32), ("reg", 64), ("lds", 1), ("lds", 8), ("lds", 16), ("lds", 64), ("global", 64), ("global", 1), ("stepwise", 0)]:
                try:
                    o2, i2 = la.mcmc(q0, k2, thin=thin, iters=iters, verb=False, seed=case, ll=ll0, mode=md, group=g, return_info=True, precision="full")
                except la.LogregHipError:
                    continue
                bad = np.where(np.max(np.abs(o2 - ref["out"]), axis=(0, 2)) > 1e-3 * sc)[0]
                print(f"   {dt2} {md}/{g} {i2['plan']}: max err {np.max(np.abs(o2 - ref['out'])):.3g}, accept diffs {(i2['accepts'] != ref['accepts']).sum()}, "
                      f"chains off {bad[:8].tolist()} margins {np.round(ref['margin'][bad[:8]], 4).tolist()}", flush=True)
print(f"fuzz: {done} cases run, {skipped} skipped, {len(fails)} failed, {time.time() - t0:.0f}s")
sys.exit(1 if fails else 0)
