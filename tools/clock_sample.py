#!/usr/bin/env python3
"""Which clock and power does the GPU hold under each workload?  (round 4: config 4's interior kernel clocks at 2.03 GHz where the
same loop in isolation holds 2.3 -- power, or the instruction mix?)  Runs each workload for ~2 s of back-to-back launches while a
thread samples `rocm-smi --showclocks --showpower --json` every 100 ms; prints median sclk / power per workload."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import logreg_amd as la

samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), "?")
            pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), "?")
            samples.append((time.perf_counter(), str(sclk), str(pw)))
        except Exception as e:  # the tool's output format is not ours to rely on: record what happened
            samples.append((time.perf_counter(), "err", repr(e)[:60]))
        time.sleep(0.1)


def run(name, cs, iters, thin, seconds=2.0):
    cs.advance(1, thin, keep=False); cs.sync()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        cs.advance(iters, thin, keep=False); cs.sync(); n += 1
    t1 = time.perf_counter()
    mine = [(s, p) for t, s, p in samples if t0 + 0.5 <= t <= t1]
    def num(x):
        import re
        m = re.search(r"([0-9.]+)", x)
        return float(m.group(1)) if m else float("nan")
    sc = [num(s) for s, _ in mine]
    pw = [num(p) for _, p in mine]
    print(f"{name:46s} {n:5d} launches in {t1 - t0:.1f} s   sclk median {np.nanmedian(sc) if sc else float('nan'):7.0f} MHz   power median "
          f"{np.nanmedian(pw) if pw else float('nan'):6.0f} W   ({len(mine)} samples; raw example {mine[-1] if mine else None})", flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
print("idle:", samples[-1] if samples else None, flush=True)
# headline: HMC, n = 200, p = 8, 4096 chains, all fp32
X, y, _ = la.synthetic_logreg(200, 8, seed=20240001)
m = la.LogReg(X, y, np.array([10.0] + [1.0] * 7))
k = la.hmcKernel(m.lpost, m.glp, eps=0.1, l=50, dmm=np.ones(8))
q0 = 0.017 * np.random.default_rng(1).standard_normal((4096, 8))
run("config 2 (reg 16x13, fp32 vector ALU)", la.ChainSet(k, q0, seed=1, precision="full"), 1, 20)
run("config 2, default policy (mfma S=4)", la.ChainSet(k, q0, seed=1, precision="auto"), 1, 20)
km = la.malaKernel(m.lpost, m.glp, dt=1e-3, pre=np.ones(8))
run("MALA 8192 chains (reg 16x13 rs16)", la.ChainSet(km, np.tile(q0, (2, 1)), seed=1), 1, 500)
for cfg in (4, 5):
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"fullsize_cfg{cfg}.json")))
    Xc, yc, _ = la.synthetic_logreg(fix["n"], fix["p"], seed=fix["data_seed"], beta_sd=fix["beta_sd"])
    mc = la.LogReg(Xc, yc, np.array(fix["pscale"]))
    kc = la.hmcKernel(mc.lpost, mc.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    qc = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((1024, fix["p"]))
    run(f"config {cfg} (stepwise, bf16 interior)", la.ChainSet(kc, qc, seed=3), 4, 1)
    run(f"config {cfg}, precision full", la.ChainSet(kc, qc, seed=3, precision="full"), 2, 1)
    if cfg == 5:  # config 5 as a whole: 8192 chains (two-tile trajectory kernel)
        q8 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.default_rng(1).standard_normal((8192, fix["p"]))
        run("config 5 whole, 8192 chains (k_wide_traj2_bf16)", la.ChainSet(kc, q8, seed=3), 4, 1, seconds=4.0)
stop.set()
