#!/usr/bin/env python3
"""Config 5 as a whole (HMC L = 50, n = 4096, p = 128, 8192 chains: k_wide_traj2_bf16) -- time, clock and power of one build, one row of
the watt / clock table of tools/gpu/r6_cfg5_power.sh (is the 13 TB/s of L2 -> LDS streaming where the power goes?).
~5 s of back-to-back launches under a 100 ms rocm-smi sampler; HIP-event time per evaluation (best of the repeats, as tools/cfg5_whole.py).
    python3 tools/cfg5_power_table.py <label>"""
import ctypes as Ct, json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
import bench

label = sys.argv[1] if len(sys.argv) > 1 else "?"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            card = d[sorted(d)[0]]
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), "?")
            pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), "?")
            samples.append((time.perf_counter(), str(sclk), str(pw)))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), "err", repr(e)[:60]))
        time.sleep(0.1)


def num(x):
    m = re.search(r"([0-9.]+)", x)
    return float(m.group(1)) if m else float("nan")


fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fullsize_cfg5.json")))
n, p = fix["n"], fix["p"]
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
m = la.LogReg(X, y, np.array(fix["pscale"]))
k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
L = _lib.load()
stream = Ct.c_void_p()
_lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)
q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.Generator(np.random.Philox(4005)).standard_normal((C, p))
cs = la.ChainSet(k, q0, seed=5, stream=stream)
threading.Thread(target=sampler, daemon=True).start()
cs.advance(1, 1, keep=False); cs.sync()
t0 = time.perf_counter()
best = None
while time.perf_counter() - t0 < 5.0:
    timer.start()
    cs.advance(4, 1, keep=False)
    ms = timer.stop_ms()
    if time.perf_counter() - t0 > 1.5:  # (after the clocks have settled under this load)
        best = ms if best is None or ms < best else best
t1 = time.perf_counter()
stop.set()
mine = [(num(s), num(pw)) for t, s, pw in samples if t0 + 1.5 <= t <= t1]
sclk, power = float(np.nanmedian([a for a, _ in mine])), float(np.nanmedian([b for _, b in mine]))
us = best * 1e3 / (4 * fix["l"])
print(f"{label:46s} {us:6.2f} us per evaluation   sclk {sclk:5.0f} MHz   {power:5.0f} W   {us * power:7.0f} uJ per evaluation   "
      f"{us * sclk / 1e3:6.1f} k shader cycles   plan {cs.plan()} ({len(mine)} samples)", flush=True)
