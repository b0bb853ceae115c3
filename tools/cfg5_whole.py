#!/usr/bin/env python3
"""BASELINE config 5 AS A WHOLE on one GPU: HMC L=50 on synthetic n=4096, p=128, 8192 chains (the fixture's design and step size).
    python3 tools/cfg5_whole.py [chains ...] [--iters K] [--prec auto|full|bf16] [--cfg 5|4] [--dtype float32|float64]      (--cfg 4: BASELINE config 4's design, n = 100 000, p = 8)
Prints one JSON line per chain count: us per log-posterior-gradient evaluation of all chains (HIP events on the launch stream, interior
steps + end points + every launch boundary included), algorithmic TFLOP/s, acceptance, the plan."""
import ctypes as Ct, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, logreg_amd as la
from logreg_amd import _lib
import bench

args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 4
prec = sys.argv[sys.argv.index("--prec") + 1] if "--prec" in sys.argv else "auto"
cfg = int(sys.argv[sys.argv.index("--cfg") + 1]) if "--cfg" in sys.argv else 5
fix = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"fullsize_cfg{cfg}.json")))
n, p = fix["n"], fix["p"]
X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
dtype = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "float32"
m = la.LogReg(X, y, np.array(fix["pscale"]), dtype=dtype)
k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
L = _lib.load()
stream = Ct.c_void_p()
_lib.check(L.lr_stream_create(0, Ct.byref(stream)))
timer = bench.Timer(L, _lib.check, 0, stream)
fg = bench.flops_per_grad_eval(n, p)
for C in [int(a) for a in args] or [8192]:
    rng = np.random.Generator(np.random.Philox(4000 + cfg))
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * rng.standard_normal((C, p))
    cs = la.ChainSet(k, q0, seed=5, stream=stream, precision=prec)
    ms = bench._timed_chainset(la, timer, cs, iters, 1)
    per_eval = ms * 1e-3 / (iters * fix["l"])
    print(json.dumps({"chains": C, "dtype": dtype, "precision": prec, "us_per_evaluation_all_chains": per_eval * 1e6, "algorithmic_TFLOPs": C * fg / per_eval / 1e12,
                      "frac_bf16_peak": C * fg / per_eval / 2.5e15, "accept_rate": float(cs.get_accepts().sum() / (C * (3 * iters + 1))),
                      "chain_iterations_per_s": C * iters / (ms * 1e-3), "plan": cs.plan(), "debug_opts": m.debug_opts()}), flush=True)
