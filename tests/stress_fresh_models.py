"""First run on a FRESH model against every later one, for each kernel family with hand-managed synchronisation (LDS-DMA rings,
counted waits, fused prologues): the probe that exposed the two-tile trajectory kernel's undone wait in round 5 (one run in ~200 on a
fresh model: cold TLBs, slow loads).  Per scenario: `reps` fresh models, a seeded HMC run and its chunked repeat on each; every run must
be bit-identical with the very first.  tests/test_gpu_builds.py runs a slice; `python3 tools/fresh_model_stress.py [reps]` a campaign."""
import os

import numpy as np

# (name, LOGREG_DEBUG_OPTS -- read once per model at creation, dtype, n, p, chains, L)
SCEN = [
    ("wide row-split (k_wide_partial_bf16r), fused prologue", "wide_traj=0", "float32", 2200, 128, 600, 6),
    ("wide row-split, float64 state", "wide_traj=0", "float64", 2200, 128, 600, 6),
    ("wide chain-split (k_wide_partial_bf16i)", "wide_traj=0", "float32", 700, 64, 4200, 5),
    ("wide trajectory kernel, one tile per workgroup", "wide_traj=1", "float32", 500, 64, 600, 9),
    ("wide trajectory kernel, one tile per workgroup, float64 state", "", "float64", 2200, 128, 600, 6),
    ("wide two-tile trajectory kernel, p = 128", "wide_traj=2", "float32", 900, 128, 600, 9),
    ("tall 16-wave interior kernel (k_tall_partial_mx16)", "", "float32", 30000, 8, 1024, 6),
    ("tall 4-wave interior kernel (k_tall_partial_mx)", "tall_mx16=0", "float32", 30000, 8, 1024, 6),
    ("tall, float64 state", "", "float64", 30000, 8, 1024, 6),
    ("fused matrix-core chain kernel (k_chain_mfma), 4096 chains", "", "float32", 200, 8, 4096, 10),
    ("fused matrix-core chain kernel, rows in LDS", "", "float32", 1500, 8, 4096, 6),
    ("fused register kernel, exact (k_chain rs16)", "", "float32", 200, 8, 4096, 10),
]


def run_scenario(la, scen, reps):
    """-> (runs that differ from the first, runs, plan) for one row of SCEN"""
    name, opt, dtype, n, p, C, L = scen
    saved = os.environ.get("LOGREG_DEBUG_OPTS")
    if opt:
        os.environ["LOGREG_DEBUG_OPTS"] = opt
    else:
        os.environ.pop("LOGREG_DEBUG_OPTS", None)
    try:
        X, y, _ = la.synthetic_logreg(n, p, seed=77 + p + n, beta_sd=0.3 / np.sqrt(p))
        b = (0.3 / np.sqrt(n)) * np.random.default_rng(n).standard_normal((C, p))
        prec = "full" if "exact" in name else "auto"
        kw = dict(thin=1, iters=2, verb=False, seed=3, precision=prec, return_info=True)
        first, bad, plan = None, 0, None
        for rep in range(reps):
            m = la.LogReg(X, y, np.full(p, 2.0), dtype=dtype)
            k = la.hmcKernel(m.lpost, m.glp, eps=0.5 / np.sqrt(n), l=L, dmm=np.ones(p))
            for chunk in (None, 1):
                out, info = la.mcmc(b, k, chunk=chunk, **kw)
                plan = info["plan"]
                first = out if first is None else first
                bad += not np.array_equal(out, first)
            m.close()
        return bad, 2 * reps, plan
    finally:
        if saved is None:
            os.environ.pop("LOGREG_DEBUG_OPTS", None)
        else:
            os.environ["LOGREG_DEBUG_OPTS"] = saved
