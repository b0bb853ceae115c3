"""Warm start on the device: the MAP the reference finds with SciPy BFGS
(`res = minimize(lambda x: -lpost(x), init, jac=lambda x: -glp(x), method='BFGS')`,
Python/fit-np-hmc.py:49) found here by damped Newton iterations whose every model evaluation is
a batched `lr_eval` launch: one launch gives lpost/glp at the iterate and at p forward-difference
points (the Hessian of this log-concave posterior), a second launch scores the line-search
candidates.  Only the p x p solve runs on the host.  
"""
from __future__ import annotations

import numpy as np


def find_map(model, init=None, max_iter=50, gtol=None, verbose=False):
    """-> (beta_map [p], info).  `model` is a LogReg; float64 models converge to ~1e-8, float32
    models to the fp32 noise floor of glp (|g| ~ 1e-2 on raw-scale Pima)."""
    p = model.p
    beta = np.zeros(p) if init is None else np.asarray(init, dtype=np.float64).copy()
    is32 = np.dtype(model.np_dtype) == np.float32
    if gtol is None:
        gtol = 5e-2 if is32 else 1e-6
    rel_h = 1e-3 if is32 else 1e-6
    alphas = np.array([1.0, 0.5, 0.25, 0.1, 0.03, 0.01, 1e-3])
    info = {"iterations": 0, "converged": False}
    for it in range(max_iter):
        h = rel_h * np.maximum(np.abs(beta), 1e-2)
        pts = np.vstack([beta[None, :], beta[None, :] + np.diag(h)])
        r = model.eval(pts, ("lpost", "glp"))
        f0, g0 = r["lpost"][0], r["glp"][0]
        gn = float(np.max(np.abs(g0) * np.maximum(np.abs(beta), 1.0)))  # scale-aware gradient norm
        if verbose:
            print(f"newton {it}: lpost={f0:.6f} |g|~{gn:.3e}")
        info.update(iterations=it, lpost=float(f0), grad=g0)
        H = (r["glp"][1:] - g0[None, :]) / h[:, None]
        H = 0.5 * (H + H.T)
        # the posterior is log-concave: -H is positive definite up to difference noise
        w, V = np.linalg.eigh(-H)
        w = np.maximum(w, 1e-8 * max(w.max(), 1.0))
        step = V @ ((V.T @ g0) / w)
        cand = beta[None, :] + alphas[:, None] * step[None, :]
        fc = model.eval(cand, ("lpost",))["lpost"]
        best = int(np.nanargmax(fc))
        improved = fc[best] > f0
        if improved:
            beta = cand[best]
        small_step = np.max(np.abs(alphas[best] * step) / np.maximum(np.abs(beta), 1e-3)) < (1e-5 if is32 else 1e-10)
        if gn < gtol or not improved or small_step:
            info["converged"] = True
            break
    r = model.eval(beta, ("lpost", "glp"))
    info.update(lpost=float(r["lpost"]), grad=r["glp"])
    # Laplace scale at the mode: sd_j = sqrt(diag((-H)^-1)) from the last difference Hessian
    try:
        info["sd"] = np.sqrt(np.maximum(np.diag(np.linalg.inv(-H)), 0.0))
    except (np.linalg.LinAlgError, UnboundLocalError):
        info["sd"] = None
    return beta, info


def overdispersed_init(beta_map, sd, chains, scale=2.0, seed=0):
    """Chain starting points MAP + scale * sd * N(0, I) for many-chain runs (the reference starts its
    single chain at the MAP, fit-np-hmc.py:107; over-dispersed starts make split-R-hat meaningful)."""
    rng = np.random.Generator(np.random.Philox(seed))
    beta_map = np.asarray(beta_map, dtype=np.float64)
    return beta_map[None, :] + scale * np.asarray(sd)[None, :] * rng.standard_normal((int(chains), beta_map.shape[0]))
