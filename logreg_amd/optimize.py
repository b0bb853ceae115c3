"""Warm start on the device: the MAP the reference finds with SciPy BFGS
(`res = minimize(lambda x: -lpost(x), init, jac=lambda x: -glp(x), method='BFGS')`,
Python/fit-np-hmc.py:49) found here by damped Newton iterations with the CLOSED-FORM Hessian X^T W X + prior
(as the reference's JAX variant computes it, Python/fit-jax-hmc.py:61-79): one `lr_hessian` launch per
iteration returns lpost, glp and the Hessian in float64 from a single pass over the rows; only the p x p
solve runs on the host.  
"""
from __future__ import annotations

import numpy as np


def find_map(model, init=None, max_iter=50, gtol=1e-8, verbose=False):
    """-> (beta_map [p], info).  Damped Newton on the device: every iteration is one `lr_hessian` launch -- lpost,
    glp and the closed-form negative Hessian X^T W X + diag(1/pscale^2) in float64, one pass over the rows -- a
    p x p Cholesky solve on the host, and step halving on the float64 lpost.  Converges quadratically (Pima:
    ~8 iterations to |step| < 1e-10) for float32 and float64 models alike: only the STORED rows differ.

    info: iterations, converged, lpost, grad, hessian, sd = sqrt(diag(H^-1)) (the Laplace scale at the mode: mass
    matrix / over-dispersed starts)."""
    p = model.p
    beta = np.zeros(p) if init is None else np.asarray(init, dtype=np.float64).copy()
    info = {"iterations": 0, "converged": False}
    f0, g0, H = model.hessian(beta)
    for it in range(max_iter):
        info["iterations"] = it
        try:
            Lc = np.linalg.cholesky(H)  # the posterior is strictly log-concave: H is positive definite
            step = np.linalg.solve(Lc.T, np.linalg.solve(Lc, g0))
        except np.linalg.LinAlgError:
            step = np.linalg.lstsq(H, g0, rcond=None)[0]
        if verbose:
            print(f"newton {it}: lpost={f0:.10f} |g|={np.max(np.abs(g0)):.3e} |step|={np.max(np.abs(step)):.3e}")
        if np.max(np.abs(step) / np.maximum(np.abs(beta), 1.0)) < gtol:
            info["converged"] = True
            break
        alpha = 1.0
        while True:  # step halving: lpost is concave, so a short enough Newton step always ascends
            f1, g1, H1 = model.hessian(beta + alpha * step)
            if np.isfinite(f1) and f1 >= f0 - 1e-12 * abs(f0):
                break
            alpha *= 0.5
            if alpha < 1e-8:
                break
        if alpha < 1e-8:
            info["converged"] = True  # no ascent at any scale: float64 resolution of lpost reached
            break
        beta, f0, g0, H = beta + alpha * step, f1, g1, H1
    info.update(lpost=float(f0), grad=g0, hessian=H)
    try:
        info["sd"] = np.sqrt(np.maximum(np.diag(np.linalg.inv(H)), 0.0))
    except np.linalg.LinAlgError:
        info["sd"] = None
    return beta, info


def overdispersed_init(beta_map, sd, chains, scale=2.0, seed=0):
    """Chain starting points MAP + scale * sd * N(0, I) for many-chain runs (the reference starts its
    single chain at the MAP, fit-np-hmc.py:107; over-dispersed starts make split-R-hat meaningful)."""
    rng = np.random.Generator(np.random.Philox(seed))
    beta_map = np.asarray(beta_map, dtype=np.float64)
    return beta_map[None, :] + scale * np.asarray(sd)[None, :] * rng.standard_normal((int(chains), beta_map.shape[0]))
