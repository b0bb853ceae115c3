"""Posterior diagnostics for MCMC output (host-side, NumPy only).

The reference has no Python implementation of effective sample size: it shells out to R
(`smfsb::mcmcSummary`, reference `Python/analyse.R:17-19`) and otherwise prints
`scipy.stats.describe` (mean, ddof=1 variance; reference `Python/fit-np-hmc.py:113-117`).
This module supplies what the BASELINE metric "ESS/sec" needs:

* `ess_geyer(x)`       -- per-chain ESS, Geyer (1992) initial-positive-sequence estimator on the
                          FFT autocovariance (the estimator BASELINE.md's ESS numbers were computed with).
* `ess_pooled(samples)`-- many-chain ESS: sum of per-chain Geyer ESS (chains are independent).
* `summarise(samples)` -- mean / sd / ESS / MCSE per parameter, pooled over chains.
* `describe(out)`      -- `scipy.stats.describe`-shaped summary (mean, variance with ddof=1).
"""
from __future__ import annotations

import numpy as np


def _autocov_fft(x: np.ndarray) -> np.ndarray:
    """Biased (1/n) autocovariance at lags 0..n-1 of a 1-D series."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    xc = x - x.mean()
    m = 1
    while m < 2 * n:
        m *= 2
    f = np.fft.rfft(xc, m)
    acov = np.fft.irfft(f * np.conj(f), m)[:n]
    return acov / n


def ess_geyer(x: np.ndarray) -> float:
    """Effective sample size of one chain (1-D array) by Geyer's initial positive sequence.

    tau = -1 + 2 * sum_k Gamma_k, Gamma_k = rho_{2k} + rho_{2k+1}, truncated at the first
    non-positive Gamma_k; ESS = n / tau (capped at n * 1 for a constant-free series)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if n < 4:
        return float(n)
    acov = _autocov_fft(x)
    if acov[0] <= 0.0:
        return float(n)
    rho = acov / acov[0]
    npairs = n // 2
    gam = rho[0:2 * npairs:2] + rho[1:2 * npairs:2]
    nonpos = np.nonzero(gam <= 0.0)[0]
    k = int(nonpos[0]) if nonpos.size else npairs
    tau = -1.0 + 2.0 * float(np.sum(gam[:k]))
    if tau <= 0.0:
        return float(n)
    return float(n / tau)


def ess_per_param(mat: np.ndarray) -> np.ndarray:
    """`mat` is `[iters, p]` (one chain, the reference's `mcmc` output shape) -> ESS `[p]`."""
    mat = np.asarray(mat)
    return np.array([ess_geyer(mat[:, j]) for j in range(mat.shape[1])])


def ess_pooled(samples: np.ndarray, max_chains: int | None = None) -> np.ndarray:
    """`samples` is `[iters, C, p]` -> pooled ESS `[p]` = sum over chains of per-chain ESS.

    If `max_chains` is given, only that many (evenly spaced) chains are analysed and the
    result is scaled by C / max_chains (chains are exchangeable)."""
    samples = np.asarray(samples)
    iters, C, p = samples.shape
    idx = np.arange(C)
    scale = 1.0
    if max_chains is not None and C > max_chains:
        idx = np.linspace(0, C - 1, max_chains).astype(np.int64)
        scale = C / float(max_chains)
    tot = np.zeros(p)
    for c in idx:
        tot += ess_per_param(samples[:, c, :])
    return tot * scale


def summarise(samples: np.ndarray, max_chains: int | None = 256) -> dict:
    """Pooled posterior summary of `[iters, C, p]` (or `[iters, p]`) samples.

    Returns dict(mean, sd, ess, mcse) with `[p]` arrays. sd uses ddof=1 over all draws,
    mcse = sd / sqrt(ess)."""
    s = np.asarray(samples, dtype=np.float64)
    if s.ndim == 2:
        s = s[:, None, :]
    flat = s.reshape(-1, s.shape[-1])
    mean = flat.mean(axis=0)
    sd = flat.std(axis=0, ddof=1)
    ess = ess_pooled(s, max_chains=max_chains)
    return {"mean": mean, "sd": sd, "ess": ess, "mcse": sd / np.sqrt(ess)}


def split_rhat(samples: np.ndarray) -> np.ndarray:
    """Split-R-hat (Gelman et al., BDA3) per parameter for `[iters, C, p]` samples: every chain is
    split in two halves; values near 1 indicate the chains agree."""
    s = np.asarray(samples, dtype=np.float64)
    iters, C, p = s.shape
    h = iters // 2
    halves = np.concatenate([s[:h], s[h:2 * h]], axis=1)  # [h, 2C, p]
    m = halves.mean(axis=0)
    W = halves.var(axis=0, ddof=1).mean(axis=0)
    B = h * m.var(axis=0, ddof=1)
    var_plus = (h - 1) / h * W + B / h
    return np.sqrt(var_plus / W)


def describe(out: np.ndarray) -> dict:
    """Same numbers `scipy.stats.describe(out)` reports in the reference
    (`Python/fit-np-hmc.py:113-117`): nobs, minmax, mean, variance (ddof=1)."""
    out = np.asarray(out, dtype=np.float64)
    flat = out.reshape(-1, out.shape[-1])
    return {
        "nobs": flat.shape[0],
        "minmax": (flat.min(axis=0), flat.max(axis=0)),
        "mean": flat.mean(axis=0),
        "variance": flat.var(axis=0, ddof=1),
    }
