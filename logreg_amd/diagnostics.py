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
* `summary_from_sums(sums, ...)` -- mean / sd / split-R-hat / batch-means ESS from the chain-pooled sums the DEVICE
                          accumulates while sampling (`lr_stats_reduce`, include/logreg_hip.h): many-chain runs
                          never materialise `[iters, C, p]`; `batch_sums(samples, ...)` is the NumPy statement of
                          the same sums (tests, and host-side samples).
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


STATS_ROWS = 7  # LR_STATS_ROWS, include/logreg_hip.h


def choose_batches(iters: int, max_batches: int = 32) -> tuple[int, int]:
    """(batch length B, slots) for `iters` kept samples: the largest EVEN batch count <= max_batches that divides
    `iters` (so the two halves of split-R-hat are whole batches and nothing is left over); if none does, B =
    iters // max_batches with the remainder in one extra partial slot (it enters mean/sd only)."""
    iters = int(iters)
    for nb in range(max_batches - max_batches % 2, 1, -2):
        if iters % nb == 0:
            return iters // nb, nb
    B = max(1, iters // max_batches)
    return B, -(-iters // B)


def batch_sums(samples: np.ndarray, batch: int, pivot) -> np.ndarray:
    """NumPy statement of `lr_stats_reduce` (include/logreg_hip.h): `[iters, C, p]` samples -> sums `[7, p]`."""
    s = np.asarray(samples, dtype=np.float64)
    n, C, p = s.shape
    piv = np.asarray(pivot, dtype=np.float64)
    nb = n // batch
    out = np.zeros((STATS_ROWS, p))
    mean_c = s.mean(axis=0)
    out[0] = (n * (mean_c - piv)).sum(axis=0)
    out[1] = (n * (mean_c - piv) ** 2).sum(axis=0)
    out[2] = ((s - mean_c) ** 2).sum(axis=(0, 1))
    full = s[: nb * batch]
    if nb >= 2 and nb % 2 == 0:
        h = nb * batch // 2
        halves = np.concatenate([full[:h], full[h:]], axis=1)  # [h, 2C, p]
        mh = halves.mean(axis=0)
        out[3] = (mh - piv).sum(axis=0)
        out[4] = ((mh - piv) ** 2).sum(axis=0)
        out[5] = halves.var(axis=0, ddof=1).sum(axis=0) if h > 1 else 0.0
    if nb >= 2:
        bmean = full.reshape(nb, batch, C, p).mean(axis=1)
        out[6] = ((bmean - full.mean(axis=0)) ** 2).sum(axis=(0, 1))
    return out


def summary_from_sums(sums, n_chains: int, kept: int, batch: int, pivot) -> dict:
    """Posterior summary from chain-pooled sums (rows as in include/logreg_hip.h; sums of several chain shards
    simply add).  Returns mean, sd (ddof = 1 over all draws, as `scipy.stats.describe` in fit-np-hmc.py:113-117),
    rhat (split-R-hat, BDA3: the two halves of every chain), ess (batch means, pooled over chains: what
    `smfsb::mcmcSummary` reports per chain in Python/analyse.R:17-19), mcse = sd / sqrt(ess); mcse_chains / ess_chains
    (from the spread between the chain means: the estimate to trust when chains are many and short)."""
    S = np.asarray(sums, dtype=np.float64)
    piv = np.asarray(pivot, dtype=np.float64)
    C, n = int(n_chains), int(kept)
    N = C * n
    nb = n // batch
    mean_d = S[0] / N
    var = (S[2] + S[1] - N * mean_d * mean_d) / max(N - 1, 1)
    res = {"n": N, "chains": C, "mean": piv + mean_d, "sd": np.sqrt(np.maximum(var, 0.0))}
    nan = np.full(S.shape[1], np.nan)
    # Monte-Carlo error of the pooled mean from the spread BETWEEN the chains' means: with many independent chains
    # this needs no autocorrelation estimate at all (valid whatever the chains' mixing time, given stationary starts)
    if C >= 2:
        var_means = np.maximum(S[1] / n - C * mean_d * mean_d, 0.0) / (C - 1)
        res["mcse_chains"] = np.sqrt(var_means / C)
        with np.errstate(divide="ignore", invalid="ignore"):
            res["ess_chains"] = var / (var_means / C)
    else:
        res["mcse_chains"], res["ess_chains"] = nan, nan
    if nb >= 2 and nb % 2 == 0:
        h = nb * batch // 2
        W = S[5] / (2 * C)
        Bv = h * (S[4] - S[3] ** 2 / (2 * C)) / (2 * C - 1)
        with np.errstate(divide="ignore", invalid="ignore"):
            res["rhat"] = np.sqrt(((h - 1) / h * W + Bv / h) / W)
    else:
        res["rhat"] = nan
    if nb >= 2:
        s2_within = S[2] / max(N - C, 1)
        sigma2_bm = batch * S[6] / (C * (nb - 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            res["ess"] = N * s2_within / sigma2_bm
        res["mcse"] = res["sd"] / np.sqrt(res["ess"])
    else:
        res["ess"], res["mcse"] = nan, nan
    return res
