"""On-device streaming posterior statistics (include/logreg_hip.h "Streaming
statistics"): per-chain running (mean, M2) per batch of kept samples, accumulated inside the chain kernels, reduced
over the chains on the device, finished on the host -- against the same quantities computed with NumPy from the
fully gathered samples (what the reference does: fit-np-hmc.py:113-117, analyse.R:17-19)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PSCALE = np.array([10.0, 1, 1, 1, 1, 1, 1, 1])
PRE = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])


@pytest.fixture(scope="module")
def la():
    import logreg_amd
    return logreg_amd


def _check(la, cs, samples, batch):
    from logreg_amd.diagnostics import batch_sums, summary_from_sums
    s = np.asarray(samples, dtype=np.float64)
    iters, Cn, p = s.shape
    sums = cs.stats_sums()
    ref_sums = batch_sums(s, batch, cs.pivot)
    scale = np.abs(ref_sums).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(sums - ref_sums) / scale) < 1e-9
    got = cs.stats_summary()
    flat = s.reshape(-1, p)
    np.testing.assert_allclose(got["mean"], flat.mean(0), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["sd"], flat.std(0, ddof=1), rtol=1e-6)
    if (iters // batch) % 2 == 0 and iters % batch == 0:
        np.testing.assert_allclose(got["rhat"], la.split_rhat(s), rtol=1e-6)
    ref = summary_from_sums(ref_sums, Cn, iters, batch, cs.pivot)
    np.testing.assert_allclose(got["ess"], ref["ess"], rtol=1e-6)
    return got


@pytest.mark.parametrize("kind,mode,group", [("hmc", "auto", 0), ("mala", "reg", 64), ("mala", "reg", 16), ("rwmh", "reg", 16), ("rwmh", "lds", 8), ("ul", "global", 1),
                                             ("hmc", "mfma", 1), ("hmc", "mfma", 4), ("hmc", "stepwise", 0), ("mala", "stepwise", 0),
                                             ("hmc", "mixed", 16)])  # ("mixed": a float64 model under the default precision policy)
def test_device_statistics_equal_numpy_on_the_gathered_samples(la, pima, map_beta, kind, mode, group):
    X, y = pima
    m = la.LogReg(X, y, PSCALE, dtype="float64" if mode == "mixed" else "float32")
    k = {"hmc": lambda: la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=20, dmm=1 / PRE),
         "mala": lambda: la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE),
         "ul": lambda: la.ulKernel(m.glp, dt=1e-6, pre=PRE),
         "rwmh": lambda: la.mhKernel(m.lpost, la.rwProposal(0.02 * np.array([10.0, 1, 1, 1, 1, 1, 5, 1])))}[kind]()
    Cn, iters, thin, batch = 300, 48, 3, 6
    q0 = map_beta + 0.01 * np.random.default_rng(1).standard_normal((Cn, 8))
    cs = la.ChainSet(k, q0, seed=5, mode=mode, group=group)
    cs.enable_stats(batch, iters // batch)
    # three launches whose boundaries fall inside batches: 7 + 20 + 21 kept samples
    parts = [cs.advance(n, thin) for n in (7, 20, 21)]
    samples = np.concatenate([o.to_host() for o in parts])
    _check(la, cs, samples, batch)
    # the same run in one launch with no samples kept: identical accumulators, bit for bit
    cs2 = la.ChainSet(k, q0, seed=5, mode=mode, group=group)
    cs2.enable_stats(batch, iters // batch)
    assert cs2.advance(iters, thin, keep=False) is None
    cs2.sync()
    assert np.array_equal(cs.stats.to_host(), cs2.stats.to_host())
    assert np.array_equal(cs.get_state(), cs2.get_state())


def test_statistics_with_a_partial_last_batch_and_wide_models(la):
    """kept samples not a multiple of the batch (the remainder enters mean/sd only); p = 40 (stepwise, lane per
    coordinate in the update kernel)."""
    n, p, Cn = 300, 40, 130
    X, y, _ = la.synthetic_logreg(n, p, seed=3, beta_sd=0.1)
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=0.02, l=5, dmm=np.ones(p))
    cs = la.ChainSet(k, 0.1 * np.random.default_rng(2).standard_normal((Cn, p)), seed=1)
    cs.enable_stats(4, 6)
    samples = cs.advance(22, 1).to_host()  # 5 full batches + 2
    got = _check(la, cs, samples, 4)
    assert np.all(np.isnan(got["rhat"]))  # odd number of full batches: no whole halves
    with pytest.raises(la.LogregHipError, match="stats buffer too small"):
        cs.advance(3, 1)


@pytest.mark.parametrize("n,p,R", [(200, 11, 4), (1100, 14, 0), (1500, 22, -1)])
def test_statistics_in_the_matrix_core_kernel_for_wider_models(la, n, p, R):
    """padded p = 16 / 32: the lane owning coordinates k + 4h folds them into the accumulators (operands in registers,
    in LDS, in device memory)."""
    X, y, _ = la.synthetic_logreg(n, p, seed=n, beta_sd=0.3 / np.sqrt(p))
    m = la.LogReg(X, y, np.ones(p))
    k = la.hmcKernel(m.lpost, m.glp, eps=0.4 / np.sqrt(n), l=6, dmm=np.ones(p))
    Cn = 90
    cs = la.ChainSet(k, 0.05 * np.random.default_rng(4).standard_normal((Cn, p)), seed=2, mode="mfma", group=4)
    assert cs.plan() == {"mode": "mfma", "group": 4, "rows_per_lane": R}
    cs.enable_stats(5, 4)
    parts = [cs.advance(kk, 2) for kk in (7, 13)]
    _check(la, cs, np.concatenate([o.to_host() for o in parts]), 5)


def test_summary_only_mcmc_never_builds_the_sample_matrix(la, pima, map_beta):
    """mcmc(summary_only=True) == summarising the samples of the same seeded run; and it agrees with the reference
    posterior (F7) like the sample-keeping path does."""
    from conftest import load_golden
    X, y = pima
    m = la.LogReg(X, y, PSCALE)
    k = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=50, dmm=1 / PRE)
    Cn = 2048
    q0 = np.tile(map_beta, (Cn, 1))
    np.random.seed(3)
    warm = la.mcmc(q0, k, thin=2000, iters=1, verb=False, seed=11)[0]
    res = la.mcmc(warm, k, thin=20, iters=64, verb=False, seed=12, summary_only=True)
    samples = la.mcmc(warm, k, thin=20, iters=64, verb=False, seed=12)
    flat = samples.reshape(-1, 8).astype(np.float64)
    np.testing.assert_allclose(res["mean"], flat.mean(0), rtol=1e-9)
    np.testing.assert_allclose(res["sd"], flat.std(0, ddof=1), rtol=1e-6)
    np.testing.assert_allclose(res["rhat"], la.split_rhat(samples), rtol=1e-6)
    assert res["n"] == Cn * 64 and res["batch"] == 4
    assert np.all(np.abs(res["rhat"] - 1) < 0.03)
    # batch-means ESS (batches of 4 kept samples) against Geyer's estimator on the same samples: same order
    geyer = la.ess_pooled(samples, max_chains=128)
    assert np.all(res["ess"] > 0.5 * geyer) and np.all(res["ess"] < 2.0 * geyer)
    ref = load_golden("posterior_hmc.json")["pooled"]
    z = (res["mean"] - np.array(ref["mean"])) / np.sqrt(res["mcse"] ** 2 + np.array(ref["mcse"]) ** 2)
    assert np.max(np.abs(z)) < 3.0
    assert 0.93 < res["accept_rate"] < 0.98


def test_statistics_through_host_pointers(la, pima, map_beta):
    """on_device = 0: the statistics buffer is a HOST array staged by the library, like state/out."""
    from logreg_amd import _lib
    X, y = pima
    m = la.LogReg(X, y, PSCALE)
    k = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=10, dmm=1 / PRE)
    Cn, iters, batch = 70, 12, 3
    q0 = map_beta + 0.01 * np.random.default_rng(4).standard_normal((Cn, 8))
    cs = la.ChainSet(k, q0, seed=2)
    cs.enable_stats(batch, iters // batch)
    cs.advance(iters, 2, keep=False)
    cs.sync()
    st = np.ascontiguousarray(q0, dtype=np.float32)
    host = np.full((iters // batch, Cn, 2, 8), np.nan)
    for first, n in ((0, 5), (5, 7)):  # two calls continue one window
        opts = _lib.RunOpts(n_chains=Cn, thin=2, iters=n, iter_offset=2 * first, seed=2, group=0, mode=_lib.MODE_AUTO,
                            on_device=0, stats=host.ctypes.data, stats_batch=batch, stats_first=first, stats_slots=iters // batch)
        k.launch(opts, st.ctypes.data, None, None, None)
    assert np.array_equal(host, cs.stats.to_host())


def test_statistics_window_survives_checkpoint_and_resume(la, pima, map_beta, tmp_path):
    X, y = pima
    m = la.LogReg(X, y, PSCALE)
    k = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE)
    q0 = map_beta + 0.01 * np.random.default_rng(6).standard_normal((90, 8))
    a = la.ChainSet(k, q0, seed=4)
    a.enable_stats(5, 4)
    a.advance(20, 3, keep=False)
    b = la.ChainSet(k, q0, seed=4)
    b.enable_stats(5, 4)
    b.advance(8, 3, keep=False)  # stop in the middle of batch 1 ...
    c = la.ChainSet.resume(k, b.save(str(tmp_path / "ck")))
    c.advance(12, 3, keep=False)  # ... and continue in another object
    assert np.array_equal(a.stats.to_host(), c.stats.to_host())
    sa, sc = a.stats_summary(), c.stats_summary()
    for key in ("mean", "sd", "rhat", "ess"):
        assert np.array_equal(sa[key], sc[key])


@pytest.mark.parametrize("kind", ["hmc", "mala"])
def test_device_statistics_of_a_run_planned_in_two_parts(la, pima, map_beta, kind, tmp_path):
    """4608 chains: the planner runs the first 4096 on 16 lanes per chain and the last 512 on 64 (two launches per call).  The
    statistics buffer and the sample matrix are indexed by the chain's position in the CALL, whichever launch advances it:
    device statistics = NumPy on the gathered samples; a checkpoint taken between launches resumes bit for bit."""
    X, y = pima
    m = la.LogReg(X, y, PSCALE)
    k = (la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=10, dmm=1 / PRE) if kind == "hmc" else la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE))
    Cn, iters, thin, batch = 4608, 12, 2, 3
    q0 = map_beta + 0.01 * np.random.default_rng(2).standard_normal((Cn, 8))
    cs = la.ChainSet(k, q0, seed=6, precision="full")
    assert cs.plan()["tail"] == {"from": 4096, "mode": "reg", "group": 64, "rows_per_lane": 4}
    cs.enable_stats(batch, iters // batch)
    first = cs.advance(5, thin)
    path = cs.save(tmp_path / "two_part")
    rest = cs.advance(7, thin)
    samples = np.concatenate([first.to_host(), rest.to_host()])
    _check(la, cs, samples, batch)
    resumed = la.ChainSet.resume(k, path, precision="full")
    assert resumed.plan() == cs.plan()
    again = resumed.advance(7, thin).to_host()
    assert np.array_equal(again, rest.to_host())
    np.testing.assert_array_equal(resumed.stats_sums(), cs.stats_sums())
