"""Pin the CPU oracle (oracle/lr_oracle.c) against golden vectors captured from the reference's
own NumPy functions (tests/golden/make_fixtures.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc


def test_philox_known_answers():  # F9
    for kat in load_golden("philox_kat.json")["kat"]:
        assert orc.philox4x32_10(kat["ctr"], kat["key"]) == kat["out"]


def test_model_values_match_reference(oracle_model):  # F1
    g = load_golden("model_eval.json")
    beta = np.array(g["beta"])
    with np.errstate(all="ignore"):
        for nm in ("ll", "lprior", "lpost"):
            ref = np.array(g[nm], dtype=np.float64)
            got = getattr(oracle_model, nm)(beta)
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin)
            np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-12, atol=0)
            assert np.array_equal(got[~fin], ref[~fin])  # -inf where the naive form overflows
        ref = np.array(g["glp"])
        got = oracle_model.glp(beta)
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-9)


def test_survey_example_values(oracle_model):
    b = np.array([-9.0, 0.1, 0.03, -0.01, 0.0, 0.08, 1.5, 0.03])
    assert oracle_model.ll(b) == pytest.approx(-93.29888360251877, rel=1e-13)
    assert oracle_model.lprior(b) == pytest.approx(-11.193243358631427, rel=1e-13)
    assert oracle_model.lpost(b) == pytest.approx(-104.4921269611502, rel=1e-13)


def test_map_is_stationary(oracle_model):  # F2
    g = load_golden("map.json")
    m = np.array(g["map"])
    assert oracle_model.lpost(m) == pytest.approx(g["lpost_map"], rel=1e-13)
    assert oracle_model.lpost(m) == pytest.approx(-100.44943693563212, abs=1e-6)
    np.testing.assert_allclose(oracle_model.glp(m), np.array(g["glp_map"]), atol=1e-8)
    assert np.linalg.norm(oracle_model.glp(m)) < 0.5  # BFGS tolerance on raw-scale covariates


def test_leapfrog_matches_reference(oracle_model):  # F3
    g = load_golden("leapfrog.json")
    dmm = np.array(g["dmm"])
    for q0, p0, q1, p1, a0, a1 in zip(g["q0"], g["p0"], g["q1"], g["p1_negated"], g["alpi0"], g["alpi1"]):
        q, p = oracle_model.leapfrog(q0, p0, g["eps"], g["l"], dmm)
        np.testing.assert_allclose(q, q1, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(p, p1, rtol=1e-10, atol=1e-12)
        assert oracle_model.alpi(q0, p0, dmm) == pytest.approx(a0, rel=1e-13)
        assert oracle_model.alpi(q, p, dmm) == pytest.approx(a1, rel=1e-12)
    s = g["short"]
    q, p = oracle_model.leapfrog(s["q0"], s["p0"], s["eps"], s["l"], np.array(s["dmm"]))
    np.testing.assert_allclose(q, s["q1"], rtol=1e-12)
    np.testing.assert_allclose(p, s["p1_negated"], rtol=1e-11, atol=1e-13)
    # SURVEY example: alpi at the MAP with the fixed p0
    assert g["alpi0"][0] == pytest.approx(-102.52068693563213, abs=1e-6)
    assert g["alpi1"][0] == pytest.approx(-102.52002008932209, abs=1e-6)


def test_mala_terms(oracle_model):  # F4
    g = load_golden("mala_terms.json")
    pre = np.array(g["pre"])
    for i in range(len(g["x"])):
        x = np.array(g["x"][i]); z = np.array(g["z"][i])
        # u=1 -> log u = 0: margin == a
        xn, lln, acc, margin = oracle_model.mala_step(x, g["lpost_x"][i], g["dt"], pre, z, 1.0 - 2.0**-25)
        a = g["a"][i]
        assert margin - (-np.log(1.0 - 2.0**-25)) == pytest.approx(a, rel=1e-9, abs=1e-9)
        if acc:
            np.testing.assert_allclose(xn, g["prop"][i], rtol=1e-13)
            assert lln == pytest.approx(g["lpost_prop"][i], rel=1e-13)
        else:
            assert a <= 0


def test_rwmh_terms(oracle_model):  # F5
    g = load_golden("rwmh_terms.json")
    sd = np.array(g["prop_sd"])
    np.testing.assert_allclose(sd, 0.02 * np.array([10.0, 1, 1, 1, 1, 1, 5, 1]))
    for i in range(len(g["x"])):
        xn, lln, acc, margin = oracle_model.rwmh_step(g["x"][i], g["lpost_x"][i], sd, g["z"][i], 1.0 - 2.0**-25)
        assert margin + np.log(1.0 - 2.0**-25) == pytest.approx(g["a"][i], rel=1e-9, abs=1e-9)
        if acc:
            np.testing.assert_allclose(xn, g["prop"][i], rtol=1e-14)


@pytest.mark.parametrize("kind", ["hmc", "mala", "rwmh", "ul"])
def test_accept_replay(oracle_model, kind):  # F6: control flow of mcmc() + each kernel
    g = load_golden("accept_replay.json")
    rec = g[kind]
    par = g["params"][kind]
    z = np.array(rec["normals"])
    steps = z.shape[0]
    u = np.array(rec["uniforms"]) if kind != "ul" else None
    kw = {"hmc": dict(step=par.get("eps"), l=par.get("l"), scale=par.get("dmm")),
          "mala": dict(step=par.get("dt"), scale=par.get("pre")),
          "ul": dict(step=par.get("dt"), scale=par.get("pre")),
          "rwmh": dict(scale=par.get("prop_sd"))}[kind]
    ref = np.array(rec["states"])
    init = np.array(rec["init"])
    prev = np.vstack([init[None], ref[:-1]])
    moved_ref = np.any(ref != prev, axis=1)
    # (1) teacher-forced: every recorded step reproduced from the reference's own previous state
    for t in range(steps):
        if kind == "hmc":
            xn, acc, _ = oracle_model.hmc_step(prev[t], kw["step"], kw["l"], kw["scale"], z[t], u[t])
        elif kind == "ul":
            xn, acc = oracle_model.ul_step(prev[t], kw["step"], kw["scale"], z[t]), True
        else:
            ll = -np.inf if t == 0 else oracle_model.lpost(prev[t])
            fn = oracle_model.mala_step if kind == "mala" else oracle_model.rwmh_step
            args = (kw["step"], kw["scale"]) if kind == "mala" else (kw["scale"],)
            xn, _, acc, _ = fn(prev[t], ll, *args, z[t], u[t])
        assert bool(acc) == bool(moved_ref[t]), t
        np.testing.assert_allclose(xn, ref[t], rtol=1e-11, atol=1e-13)
    # (2) free-running: the whole mcmc() loop.  MALA at the reference's dt=1e-5 has an expansive
    # drift map on raw-scale Pima (|1 - dt*pre*lambda_max/2| ~ 7 per ACCEPTED step), so last-bit
    # BLAS-vs-loop differences grow ~7x per acceptance: compare a prefix only for MALA.
    span = 60 if kind == "mala" else steps
    r = oracle_model.run(kind, init, thin=1, iters=steps, ext_z=z[:, None, :],
                         ext_u=None if u is None else u[:, None], **kw)
    out = r["out"]
    np.testing.assert_allclose(out[:span], ref[:span], rtol=1e-9 if kind != "mala" else 1e-6, atol=1e-11)
    prev_o = np.vstack([init[None], out[:-1]])
    assert np.array_equal(np.any(out != prev_o, axis=1)[:span], moved_ref[:span])
    if kind != "mala":
        assert int(r["accepts"][0]) == int(moved_ref.sum())
    if kind in ("mala", "rwmh"):
        assert moved_ref[0]  # ll starts at -inf: the first proposal is always accepted
    # thin indexing: row i == state after (i+1)*thin steps
    r4 = oracle_model.run(kind, rec["init"], thin=4, iters=steps // 4, ext_z=z[:, None, :],
                          ext_u=None if u is None else u[:, None], **kw)
    np.testing.assert_array_equal(r4["out"], out[3::4])


def test_philox_stream_is_chunk_and_shard_invariant(oracle_model, map_beta):
    dmm = 1.0 / np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    init = np.tile(map_beta, (6, 1))
    full = oracle_model.run("hmc", init, step=1e-3, l=5, scale=dmm, thin=2, iters=4, seed=9)
    a = oracle_model.run("hmc", init[:3], step=1e-3, l=5, scale=dmm, thin=2, iters=2, seed=9)
    b = oracle_model.run("hmc", a["state"], step=1e-3, l=5, scale=dmm, thin=2, iters=2, seed=9, iter_offset=4)
    c = oracle_model.run("hmc", init[3:], step=1e-3, l=5, scale=dmm, thin=2, iters=4, seed=9, chain_offset=3)
    np.testing.assert_array_equal(full["out"][:2, :3], a["out"])
    np.testing.assert_array_equal(full["out"][2:, :3], b["out"])
    np.testing.assert_array_equal(full["out"][:, 3:], c["out"])


def test_stream_moments():
    z = np.array([orc.draws(123, c, 0, 8)[0] for c in range(20000)])
    u = np.array([orc.draws(123, c, 0, 8)[1] for c in range(20000)])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert np.all(np.abs(np.corrcoef(z.T) - np.eye(8)) < 0.03)
    assert abs(u.mean() - 0.5) < 0.01 and u.min() > 0 and u.max() < 1


def test_oracle_acceptance_rates_match_reference(oracle_model, map_beta):  # F8b
    g = load_golden("accept_rates.json")
    pre = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    C = 64
    init = np.tile(map_beta, (C, 1))
    r = oracle_model.run("hmc", init, step=1e-3, l=50, scale=1 / pre, thin=1, iters=60, seed=5, keep=False, threads=0)
    rate = r["accepts"].sum() / (C * 60)
    assert abs(rate - g["hmc"]["rate"]) < 0.02
    r = oracle_model.run("mala", init, step=1e-5, scale=pre, thin=1, iters=2000, seed=5, keep=False, threads=0)
    rate = r["accepts"].sum() / (C * 2000)
    assert abs(rate - g["mala"]["rate"]) < 0.03
    r = oracle_model.run("rwmh", init, scale=0.02 * np.array([10.0, 1, 1, 1, 1, 1, 5, 1]), thin=1, iters=3000,
                         seed=5, keep=False, threads=0)
    rate = r["accepts"].sum() / (C * 3000)
    assert abs(rate - g["rwmh"]["rate"]) < 0.01


# ---- shapes other than Pima's: vectors the reference's own closures produced on synthetic designs put in place of
# ---- its globals X, y, pscale (tests/golden/make_shape_fixtures.py): BASELINE configs 4 and 5 at full size, one mid shape
@pytest.mark.parametrize("name", ["cfg4", "cfg5", "mid"])
def test_oracle_matches_reference_at_other_shapes(name):
    from logreg_amd.data import synthetic_logreg
    g = load_golden(f"shape_{name}.json")
    X, y, _ = synthetic_logreg(g["n"], g["p"], seed=g["data_seed"], beta_sd=g["beta_sd"])
    m = orc.OracleModel(X, y, np.array(g["pscale"]))
    beta = np.array(g["beta"])
    for nm in ("ll", "lprior", "lpost"):
        np.testing.assert_allclose(getattr(m, nm)(beta), np.array(g[nm]), rtol=1e-11, atol=0)
    ref = np.array(g["glp"])
    # a gradient coordinate is a sum of n terms that largely cancel: tolerance relative to the sum of the terms'
    # magnitudes (<= sum_i |x_ij|), as everywhere (1e-4 of the term sum for fp32; here float64, 1e-11 would do)
    scale = np.abs(X).sum(axis=0)
    assert np.max(np.abs(m.glp(beta) - ref) / scale) < 1e-13
    lf = g["leap"]
    dmm = np.array(lf["dmm"])
    q, p = m.leapfrog(lf["q0"], lf["p0"], lf["eps"], lf["l"], dmm)
    np.testing.assert_allclose(q, lf["q1"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(p, lf["p1_negated"], rtol=1e-9, atol=1e-10)
    assert m.alpi(lf["q0"], lf["p0"], dmm) == pytest.approx(lf["alpi0"], rel=1e-12)
    assert m.alpi(q, p, dmm) == pytest.approx(lf["alpi1"], rel=1e-11)


def test_oracle_ul_posterior_matches_the_reference_run(oracle_model, map_beta):
    """Unadjusted Langevin end to end on the CPU restatement against the seeded full runs of the unmodified fit-np-ul.py
    (tests/golden/posterior_ul.json: dt = 1e-6, pre, thin 2000; UL is biased by construction, so the target is the
    reference's own output): 96 chains from the MAP, 200 000 iterations dropped, 40 kept x thin 2000, standard errors from
    the spread between the independent chains, 3.5 combined standard errors over the 16 statistics."""
    pre = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    C = 96
    r = oracle_model.run("ul", np.tile(map_beta, (C, 1)), step=1e-6, scale=pre, thin=200000, iters=1, seed=17, threads=0)
    r = oracle_model.run("ul", r["state"], step=1e-6, scale=pre, thin=2000, iters=40, seed=17, iter_offset=200000, threads=0)
    s = r["out"]
    ref = load_golden("posterior_ul.json")["pooled"]
    mean, sd = s.reshape(-1, 8).mean(axis=0), s.reshape(-1, 8).std(axis=0, ddof=1)
    mcse = s.mean(axis=0).std(axis=0, ddof=1) / np.sqrt(C)
    se_sd = ((s - mean) ** 2).mean(axis=0).std(axis=0, ddof=1) / np.sqrt(C) / (2 * sd)
    zm = (mean - np.array(ref["mean"])) / np.sqrt(mcse ** 2 + np.array(ref["mcse"]) ** 2)
    zs = (sd - np.array(ref["sd"])) / np.sqrt(se_sd ** 2 + np.array(ref["se_sd"]) ** 2)
    print("oracle UL z(mean)", np.round(zm, 2), "z(sd)", np.round(zs, 2))
    assert np.max(np.abs(zm)) < 3.5 and np.max(np.abs(zs)) < 3.5
