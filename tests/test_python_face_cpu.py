"""The host side of the product in the GPU-less container: logreg_amd's Python face (model closures, kernel objects, ChainSet
chunking / checkpoint / streaming statistics, mcmc()) and the plain-C client, run against the CPU TEST DOUBLE of the C ABI
(tests/host/lr_cpu_twin.c, injected by tests/twin.py for this module only).  What is checked here is everything ABOVE the ABI
-- argument plumbing, iteration counters, thinning, chunk invariance, checkpoints, the statistics pipeline, output formats --
against the oracle called directly and against the reference's fixtures; the kernels themselves are the GPU tests' business.
The product is unchanged by this: it loads liblogreg_hip.so only (tests/test_abi.py checks that it fails loudly without a GPU).
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden

import twin


@pytest.fixture(scope="module", autouse=True)
def _abi_twin():
    twin.install()
    yield
    twin.uninstall()


@pytest.fixture(scope="module")
def la():
    import logreg_amd
    return logreg_amd


@pytest.fixture(scope="module")
def models(la, pima, pscale):
    X, y = pima
    return {d: la.LogReg(X, y, pscale, dtype=d) for d in ("float32", "float64")}


PRE = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])


def test_model_closures_through_the_python_face(models):  # F1
    g = load_golden("model_eval.json")
    beta = np.array(g["beta"])
    m = models["float64"]
    r = m.eval(beta)
    for nm in ("ll", "lprior", "lpost", "glp"):
        np.testing.assert_allclose(r[nm], np.array(g[nm]), rtol=1e-12, atol=1e-9)
    b = beta[2]  # reference call shapes: beta [p] -> float / ndarray [p]
    assert isinstance(m.lpost(b), float) and m.glp(b).shape == (8,) and isinstance(m.ll(b), float)
    assert m.ll(b) == pytest.approx(-93.29888360251877, rel=1e-12)
    assert set(m.eval(b, ("ll", "glp"))) == {"ll", "glp"}
    assert m.interior_format() == "none" and m.debug_opts() == ""  # (the test double computes everything in float64)
    with pytest.raises(ValueError):
        m.eval(np.zeros(7))
    # a float32 model hands float32 values back through the same face
    r32 = models["float32"].eval(beta)
    np.testing.assert_allclose(r32["lpost"], np.array(g["lpost"]), rtol=2e-5)


@pytest.mark.parametrize("kind", ["rwmh", "mala", "hmc", "ul"])
def test_mcmc_is_the_oracle_run_with_the_same_stream(la, models, oracle_model, map_beta, kind):
    """mcmc(init, kernel, thin, iters, seed=) through ChainSet and the ABI == one direct oracle run: parameters, thinning
    (`mat[i]` = state after (i + 1) thin iterations), chain ids, the -inf start of the threaded kernels."""
    m = models["float64"]
    kern = {"rwmh": lambda: la.mhKernel(m.lpost, la.rwProposal(0.02 * np.sqrt(PRE))),
            "mala": lambda: la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE),
            "hmc": lambda: la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=7, dmm=1 / PRE),
            "ul": lambda: la.ulKernel(m.glp, dt=1e-6, pre=PRE)}[kind]()
    assert isinstance(kern, la.kernels.FusedKernel)
    C, thin, iters, seed = 5, 3, 6, 1234
    init = map_beta + 0.01 * np.random.default_rng(2).standard_normal((C, 8))
    mat, info = la.mcmc(init, kern, thin=thin, iters=iters, verb=False, seed=seed, chain_offset=11, return_info=True, chunk=4)
    step, scale, l = {"rwmh": (0.0, 0.02 * np.sqrt(PRE), 0), "mala": (1e-5, PRE, 0), "hmc": (1e-3, 1 / PRE, 7), "ul": (1e-6, PRE, 0)}[kind]
    st = init.copy()
    ref = oracle_model.run(kind, st, step=step, l=l, scale=scale, thin=thin, iters=iters, seed=seed, chain_offset=11, keep=True)
    np.testing.assert_allclose(mat, ref["out"], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(info["accepts"], ref["accepts"])
    assert info["iterations"] == thin * iters and mat.shape == (iters, C, 8)
    # one chain, reference shapes: [p] in, float64 [iters, p] out
    one = la.mcmc(init[0], kern, thin=thin, iters=iters, verb=False, seed=seed, chain_offset=11)
    assert one.shape == (iters, 8) and one.dtype == np.float64
    np.testing.assert_allclose(one, ref["out"][:, 0, :], atol=1e-13)


def test_chunked_and_resumed_runs_continue_bit_for_bit(la, models, map_beta, tmp_path):
    for dtype in ("float32", "float64"):
        m = models[dtype]
        kern = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE)
        init = np.tile(map_beta, (4, 1))
        whole = la.mcmc(init, kern, thin=5, iters=12, verb=False, seed=9)
        for chunk in (1, 5, 12):
            np.testing.assert_array_equal(la.mcmc(init, kern, thin=5, iters=12, verb=False, seed=9, chunk=chunk), whole)
        # ChainSet: advance in pieces, checkpoint to disk in the middle, resume in a "new process"
        cs = la.ChainSet(kern, init, seed=9)
        a = cs.advance(5, 5).to_host()
        path = cs.save(tmp_path / f"ck_{dtype}")
        assert path.endswith(".npz") and os.path.exists(path)
        cs2 = la.ChainSet.resume(kern, path)
        b = cs2.advance(7, 5).to_host()
        np.testing.assert_array_equal(np.concatenate([a, b]), whole)
        assert cs2.iter_offset == 60 and cs2.get_accepts().sum() > 0
        # a checkpoint of another kernel / model is refused
        other = la.malaKernel(m.lpost, m.glp, dt=2e-5, pre=PRE)
        with pytest.raises(ValueError, match="param_dt"):
            la.ChainSet.resume(other, path)
        with pytest.raises(ValueError, match="hmc"):
            la.ChainSet.resume(la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=3, dmm=1 / PRE), path)


def test_single_step_calls_have_the_reference_signatures(la, models, map_beta):
    m = models["float64"]
    np.random.seed(5)
    k = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE)
    x0 = map_beta.copy()
    x, ll = k(x0, -np.inf)                            # kernel(x, ll) -> (x, ll): fit-np-mala.py:61-70
    assert x.shape == (8,) and isinstance(ll, float) and np.isfinite(ll)
    np.testing.assert_array_equal(x0, map_beta)       # the caller's x is not touched (a float64 model once stepped it in place)
    assert not np.array_equal(x, x0)
    assert ll == pytest.approx(m.lpost(x), rel=1e-12)  # -inf start: the first proposal is always accepted
    h = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=5, dmm=1 / PRE)
    assert h(map_beta).shape == (8,)                  # kern(q) -> q: fit-np-hmc.py:65-87
    xs, lls = k(np.tile(map_beta, (3, 1)), np.full(3, -np.inf))
    assert xs.shape == (3, 8) and lls.shape == (3,)


def test_mcmc_prints_and_seeds_like_the_reference(la, models, map_beta, capsys):
    m = models["float64"]
    kern = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=3, dmm=1 / PRE)
    np.random.seed(3)
    a = la.mcmc(map_beta, kern, thin=2, iters=4)
    out = capsys.readouterr().out
    assert out.startswith("4 iterations\n") and "Done." in out  # fit-np-hmc.py:91-102
    np.random.seed(3)
    b = la.mcmc(map_beta, kern, thin=2, iters=4, verb=False)
    np.testing.assert_array_equal(a, b)                # np.random.seed(s) makes a run reproducible, as for the reference
    assert capsys.readouterr().out == ""


def test_summary_only_equals_numpy_on_the_samples(la, models, map_beta):
    """mcmc(summary_only=True): the streaming statistics of the ABI + lr_stats_reduce + diagnostics.summary_from_sums against the
    same statistics computed from the full sample matrix."""
    m = models["float64"]
    kern = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=PRE)
    C, thin, iters = 6, 20, 96
    init = map_beta + 0.02 * np.random.default_rng(4).standard_normal((C, 8))
    res = la.mcmc(init, kern, thin=thin, iters=iters, verb=False, seed=77, summary_only=True, max_batches=16, chunk=10)
    mat = la.mcmc(init, kern, thin=thin, iters=iters, verb=False, seed=77)
    pooled = mat.reshape(-1, 8)
    np.testing.assert_allclose(res["mean"], pooled.mean(axis=0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(res["sd"], pooled.std(axis=0, ddof=1), rtol=1e-8)
    np.testing.assert_allclose(res["rhat"], la.split_rhat(mat), rtol=1e-8)
    from logreg_amd.diagnostics import batch_sums, summary_from_sums
    ref = summary_from_sums(batch_sums(mat, res["batch"], init[0]), C, iters, res["batch"], init[0])
    np.testing.assert_allclose(res["ess"], ref["ess"], rtol=1e-8)
    np.testing.assert_allclose(res["state"], mat[-1], atol=0)
    assert 0.0 < res["accept_rate"] <= 1.0 and res["plan"]["mode"] == "global"


def test_errors_reach_python_as_exceptions(la, models, map_beta):
    m = models["float64"]
    with pytest.raises(la.LogregHipError, match="prop_sd"):
        la.mcmc(map_beta, la.mhKernel(m.lpost, la.rwProposal(np.zeros(8))), thin=1, iters=1, verb=False, seed=1)
    with pytest.raises(la.LogregHipError, match="thin"):
        la.ChainSet(la.ulKernel(m.glp, dt=1e-6, pre=PRE), map_beta, seed=1).advance(1, 0)
    with pytest.raises(ValueError):
        la.LogReg(np.zeros((5, 2)), np.zeros(4), 1.0)
    with pytest.raises(la.LogregHipError, match="y\\["):
        la.LogReg(np.ones((3, 2)), np.array([0.0, 2.0, 1.0]), 1.0)


def test_find_map_and_output_writer_on_the_python_face(la, models, map_beta, tmp_path):
    m = models["float64"]
    b, info = la.find_map(m)
    assert info["converged"] and np.max(np.abs(b - map_beta)) < 1e-5            # F2
    assert info["lpost"] == pytest.approx(-100.44943693563212, rel=1e-10)
    kern = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=5, dmm=1 / PRE)
    out = la.mcmc(b, kern, thin=2, iters=10, verb=False, seed=2)
    path = str(tmp_path / "fit-np-hmc.parquet")
    la.write_parquet(out, path)
    import pandas as pd
    assert list(pd.read_parquet(path).columns) == [f"b{j}" for j in range(8)]    # fit-np-hmc.py:111-112
    np.testing.assert_array_equal(la.read_parquet(path), out)


def test_plain_c_client_on_the_cpu(tmp_path):
    """examples/fit_bayes.c (the counterpart of the reference's C/fit-bayes.c) linked against the CPU test double: the program's
    own logic -- reading the data file, the run parameters of C/fit-bayes.c, the output format of :104-118 -- and a short chain's
    posterior against the reference's seeded RWMH run (a native RWMH CLI on the CPU restatement)."""
    lib = twin.build()
    exe = tmp_path / "fit_bayes"
    subprocess.run(["gcc", "-O2", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "fit_bayes.c"), lib,
                    "-Wl,-rpath," + os.path.dirname(lib), "-lm", "-o", str(exe)], check=True, capture_output=True)
    r = subprocess.run([str(exe), os.path.join(REPO, "logreg_amd", "data", "Pima.tr.txt"), "400", "250"], capture_output=True, text=True,
                       check=True)
    lines = r.stdout.strip().split("\n")
    assert lines[0].split() == [f"beta{j}" for j in range(8)]  # C/fit-bayes.c:104-107
    a = np.array([[float(v) for v in ln.split()] for ln in lines[1:]])
    assert a.shape == (400, 8) and np.isfinite(a).all()
    ref = load_golden("posterior_rwmh.json")["pooled"]
    # 10^5 iterations of one chain: a coarse check (|z| < 5 with the chain's own Geyer MCSE) that it samples the right posterior
    import logreg_amd as la
    summ = la.summarise(a[100:, None, :], max_chains=None)
    z = (summ["mean"] - np.array(ref["mean"])) / np.sqrt(summ["mcse"] ** 2 + np.array(ref["mcse"]) ** 2)
    assert np.max(np.abs(z)) < 5.0, z


def _worker_sharded_on_twin(rank, world, port, C, tmp):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import twin as tw
    tw.install()
    import logreg_amd as la
    from logreg_amd.distributed import mcmc_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y = la.load_pima()
    init = np.load(os.path.join(tmp, "init.npy"))

    def make_kernel(dev):
        m = la.LogReg(X, y, [10, 1, 1, 1, 1, 1, 1, 1], dtype="float64", device=0)
        return la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=5, dmm=1 / PRE)

    class HostChainSet(la.ChainSet):  # the product's ChainSet; only the hand-over to torch differs (no CUDA tensors here)
        def advance(self, iters, thin, keep=True, **kw):
            out = super().advance(iters, thin, keep=keep, **kw)
            return out.to_host() if keep else None
    out = mcmc_sharded(init, make_kernel, thin=2, iters=8, seed=31, chunk=3, chainset_factory=HostChainSet)
    glob = mcmc_sharded(init, make_kernel, thin=2, iters=8, seed=31, chunk=5, chainset_factory=HostChainSet, plan="global", precision="full")
    summ = mcmc_sharded(init, make_kernel, thin=2, iters=8, seed=31, summary_only=True, max_batches=4, chainset_factory=HostChainSet)
    if rank == 0:
        assert np.array_equal(out.numpy(), glob.numpy())
        np.save(os.path.join(tmp, "gathered.npy"), out.numpy())
    else:
        assert out is None and glob is None
    np.savez(os.path.join(tmp, f"summary{rank}.npz"), **{k: v for k, v in summ.items()})
    dist.barrier()
    dist.destroy_process_group()
    tw.uninstall()


@pytest.mark.parametrize("C", [7, 1])
def test_mcmc_sharded_with_the_products_chainset(tmp_path, oracle_model, map_beta, C):
    """world size 2 over gloo, the product's own ChainSet on the ABI twin (tests/test_distributed_gloo.py drives mcmc_sharded
    with a stand-in chain set): ragged shards (4 + 3) and an empty one, global chain ids, chunked launches, the gather and the
    statistics all-reduce -- equal to ONE oracle run of all chains."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    init = map_beta + 0.01 * np.random.default_rng(1).standard_normal((C, 8))
    np.save(tmp_path / "init.npy", init)
    mp.spawn(_worker_sharded_on_twin, args=(2, port, C, str(tmp_path)), nprocs=2, join=True)
    ref = oracle_model.run("hmc", init, step=1e-3, l=5, scale=1 / PRE, thin=2, iters=8, seed=31)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == (8, C, 8)
    np.testing.assert_allclose(got, ref["out"], rtol=0, atol=1e-13)
    flat = ref["out"].reshape(-1, 8)
    for rank in range(2):
        sm = np.load(tmp_path / f"summary{rank}.npz")
        assert int(sm["n"]) == flat.shape[0] and int(sm["chains"]) == C
        np.testing.assert_allclose(sm["mean"], flat.mean(0), rtol=1e-11)
        np.testing.assert_allclose(sm["sd"], flat.std(0, ddof=1), rtol=1e-8)
        assert float(sm["accept_rate"]) == pytest.approx(ref["accepts"].sum() / (C * 16))


@pytest.mark.parametrize("kind", ["hmc", "mala", "rwmh"])
def test_reference_composition_with_our_closures_replays_the_recorded_run(la, models, kind):  # F6 through the Python face
    """The drop-in claim itself: the reference's higher-order functions (mhKernel / malaKernel / hmcKernel / mcmc, generic path)
    composed with OUR model closures; seeding NumPy as the fixture generator did reproduces the reference's recorded states,
    because the generic kernels draw randn / rand in the reference's order (fit-np-hmc.py:56-103, fit-np-mala.py:61-95)."""
    g = load_golden("accept_replay.json")[kind]
    m = models["float64"]
    lpost, glp = (lambda b: m.lpost(b)), (lambda b: m.glp(b))  # plain callables: forces the generic path
    if kind == "hmc":
        kern = la.hmcKernel(lpost, glp, eps=1e-3, l=50, dmm=1 / PRE)
    elif kind == "mala":
        kern = la.malaKernel(lpost, glp, dt=1e-5, pre=PRE)
    else:
        pre = np.array([10.0, 1, 1, 1, 1, 1, 5, 1])
        kern = la.mhKernel(lpost, lambda beta: beta + 0.02 * pre * np.random.randn(8))  # fit-numpy.py:81-84
    assert not isinstance(kern, la.kernels.FusedKernel)
    np.random.seed(1000 + len(kind))
    nrep = 24
    out = la.mcmc(np.array(g["init"]), kern, thin=1, iters=nrep, verb=False)
    np.testing.assert_allclose(out, np.array(g["states"])[:nrep], rtol=1e-9, atol=1e-11)


def test_example_script_runs_end_to_end(tmp_path):
    """examples/fit_hmc.py (the reference's fit-np-hmc.py on the drop-in) as a child process with the twin injected through
    sitecustomize-free means: a two-line launcher that installs the twin and runs the script."""
    launcher = tmp_path / "run.py"
    out = tmp_path / "fit.parquet"
    launcher.write_text(
        "import runpy, sys\n"
        f"sys.path[:0] = [{REPO!r}, {os.path.join(REPO, 'tests')!r}]\n"
        "import twin; twin.install()\n"
        f"sys.argv = ['fit_hmc.py', '--iters', '30', '--out', {str(out)!r}]\n"
        f"runpy.run_path({os.path.join(REPO, 'examples', 'fit_hmc.py')!r}, run_name='__main__')\n"
        "twin.uninstall()\n")
    import sys
    r = subprocess.run([sys.executable, str(launcher)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "MAP:" in r.stdout and "HMC:" in r.stdout and "30 iterations" in r.stdout and "Done." in r.stdout and "ESS:" in r.stdout
    import pandas as pd
    df = pd.read_parquet(out)
    assert df.shape == (30, 8) and list(df.columns) == [f"b{j}" for j in range(8)] and np.isfinite(df.to_numpy()).all()


def test_chainset_edges(la, models, map_beta):
    m = models["float32"]
    kern = la.mhKernel(m.lpost, la.rwProposal(0.02 * np.sqrt(PRE)))
    # a given threaded log-density is used (no free first acceptance), chain ids offset the stream
    ll0 = float(m.lpost(map_beta))
    a = la.mcmc(map_beta, kern, thin=1, iters=20, verb=False, seed=4, ll=ll0, chain_offset=3)
    b = la.mcmc(map_beta, kern, thin=1, iters=20, verb=False, seed=4, chain_offset=3)
    assert not np.array_equal(a, b)          # -inf start accepts the first proposal, the threaded value may not
    c = la.mcmc(np.tile(map_beta, (5, 1)), kern, thin=1, iters=20, verb=False, seed=4, ll=ll0)
    np.testing.assert_array_equal(c[:, 3, :].astype(np.float64), a)   # chain 3 of a 5-chain run = the one-chain run at chain_offset 3
    # device-array views and stats misuse
    cs = la.ChainSet(kern, np.tile(map_beta, (4, 1)), seed=1)
    out = cs.advance(6, 2)
    np.testing.assert_array_equal(out.rows(2, 5).to_host(), out.to_host()[2:5])
    with pytest.raises(ValueError, match="enable_stats"):
        cs.advance(1, 1, stats=True)
    with pytest.raises(ValueError, match="statistics window"):
        cs.stats_sums()
    cs.enable_stats(batch=2, slots=2)
    cs.advance(4, 1, keep=False)
    with pytest.raises(la.LogregHipError, match="window"):
        cs.advance(1, 1, keep=False)     # the window is full: the ABI refuses instead of writing past the buffer
    with pytest.raises(KeyError):
        la.ChainSet(kern, map_beta, seed=1, precision="half")
    mm = la.LogReg(np.ones((3, 2)), np.array([0.0, 1.0, 1.0]), 1.0)
    mm.close()
    with pytest.raises(la.LogregHipError, match="closed"):
        mm.lpost(np.zeros(2))


def test_plain_c_client_under_the_sanitizers(tmp_path):
    """The C client and the ABI test double compiled into ONE program with AddressSanitizer + UndefinedBehaviorSanitizer (CPU only:
    GPU sanitizers are not available on the pool): the example's buffer handling, the twin and the oracle underneath run clean."""
    exe = tmp_path / "fit_bayes_asan"
    cc = subprocess.run(["gcc", "-O1", "-g", "-std=gnu99", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(REPO, "include"),
                         os.path.join(REPO, "examples", "fit_bayes.c"), os.path.join(REPO, "tests", "host", "lr_cpu_twin.c"), "-lm", "-o", str(exe)],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert cc.returncode == 0, cc.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([str(exe), os.path.join(REPO, "logreg_amd", "data", "Pima.tr.txt"), "40", "25"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert len(r.stdout.strip().split("\n")) == 41


def test_the_scripts_own_rprop_is_recognised_and_fused(la, models, map_beta):
    """fit-numpy.py:81-86 passes a Python FUNCTION `rprop` to mhKernel.  The constructor probes it (zeros, unit vectors, random
    draws through a stand-in for np.random.randn) and fuses it when it is `beta + sd * randn(p)`: the reference's call runs
    unchanged on the fused kernel, with the same samples as the proposal stated as data; other proposals stay generic and the
    probing consumes nothing from NumPy's global generator."""
    from logreg_amd import kernels as K
    m = models["float64"]
    pre = np.array([10.0, 1, 1, 1, 1, 1, 5, 1])
    p = 8

    def rprop(beta):  # the reference's text
        return beta + 0.02 * pre * np.random.randn(p)
    np.random.seed(5)
    before = np.random.get_state()[1].copy()
    k = la.mhKernel(m.lpost, rprop)
    assert np.array_equal(np.random.get_state()[1], before)  # the probes drew nothing from the global generator
    assert isinstance(k, K.FusedKernel) and k.kind == "rwmh"
    np.testing.assert_allclose(k.params["prop_sd"], 0.02 * pre, rtol=1e-15)
    # F5 through it: the recorded (x, z) pairs of the reference reproduce its proposals with the recognised scale
    g = load_golden("rwmh_terms.json")
    np.testing.assert_allclose(np.array(g["x"]) + k.params["prop_sd"] * np.array(g["z"]), np.array(g["prop"]), rtol=1e-15)
    np.testing.assert_allclose(m.lpost(np.array(g["prop"])) - m.lpost(np.array(g["x"])), np.array(g["a"]), rtol=1e-9, atol=1e-9)
    a = la.mcmc(map_beta, k, thin=7, iters=30, verb=False, seed=3)
    b = la.mcmc(map_beta, la.mhKernel(m.lpost, la.rwProposal(0.02 * pre)), thin=7, iters=30, verb=False, seed=3)
    assert np.array_equal(a, b)
    # not of the form: dense scale, state-dependent scale, another generator, a drift -- all stay generic
    rng = np.random.default_rng(0)
    A = np.eye(p) + 0.1
    for other in (lambda beta: beta + A @ np.random.randn(p), lambda beta: beta + 0.1 * (1 + beta ** 2) * np.random.randn(p),
                  lambda beta: beta + 0.02 * rng.standard_normal(p), lambda beta: 0.999 * beta + 0.02 * np.random.randn(p),
                  lambda beta: beta + 0.02 * np.random.randn(p) * np.random.randn(p)):
        assert not isinstance(la.mhKernel(m.lpost, other), K.FusedKernel)
    # a custom dprop keeps the generic path too (the fused kernel is the symmetric random walk)
    assert not isinstance(la.mhKernel(m.lpost, rprop, lambda new, old: 0.0), K.FusedKernel)
    # how the proposal was recognised is on the kernel and in the run's info; fuse=False opts out of the recognition altogether
    assert k.proposal == "probed" and la.mhKernel(m.lpost, la.rwProposal(0.02 * pre)).proposal == "rwProposal"
    _, info = la.mcmc(map_beta, k, thin=2, iters=2, verb=False, seed=3, return_info=True)
    assert info["proposal"] == "probed"
    assert not isinstance(la.mhKernel(m.lpost, rprop, fuse=False), K.FusedKernel)
    # a proposal that draws through ANOTHER function of NumPy's global generator is not recognised -- and the probe leaves the caller's
    # seeded stream where it was (the generator's state is saved and restored around the probe)
    np.random.seed(11)
    before = np.random.get_state()[1].copy()
    assert not isinstance(la.mhKernel(m.lpost, lambda beta: beta + 0.02 * np.random.normal(size=p)), K.FusedKernel)
    assert np.array_equal(np.random.get_state()[1], before)


def test_c_level_exchange_on_the_test_double(la):
    """include/logreg_hip.h lr_comm_* / lr_gather / lr_allreduce_sum_f64 through the binding on the CPU test double: a world of one rank
    (identifier, communicator, the gather as a copy, the sum in place).  Several ranks: tests/test_distributed_gloo.py (processes)."""
    import ctypes as C
    from logreg_amd import _lib
    L = _lib.load()
    ident = (C.c_ubyte * 128)()
    _lib.check(L.lr_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(L.lr_comm_create(ident, 0, 1, 0, C.byref(comm)))
    src = np.arange(24, dtype=np.float32)
    dst = np.zeros(24, dtype=np.float32)
    _lib.check(L.lr_gather(comm, src.ctypes.data, dst.ctypes.data, src.nbytes, 0, None))
    assert np.array_equal(dst, src)
    sums = np.linspace(0, 1, 57)
    keep = sums.copy()
    _lib.check(L.lr_allreduce_sum_f64(comm, sums.ctypes.data, sums.size, None))
    assert np.array_equal(sums, keep)
    assert L.lr_gather(comm, src.ctypes.data, dst.ctypes.data, src.nbytes, 1, None) < 0
    _lib.check(L.lr_comm_destroy(comm))
    assert L.lr_comm_create(ident, 2, 2, 0, C.byref(comm)) < 0 and b"rank 2 of world 2" in L.lr_last_error()
