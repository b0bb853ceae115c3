#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's NumPy scripts.

Runs only in the build container (needs /root/reference, read-only). Nothing here ships
reference source: the scripts are read at run time, `exec`-ed in a scratch namespace, and only
numeric inputs/outputs are written out as JSON / NPZ.  The GPU box never runs this file.

Import recipe: the reference scripts are not modules (hyphenated names, the
whole 10^4 x thin run executes at import, no __main__ guard).  We read the source text, cut it
at the first "out = mcmc(", seed NumPy's global RNG, and exec the head with stdout silenced and
the cwd set so that "../pima.parquet" resolves.

    python tests/golden/make_fixtures.py small            # F1-F6, F8b, F9, data  (~1 min)
    python tests/golden/make_fixtures.py posterior-hmc  --seeds 42 43 44 45   (~2.5 min each, parallel)
    python tests/golden/make_fixtures.py posterior-mala --seeds 42            (~35 min)
    python tests/golden/make_fixtures.py posterior-rwmh --seeds 42            (~26 min)
    python tests/golden/make_fixtures.py posterior-ul   --seeds 42 43         (~8 min each, parallel; fit-np-ul.py:88)

Fixture ids follow SURVEY.md section 8(c).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LOGREG_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

SCRIPTS = {
    "hmc": "fit-np-hmc.py",
    "mala": "fit-np-mala.py",
    "rwmh": "fit-numpy.py",
    "ul": "fit-np-ul.py",
}


def load_reference(kind: str, seed: int = 0) -> dict:
    """exec the head of a reference script (everything before the production run)."""
    path = os.path.join(REF, "Python", SCRIPTS[kind])
    src = open(path).read()
    head = src[: src.index("out = mcmc(")]
    ns: dict = {"__name__": "reference_" + kind}
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "Python"))
    try:
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(head, path, "exec"), ns)
    finally:
        os.chdir(cwd)
    return ns


def closure_of(fn, name):
    return fn.__closure__[fn.__code__.co_freevars.index(name)].cell_contents


def jdump(name, obj):
    def conv(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, (np.integer,)):
            return int(o)
        raise TypeError(type(o))
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, default=conv, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


# ----------------------------------------------------------------------------------------------
class DrawRecorder:
    """Wrap np.random.randn / np.random.rand to log (or replay) every draw."""

    def __init__(self):
        self.normals = []
        self.uniforms = []
        self._randn = np.random.randn
        self._rand = np.random.rand

    def __enter__(self):
        def randn(*a):
            z = self._randn(*a)
            self.normals.append(np.array(z, dtype=np.float64).copy())
            return z

        def rand(*a):
            u = self._rand(*a)
            self.uniforms.append(float(u))
            return u
        np.random.randn = randn
        np.random.rand = rand
        return self

    def __exit__(self, *exc):
        np.random.randn = self._randn
        np.random.rand = self._rand


def make_small():
    hmc = load_reference("hmc", 1)
    mala = load_reference("mala", 2)
    rw = load_reference("rwmh", 3)
    ul = load_reference("ul", 4)

    # ---- data (public MASS::Pima.tr, via the reference's pima.parquet) -------------------------
    X, y = hmc["X"], hmc["y"]
    n, p = X.shape
    assert (n, p) == (200, 8)
    for other in (mala, rw, ul):
        assert np.array_equal(other["X"], X) and np.array_equal(other["y"], y)
    jdump("pima_xy.json", {"source": "reference pima.parquet loaded by Python/fit-np-hmc.py:12-19",
                            "n": n, "p": p, "X": X, "y": y.astype(np.float64)})

    # ---- F2: MAP -------------------------------------------------------------------------------
    mapx = hmc["res"].x
    jdump("map.json", {
        "source": "scipy BFGS in Python/fit-np-hmc.py:49 (seed 1)",
        "map": mapx, "lpost_map": hmc["lpost"](mapx), "ll_map": hmc["ll"](mapx),
        "glp_map": hmc["glp"](mapx),
        "map_mala_script": mala["res"].x, "map_rwmh_script": rw["res"].x,
        "pscale": hmc["pscale"],
    })

    # ---- F1: model evaluations -----------------------------------------------------------------
    rng = np.random.RandomState(20240101)
    post_sd = np.array([1.71, 0.0655, 0.0068, 0.0184, 0.0226, 0.0429, 0.547, 0.0225])
    betas = [mapx.copy(), np.array([-9.0] + [0.0] * 7),
             np.array([-9.0, 0.1, 0.03, -0.01, 0.0, 0.08, 1.5, 0.03]), np.zeros(8)]
    for k in range(64):
        scale = 1.0 if k < 32 else 5.0
        betas.append(mapx + scale * post_sd * rng.randn(8))
    # extreme points: exercise the overflow behaviour of the naive log(1+exp(.)) form
    betas.append(np.array([50.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]))
    betas.append(-np.array([50.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]))
    betas = np.array(betas)
    rows = {"ll": [], "lprior": [], "lpost": [], "glp": []}
    with np.errstate(all="ignore"):
        for b in betas:
            for nm in rows:
                rows[nm].append(hmc[nm](b))
            # all scripts state the same model: check bit-agreement across them
            assert mala["ll"](b) == hmc["ll"](b) or np.isinf(hmc["ll"](b))
            assert np.array_equal(mala["glp"](b), hmc["glp"](b))
            lp_rw = rw["lprior"](b)
            assert abs(lp_rw - hmc["lprior"](b)) <= 1e-12 * abs(lp_rw)
    jdump("model_eval.json", {
        "source": "ll/lprior/lpost/glp closures of Python/fit-np-hmc.py:23-24,33-34,36-37,44-47",
        "beta": betas, **{k: np.array(v) for k, v in rows.items()}})

    # ---- F3: leapfrog + alpi -------------------------------------------------------------------
    pre_h = hmc["pre"]
    dmm = 1.0 / pre_h
    kern = hmc["hmcKernel"](hmc["lpost"], hmc["glp"], eps=1e-3, l=50, dmm=dmm)
    mhk = closure_of(kern, "mhk")
    alpi = closure_of(mhk, "lpost")
    rprop_h = closure_of(mhk, "rprop")
    leapf = closure_of(rprop_h, "leapf")
    q0s, p0s, q1s, p1s, a0s, a1s = [], [], [], [], [], []
    cases = [(mapx.copy(), np.array([0.1, -1, 0.5, 0.3, -0.2, 1.1, 0.05, -0.7]))]
    for k in range(31):
        cases.append((mapx + post_sd * rng.randn(8), rng.randn(8) * np.sqrt(dmm)))
    for q0, p0 in cases:
        q1, p1 = leapf(q0, p0)
        q0s.append(q0); p0s.append(p0); q1s.append(q1); p1s.append(p1)
        a0s.append(alpi((q0, p0))); a1s.append(alpi((q1, p1)))
    # a second parameter set (short trajectory, unit mass) to pin l/eps/dmm handling
    kern2 = hmc["hmcKernel"](hmc["lpost"], hmc["glp"], eps=2e-4, l=3, dmm=1)
    leapf2 = closure_of(closure_of(closure_of(kern2, "mhk"), "rprop"), "leapf")
    q1b, p1b = leapf2(cases[0][0], cases[0][1])
    jdump("leapfrog.json", {
        "source": "leapf/alpi closure cells of hmcKernel, Python/fit-np-hmc.py:65-87",
        "eps": 1e-3, "l": 50, "dmm": dmm,
        "q0": np.array(q0s), "p0": np.array(p0s), "q1": np.array(q1s), "p1_negated": np.array(p1s),
        "alpi0": np.array(a0s), "alpi1": np.array(a1s),
        "short": {"eps": 2e-4, "l": 3, "dmm": np.ones(8), "q0": cases[0][0], "p0": cases[0][1],
                  "q1": q1b, "p1_negated": p1b},
    })

    # ---- F4: MALA terms ------------------------------------------------------------------------
    pre_m = mala["pre"]
    dt = 1e-5
    mk = mala["malaKernel"](mala["lpost"], mala["glp"], dt=dt, pre=pre_m)
    rprop_m = closure_of(mk, "rprop")
    dprop_m = closure_of(mk, "dprop")
    advance = closure_of(dprop_m, "advance")
    recs = {"x": [], "z": [], "advance_x": [], "prop": [], "dprop_x_prop": [], "dprop_prop_x": [],
            "lpost_x": [], "lpost_prop": [], "a": []}
    for k in range(32):
        x = mapx + (0.5 if k else 0.0) * post_sd * rng.randn(8)
        z = rng.randn(8)
        saved = np.random.randn
        np.random.randn = lambda *a, _z=z: _z.copy()
        try:
            prop = rprop_m(x)
        finally:
            np.random.randn = saved
        d1, d2 = dprop_m(x, prop), dprop_m(prop, x)
        lx, lp = mala["lpost"](x), mala["lpost"](prop)
        recs["x"].append(x); recs["z"].append(z); recs["advance_x"].append(advance(x))
        recs["prop"].append(prop); recs["dprop_x_prop"].append(d1); recs["dprop_prop_x"].append(d2)
        recs["lpost_x"].append(lx); recs["lpost_prop"].append(lp); recs["a"].append(lp - lx + d1 - d2)
    jdump("mala_terms.json", {
        "source": "malaKernel internals, Python/fit-np-mala.py:61-78 (a = lp - ll + dprop(x,prop) - dprop(prop,x) with ll=lpost(x))",
        "dt": dt, "pre": pre_m, **{k: np.array(v) for k, v in recs.items()}})

    # ---- F5: RWMH terms ------------------------------------------------------------------------
    pre_r = rw["pre"]
    recs = {"x": [], "z": [], "prop": [], "lpost_x": [], "lpost_prop": [], "a": []}
    for k in range(32):
        x = mapx + (0.5 if k else 0.0) * post_sd * rng.randn(8)
        z = rng.randn(8)
        saved = np.random.randn
        np.random.randn = lambda *a, _z=z: _z.copy()
        try:
            prop = rw["rprop"](x)
        finally:
            np.random.randn = saved
        lx, lp = rw["lpost"](x), rw["lpost"](prop)
        recs["x"].append(x); recs["z"].append(z); recs["prop"].append(prop)
        recs["lpost_x"].append(lx); recs["lpost_prop"].append(lp); recs["a"].append(lp - lx + 1.0 - 1.0)
    jdump("rwmh_terms.json", {
        "source": "rprop + mhKernel, Python/fit-numpy.py:53-62,81-84",
        "prop_sd": 0.02 * pre_r, **{k: np.array(v) for k, v in recs.items()}})

    # ---- F6: accept/reject replay (control flow of mcmc + kernels) -----------------------------
    replay = {}
    steps = {"hmc": 96, "mala": 512, "rwmh": 1024, "ul": 64}
    for kind, ns in (("hmc", hmc), ("mala", mala), ("rwmh", rw), ("ul", ul)):
        if kind == "hmc":
            k = ns["hmcKernel"](ns["lpost"], ns["glp"], eps=1e-3, l=50, dmm=1.0 / ns["pre"])
        elif kind == "mala":
            k = ns["malaKernel"](ns["lpost"], ns["glp"], dt=1e-5, pre=ns["pre"])
        elif kind == "rwmh":
            k = ns["mhKernel"](ns["lpost"], ns["rprop"])
        else:
            k = ns["ulKernel"](ns["glp"], dt=1e-6, pre=ns["pre"])
        init = ns["res"].x.copy()
        np.random.seed(1000 + len(kind))
        with DrawRecorder() as rec:
            out = ns["mcmc"](init, k, thin=1, iters=steps[kind], verb=False)
        # thin>1 indexing check: row i of a thin=4 run == row 4(i+1)-1 of the thin=1 run
        np.random.seed(1000 + len(kind))
        out4 = ns["mcmc"](init, k, thin=4, iters=steps[kind] // 4, verb=False)
        assert np.array_equal(out4, out[3::4])
        replay[kind] = {"init": init, "normals": np.array(rec.normals),
                        "uniforms": np.array(rec.uniforms), "states": out, "thin4_rows_equal_every_4th": True}
        moved = np.any(np.diff(np.vstack([init[None], out]), axis=0) != 0, axis=1)
        print(kind, "steps", steps[kind], "accepted", int(moved.sum()))
    jdump("accept_replay.json", {
        "source": "mcmc(...) of each script with np.random.randn/rand wrapped to log every draw; thin=1",
        "params": {"hmc": {"eps": 1e-3, "l": 50, "dmm": 1.0 / hmc["pre"]},
                   "mala": {"dt": 1e-5, "pre": mala["pre"]},
                   "rwmh": {"prop_sd": 0.02 * rw["pre"]},
                   "ul": {"dt": 1e-6, "pre": ul["pre"]}},
        **replay})

    # ---- F8b: acceptance rates -----------------------------------------------------------------
    rates = {}
    for kind, ns, nstep in (("hmc", hmc, 4000), ("mala", mala, 40000), ("rwmh", rw, 40000)):
        if kind == "hmc":
            k = ns["hmcKernel"](ns["lpost"], ns["glp"], eps=1e-3, l=50, dmm=1.0 / ns["pre"])
        elif kind == "mala":
            k = ns["malaKernel"](ns["lpost"], ns["glp"], dt=1e-5, pre=ns["pre"])
        else:
            k = ns["mhKernel"](ns["lpost"], ns["rprop"])
        np.random.seed(77)
        x = ns["res"].x.copy()
        llv = -np.inf
        acc = 0
        for i in range(nstep):
            if kind == "hmc":
                xn = k(x)
            else:
                xn, llv = k(x, llv)
            acc += int(np.any(xn != x))
            x = xn
        rates[kind] = {"steps": nstep, "accepted": acc, "rate": acc / nstep}
        print(kind, rates[kind])
    jdump("accept_rates.json", {"source": "kernels at run settings from the MAP, np.random.seed(77)", **rates})

    # ---- F9: Philox4x32-10 known answers (Random123 kat_vectors; not from the reference) -------
    jdump("philox_kat.json", {
        "source": "Random123 known-answer vectors for philox4x32-10 (the reference has no RNG contract)",
        "kat": [
            {"ctr": [0, 0, 0, 0], "key": [0, 0],
             "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
            {"ctr": [0xffffffff] * 4, "key": [0xffffffff, 0xffffffff],
             "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
            {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
             "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]},
        ]})


# ----------------------------------------------------------------------------------------------
def run_full(kind: str, seed: int) -> dict:
    """Run the FULL, unmodified reference script after np.random.seed(seed) in a scratch dir."""
    from logreg_amd.diagnostics import ess_per_param
    script = SCRIPTS[kind]
    tmp = tempfile.mkdtemp(prefix=f"ref_{kind}_{seed}_")
    os.makedirs(os.path.join(tmp, "Python"))
    os.symlink(os.path.join(REF, "pima.parquet"), os.path.join(tmp, "pima.parquet"))
    runner = (
        "import numpy as np, sys\n"
        f"src = open({os.path.join(REF, 'Python', script)!r}).read()\n"
        f"np.random.seed({seed})\n"
        f"exec(compile(src, {script!r}, 'exec'), {{'__name__': '__main__'}})\n"
    )
    t0 = time.time()
    with open(os.path.join(tmp, "stdout.txt"), "w") as so:
        subprocess.run([sys.executable, "-c", runner], cwd=os.path.join(tmp, "Python"), stdout=so,
                       stderr=subprocess.STDOUT, check=True)
    wall = time.time() - t0
    import pandas as pd
    pq = [f for f in os.listdir(os.path.join(tmp, "Python")) if f.endswith(".parquet")]
    assert len(pq) == 1, pq
    out = pd.read_parquet(os.path.join(tmp, "Python", pq[0])).to_numpy()
    ess = ess_per_param(out)
    sd = out.std(axis=0, ddof=1)
    return {"seed": seed, "wall_s": wall, "iters": out.shape[0], "mean": out.mean(axis=0), "sd": sd,
            "ess": ess, "mcse": sd / np.sqrt(ess), "first_row": out[0], "last_row": out[-1]}


def make_posterior(kind: str, seeds):
    import multiprocessing as mp
    with mp.Pool(min(len(seeds), 4)) as pool:
        runs = pool.starmap(run_full, [(kind, s) for s in seeds])
    mean = np.mean([r["mean"] for r in runs], axis=0)
    # pooled: independent runs -> variances of the means add
    mcse = np.sqrt(np.sum([r["mcse"] ** 2 for r in runs], axis=0)) / len(runs)
    sd = np.sqrt(np.mean([r["sd"] ** 2 for r in runs], axis=0))
    ess = np.sum([r["ess"] for r in runs], axis=0)
    thin = {"hmc": 20, "mala": 1000, "rwmh": 1000, "ul": 2000}[kind]
    jdump(f"posterior_{kind}.json", {
        "source": f"full unmodified Python/{SCRIPTS[kind]} after np.random.seed(seed); 10000 kept x thin {thin}",
        "runs": runs, "pooled": {"mean": mean, "sd": sd, "ess": ess, "mcse": mcse,
                                 "se_sd": sd / np.sqrt(2.0 * ess)}})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["small", "posterior-hmc", "posterior-mala", "posterior-rwmh", "posterior-ul"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[42])
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    if a.what == "small":
        make_small()
    else:
        make_posterior(a.what.split("-")[1], a.seeds)
