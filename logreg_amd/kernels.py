"""The reference's kernel constructors and chain driver, with the same names and signatures:

    mhKernel(lpost, rprop, dprop=...)            fit-numpy.py:53-62 / fit-np-mala.py:61-70 / fit-np-hmc.py:56-63
    malaKernel(lpi, glpi, dt=1e-4, pre=1)        fit-np-mala.py:72-78
    hmcKernel(lpi, glpi, eps=1e-4, l=10, dmm=1)  fit-np-hmc.py:65-87
    ulKernel(glpi, dt=1e-4, pre=1)               fit-np-ul.py:61-68
    mcmc(init, kernel, thin=10, iters=10000, verb=True)   fit-np-hmc.py:89-103

When the callables passed in are the closures of a `LogReg` model (and, for RWMH, a
`rwProposal`), the constructors return a `FusedKernel`: still callable one step at a time with
the reference's per-step signature, but `mcmc()` recognises it and runs the whole
`iters x thin` loop for all chains inside one HIP kernel launch per chunk.

Any other callables get the reference's generic composition semantics (the kernel simply calls
what it was given, drawing from NumPy's global RNG exactly where the reference does); that path
contains no model arithmetic of its own.

Extensions (additive): `init` may be `[C, p]` (then `mcmc` returns `[iters, C, p]`);
`mcmc(..., seed=, chunk=, group=, mode=, chain_offset=, return_info=)`.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import RunOpts, check
from .model import DeviceArray, LogReg, ModelFn


# ------------------------------------------------------------------------------------------------
def _default_dprop(new, old):  # fit-numpy.py:53 `dprop = lambda new, old: 1.`
    return 1.0


class RandomWalkProposal:
    """rprop(beta) = beta + sd * N(0, I)   (fit-numpy.py:81-84 with sd = 0.02*pre)."""

    def __init__(self, sd):
        self.sd = np.asarray(sd, dtype=np.float64)

    def __call__(self, beta):
        beta = np.asarray(beta, dtype=np.float64)
        return beta + self.sd * np.random.randn(*beta.shape)


def rwProposal(sd) -> RandomWalkProposal:
    return RandomWalkProposal(sd)


_PROBE_LOCK = __import__("threading").Lock()


def _model_of(fn, kind):
    return fn.model if isinstance(fn, ModelFn) and fn.kind == kind else None


def _recognise_random_walk(rprop, p):
    """The proposal scale `sd` if the Python callable `rprop` is the reference's random-walk proposal
    `beta + sd * np.random.randn(p)` (fit-numpy.py:81-84, `sd = 0.02*pre`), else None.

    The script's own `rprop` is an opaque function, so it is PROBED, with `np.random.randn` replaced by recorded stand-ins
    (NumPy's global generator is not touched): zeros must give the identity, unit vectors the diagonal scale, and two
    random draws at two random points must reproduce `x + sd * z` to the last bit or two; exactly one `randn` call of p
    values per proposal.  Anything else -- another generator, a dense or state-dependent scale, side effects that raise --
    is not recognised and the kernel stays on the generic path."""
    if isinstance(rprop, RandomWalkProposal):
        return rprop.sd
    if not callable(rprop):
        return None
    with _PROBE_LOCK:  # np.random.randn is process-wide state while patched: one probe at a time
        state = np.random.get_state()  # a proposal that draws through another np.random function must not move the caller's seeded stream
        try:
            return _probe_random_walk(rprop, p)
        finally:
            np.random.set_state(state)


def _probe_random_walk(rprop, p):
    saved = np.random.randn
    calls = []

    def probe(x, z):
        def fake(*shape):
            calls.append(shape)
            return np.array(z, dtype=np.float64).reshape(shape if shape else ())
        np.random.randn = fake
        try:
            out = rprop(np.array(x, dtype=np.float64))
        finally:
            np.random.randn = saved
        out = np.asarray(out, dtype=np.float64)
        if out.shape != (p,) or not np.all(np.isfinite(out)):
            raise ValueError("not a [p] -> [p] map")
        return out
    try:
        x0 = np.linspace(-1.0, 1.0, p) * 0.37 + 0.11
        if not np.array_equal(probe(x0, np.zeros(p)), x0):
            return None
        sd = np.empty(p)
        for j in range(p):
            d = probe(np.zeros(p), np.eye(p)[j])  # at the origin: 0 + sd_j * 1 is sd_j to the last bit
            if np.any(d[np.arange(p) != j] != 0.0):
                return None  # a dense proposal covariance
            sd[j] = d[j]
        if np.any(sd < 0) or len(calls) != p + 1 or any(int(np.prod(c)) != p for c in calls):
            return None
        rng = np.random.RandomState(12345)  # a private generator: the caller's seeding of the global one stays as it is
        for _ in range(2):
            x, z = 3.0 * rng.standard_normal(p), rng.standard_normal(p)
            if not np.allclose(probe(x, z), x + sd * z, rtol=1e-14, atol=1e-300):
                return None
    except Exception:
        return None
    return sd


# ------------------------------------------------------------------------------------------------
class FusedKernel:
    """A transition kernel whose whole step runs inside the HIP chain kernel.

    kind in {"rwmh", "mala", "hmc", "ul"}.  Calling it performs ONE iteration on the device with
    the reference's per-step signature: `kernel(x, ll) -> (x, ll)` for rwmh/mala (the threaded
    log-density, fit-np-mala.py:61-70), `kernel(x) -> x` for hmc/ul.  Successive calls use
    successive iteration indices of the kernel's own Philox stream.
    """

    def __init__(self, kind: str, model: LogReg, **params):
        self.kind = kind
        self.model = model
        self.params = params
        self.threaded = kind in ("rwmh", "mala")
        self._seed = None
        self._calls = 0

    def __repr__(self):
        return f"FusedKernel({self.kind}, {self.params}, {self.model!r})"

    def _vec(self, name):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(self.params[name], dtype=np.float64), (self.model.p,)))

    def launch(self, opts: RunOpts, state_ptr, lp_ptr, out_ptr, acc_ptr):
        L = self.model._L
        h = self.model.handle
        if self.kind == "rwmh":
            v = self._vec("prop_sd")
            check(L.lr_run_rwmh(h, state_ptr, lp_ptr, v.ctypes.data, C.byref(opts), out_ptr, acc_ptr))
        elif self.kind == "mala":
            v = self._vec("pre")
            check(L.lr_run_mala(h, state_ptr, lp_ptr, float(self.params["dt"]), v.ctypes.data, C.byref(opts), out_ptr,
                                acc_ptr))
        elif self.kind == "ul":
            v = self._vec("pre")
            check(L.lr_run_ul(h, state_ptr, float(self.params["dt"]), v.ctypes.data, C.byref(opts), out_ptr, acc_ptr))
        elif self.kind == "hmc":
            v = self._vec("dmm")
            check(L.lr_run_hmc(h, state_ptr, float(self.params["eps"]), int(self.params["l"]), v.ctypes.data,
                               C.byref(opts), out_ptr, acc_ptr))
        else:
            raise ValueError(self.kind)

    # one step, reference signature
    def __call__(self, x, ll=None):
        if self._seed is None:
            self._seed = int(np.random.randint(0, 2**31 - 1))
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        st = np.array(np.atleast_2d(x), dtype=self.model.np_dtype, order="C")  # a copy: the launch updates it in place, the caller's x stays
        Cn = st.shape[0]
        lp = np.ascontiguousarray(np.broadcast_to(np.asarray(-np.inf if ll is None else ll, dtype=np.float64), (Cn,))).copy()
        opts = RunOpts(n_chains=Cn, chain_offset=0, thin=1, iters=1, iter_offset=self._calls, seed=self._seed,
                       group=0, mode=_lib.MODE_AUTO, on_device=0)
        self.launch(opts, st.ctypes.data, lp.ctypes.data, None, None)
        self._calls += 1
        xo = st.astype(np.float64)
        if single:
            xo = xo[0]
        if self.threaded:
            return xo, (float(lp[0]) if single else lp)
        return xo


# ------------------------------------------------------------------------------------------------
def mhKernel(lpost, rprop, dprop=_default_dprop, *, fuse=True):
    """Metropolis-Hastings kernel constructor.

    `fuse=False` (an addition to the reference's signature) keeps the reference's generic composition whatever `rprop` is: the caller's
    own Python proposal is then called every step and draws from NumPy's generator, instead of being recognised and replaced by the
    device kernel's Philox draws.  A fused kernel says how its proposal was recognised in `kernel.proposal` ("rwProposal" | "probed"),
    which `mcmc(..., return_info=True)` reports as `info["proposal"]`.

    Fused when `lpost` is a LogReg's lpost and `rprop` is a random-walk proposal `beta + sd * N(0, I)` -- a
    `rwProposal(sd)`, or the script's own Python function `rprop` (fit-numpy.py:81-84), recognised by probing it
    (`_recognise_random_walk`) -- so that the reference's call `mhKernel(lpost, rprop)` runs fused unchanged
    (fit-numpy.py:53-62,86).  Otherwise the reference's generic composition: the returned
    `kernel(x, ll)` threads the current log-density (fit-numpy.py:54-61); called as `kernel(x)`
    it re-evaluates both ends like the HMC script's variant (fit-np-hmc.py:56-63)."""
    model = _model_of(lpost, "lpost")
    if fuse and model is not None and dprop is _default_dprop:
        sd = _recognise_random_walk(rprop, model.p)  # a rwProposal, or the script's own `rprop` recognised by probing it
        if sd is not None:
            k = FusedKernel("rwmh", model, prop_sd=sd)
            k.proposal = "rwProposal" if isinstance(rprop, RandomWalkProposal) else "probed"
            return k

    def kernel(x, ll=None):
        prop = rprop(x)
        lp = lpost(prop)
        if ll is None:  # fit-np-hmc.py:59
            a = lp - lpost(x)
        else:  # fit-numpy.py:57
            a = lp - ll + dprop(x, prop) - dprop(prop, x)
        if np.log(np.random.rand()) < a:
            x = prop
            ll = lp if ll is not None else None
        return x if ll is None else (x, ll)
    return kernel


def malaKernel(lpi, glpi, dt=1e-4, pre=1):
    """MALA with diagonal pre-conditioner (fit-np-mala.py:72-78)."""
    model = _model_of(lpi, "lpost")
    if model is not None and _model_of(glpi, "glp") is model:
        return FusedKernel("mala", model, dt=float(dt), pre=pre)
    sdt = np.sqrt(dt)
    spre = np.sqrt(pre)

    def advance(x):
        return x + 0.5 * pre * glpi(x) * dt

    def lognorm(x, loc, scale):
        z = (x - loc) / scale
        return np.sum(-0.5 * z * z - 0.5 * np.log(2 * np.pi) - np.log(scale))
    return mhKernel(lpi, lambda x: advance(x) + np.random.randn(*np.shape(x)) * spre * sdt,
                    lambda new, old: lognorm(new, advance(old), spre * sdt * np.ones(np.shape(new))))


def ulKernel(glpi, dt=1e-4, pre=1):
    """Unadjusted Langevin (fit-np-ul.py:61-68)."""
    model = _model_of(glpi, "glp")
    if model is not None:
        return FusedKernel("ul", model, dt=float(dt), pre=pre)
    sdt = np.sqrt(dt)
    spre = np.sqrt(pre)

    def kernel(x):
        return x + 0.5 * pre * glpi(x) * dt + np.random.randn(*np.shape(x)) * spre * sdt
    return kernel


def hmcKernel(lpi, glpi, eps=1e-4, l=10, dmm=1):
    """HMC with diagonal mass matrix (fit-np-hmc.py:65-87)."""
    model = _model_of(lpi, "lpost")
    if model is not None and _model_of(glpi, "glp") is model:
        return FusedKernel("hmc", model, eps=float(eps), l=int(l), dmm=dmm)
    sdmm = np.sqrt(dmm)

    def leapf(q, p):
        p = p + 0.5 * eps * glpi(q)
        for i in range(l):
            q = q + eps * p / dmm
            p = p + (eps if i < l - 1 else 0.5 * eps) * glpi(q)
        return (q, -p)

    def alpi(x):
        q, p = x
        return lpi(q) - 0.5 * np.sum((p ** 2) / dmm)
    mhk = mhKernel(alpi, lambda x: leapf(*x))

    def kern(q):
        p = np.random.randn(len(q)) * sdmm
        return mhk((q, p))[0]
    return kern


# ------------------------------------------------------------------------------------------------
class ChainSet:
    """C chains of one FusedKernel, resident on the device between launches.

    `advance(iters, thin)` enqueues one fused launch (iters*thin iterations per chain) and
    returns the thinned samples as a DeviceArray [iters, C, p] (or None with keep=False).
    Iteration counters advance automatically, so any chunking of a run gives identical output.
    """

    def __init__(self, kernel: FusedKernel, init, seed: int, chain_offset: int = 0, ll=None, group: int = 0,
                 mode: str = "auto", stream=None, precision: str = "auto", plan_chains: int = 0, plan_first: int = 0):
        self.kernel = kernel
        self.model = kernel.model
        m = self.model
        st = np.ascontiguousarray(np.atleast_2d(np.asarray(init, dtype=np.float64)), dtype=m.np_dtype)
        if st.ndim != 2 or st.shape[1] != m.p:
            raise ValueError(f"init must be [p] or [C,p] with p={m.p}; got {np.shape(init)}")
        self.C = st.shape[0]
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.chain_offset = int(chain_offset)
        self.iter_offset = 0
        self.group = int(group)
        self.mode = _lib.MODE_BY_NAME[mode]
        # arithmetic of HMC's interior leapfrog gradients (include/logreg_hip.h LR_PREC_*): "auto" | "full" | "bf16"
        self.precision = _lib.PREC_BY_NAME[precision]
        # chain count every chain-count-dependent choice is made for (0 = this set's own): a shard of a larger run passes the
        # whole run's count and reproduces the one-GPU run bit for bit (lr_run_opts.plan_chains)
        self.plan_chains = int(plan_chains)
        # ... and the global id of that run's first chain (a two-part plan assigns a chain to its part by its position in the run)
        self.plan_first = int(plan_first)
        self.stream = stream
        self.state = DeviceArray.from_host(m.device, st)
        lp0 = np.full(self.C, -np.inf) if ll is None else np.broadcast_to(np.asarray(ll, dtype=np.float64), (self.C,))
        self.lp = DeviceArray.from_host(m.device, lp0, dtype=np.float64)
        self.acc = DeviceArray(m.device, (self.C,), np.uint32)
        self.acc.zero_()
        check(m._L.lr_stream_sync(m.device, None))
        # streaming statistics (enable_stats): device buffer [slots, C, 2, p], batch length, kept samples folded in
        self.stats = None
        self.stats_batch = 0
        self.stats_kept = 0
        self.pivot = st[0].astype(np.float64)  # any point near the posterior; shards of one run must share it

    def plan(self):
        """The kernel variant `advance` launches for this chain set (family- and precision-aware: `lr_plan_run`)."""
        opts = RunOpts(n_chains=self.C, group=self.group, mode=self.mode, precision=self.precision, plan_chains=self.plan_chains,
                       chain_offset=self.chain_offset, plan_first=self.plan_first)
        info = _lib.PlanInfo()
        check(self.model._L.lr_plan_run_info(self.model.handle, _lib.KIND_BY_NAME[self.kernel.kind], C.byref(opts), C.byref(info)))
        plan = {"mode": _lib.MODE_NAMES[info.mode], "group": info.group, "rows_per_lane": info.rows}
        if info.split > 0:  # a run planned in two parts: chains [split, n) of the planned run on wider lane groups
            plan["tail"] = {"from": int(info.split), "mode": _lib.MODE_NAMES[info.tail_mode], "group": info.tail_group,
                            "rows_per_lane": info.tail_rows}
        return plan

    def enable_stats(self, batch: int, slots: int, pivot=None):
        """Start a statistics window: from now on every kept sample is folded, on the device, into the running
        (mean, M2) of its batch of `batch` kept samples (include/logreg_hip.h "Streaming statistics"); `slots`
        batches are provided for.  The window restarts at the current state."""
        m = self.model
        if self.stats is not None:
            self.stats.free()
        self.stats = DeviceArray(m.device, (int(slots), self.C, 2, m.p), np.float64)
        self.stats_batch = int(batch)
        self.stats_kept = 0
        if pivot is not None:
            self.pivot = np.asarray(pivot, dtype=np.float64).copy()

    def advance(self, iters: int, thin: int, keep: bool = True, out: DeviceArray | None = None, stats: bool | None = None):
        """One fused launch: iters*thin iterations per chain.  keep: return the thinned samples `[iters, C, p]`
        (a DeviceArray); stats (default: on when enable_stats was called): fold the kept samples into the
        streaming statistics -- with keep=False nothing of size iters*C*p is ever allocated."""
        m = self.model
        if keep and out is None:
            out = DeviceArray(m.device, (iters, self.C, m.p), m.np_dtype)
        opts = RunOpts(n_chains=self.C, chain_offset=self.chain_offset, thin=int(thin), iters=int(iters),
                       iter_offset=self.iter_offset, seed=self.seed, group=self.group, mode=self.mode, on_device=1,
                       stream=self.stream, precision=self.precision, plan_chains=self.plan_chains, plan_first=self.plan_first)
        use_stats = self.stats is not None if stats is None else bool(stats)
        if use_stats:
            if self.stats is None:
                raise ValueError("advance(stats=True) needs enable_stats(batch, slots) first")
            opts.stats, opts.stats_batch = self.stats.ptr, self.stats_batch
            opts.stats_first, opts.stats_slots = self.stats_kept, self.stats.shape[0]
        self.kernel.launch(opts, self.state.ptr, self.lp.ptr, out.ptr if keep else None, self.acc.ptr)
        self.iter_offset += int(iters) * int(thin)
        if use_stats:
            self.stats_kept += int(iters)
        return out if keep else None

    def stats_sums(self) -> np.ndarray:
        """Chain-pooled sums `[7, p]` of the statistics window (`lr_stats_reduce`: a device reduction over the
        chains; only 7p doubles cross PCIe).  Sums of chain shards add: see distributed.reduce_stats."""
        if self.stats is None:
            raise ValueError("no statistics window: call enable_stats first")
        m = self.model
        piv = np.ascontiguousarray(self.pivot, dtype=np.float64)
        sums = np.empty((_lib.STATS_ROWS, m.p), dtype=np.float64)
        check(m._L.lr_stats_reduce(m.device, self.stats.ptr, self.C, m.p, self.stats_batch, self.stats_kept,
                                          piv.ctypes.data, sums.ctypes.data, self.stream))
        return sums

    def stats_summary(self) -> dict:
        """mean / sd / split-R-hat / batch-means ESS / MCSE of the window, from the device accumulators."""
        from .diagnostics import summary_from_sums
        return summary_from_sums(self.stats_sums(), self.C, self.stats_kept, self.stats_batch, self.pivot)

    def sync(self):
        check(self.model._L.lr_stream_sync(self.model.device, self.stream))

    def get_state(self):
        self.sync()
        return self.state.to_host().astype(np.float64)

    def get_ll(self):
        self.sync()
        return self.lp.to_host()

    def get_accepts(self):
        self.sync()
        return self.acc.to_host()

    # -- checkpoint / resume (the reference has none: a run is all-or-nothing).  Because the random
    # stream is counter-based, (state, threaded ll, iteration counter, seed, chain offset) IS the
    # complete sampler state: a resumed run continues bit-for-bit.
    def _fingerprint(self) -> dict:
        """What a resumed run must share with the saved one for the continuation to be bit-exact: model shape and
        arithmetic type, a hash of the data block, and the kernel's parameters."""
        m, k = self.model, self.kernel
        par = {name: np.broadcast_to(np.asarray(v, dtype=np.float64), (m.p,) if np.ndim(v) else ()).copy()
               for name, v in sorted(k.params.items())}
        return {"kind": k.kind, "n": np.int64(m.n), "p": np.int64(m.p), "dtype": np.dtype(m.np_dtype).name,
                "data_hash": m.data_hash, **{"param_" + name: v for name, v in par.items()}}

    def checkpoint(self) -> dict:
        self.sync()
        ck = {"state": self.state.to_host(), "ll": self.lp.to_host(), "accepts": self.acc.to_host(),
              "iter_offset": np.int64(self.iter_offset), "seed": np.uint64(self.seed),
              "chain_offset": np.int64(self.chain_offset), "plan_chains": np.int64(self.plan_chains),
              "plan_first": np.int64(self.plan_first), **self._fingerprint()}
        if self.stats is not None:  # the statistics window travels with the run
            ck.update(stats=self.stats.to_host(), stats_batch=np.int64(self.stats_batch), stats_kept=np.int64(self.stats_kept),
                      stats_pivot=np.asarray(self.pivot, dtype=np.float64))
        return ck

    def save(self, path: str) -> str:
        """Write the checkpoint as an .npz archive; returns the path actually written (np.savez appends
        ".npz" to a name without it), so `ChainSet.resume(kernel, cs.save("ckpt"))` works."""
        path = str(path)
        if not path.endswith(".npz"):
            path += ".npz"
        np.savez(path, **self.checkpoint())
        return path

    @classmethod
    def resume(cls, kernel: "FusedKernel", ckpt, group: int = 0, mode: str = "auto", stream=None,
               precision: str = "auto") -> "ChainSet":
        if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
            path = str(ckpt)
            if not path.endswith(".npz"):
                path += ".npz"
            ckpt = dict(np.load(path, allow_pickle=False))
        if str(ckpt["kind"]) != kernel.kind:
            raise ValueError(f"checkpoint is for a {ckpt['kind']} kernel, got {kernel.kind}")
        # (a shard resumes as the shard of the same planned run: its chains keep the kernel variants they had)
        cs = cls(kernel, ckpt["state"], int(ckpt["seed"]), chain_offset=int(ckpt["chain_offset"]), ll=ckpt["ll"],
                 group=group, mode=mode, stream=stream, precision=precision,
                 plan_chains=int(ckpt["plan_chains"]) if "plan_chains" in ckpt else 0,
                 plan_first=int(ckpt["plan_first"]) if "plan_first" in ckpt else 0)
        want = cs._fingerprint()
        for key, val in want.items():
            if key not in ckpt:
                continue  # checkpoints written before the fingerprint existed carry only `kind`
            have = ckpt[key]
            same = np.array_equal(np.asarray(have), np.asarray(val)) if key.startswith("param_") else str(have) == str(val)
            if not same:
                raise ValueError(f"checkpoint does not match this kernel/model: {key} = {have!r} in the checkpoint, "
                                 f"{val!r} here (a resumed run would silently stop being the continuation)")
        cs.iter_offset = int(ckpt["iter_offset"])
        cs.acc.copy_from(np.asarray(ckpt["accepts"], dtype=np.uint32))
        if "stats" in ckpt:
            st = np.asarray(ckpt["stats"], dtype=np.float64)
            cs.enable_stats(int(ckpt["stats_batch"]), st.shape[0], pivot=ckpt["stats_pivot"])
            cs.stats.copy_from(st)
            cs.stats_kept = int(ckpt["stats_kept"])
        return cs


def _auto_chunk(kernel: FusedKernel, C: int, thin: int, iters: int) -> int:
    """Kept samples per launch.  A launch is bounded to ~2e9 data-element visits per chain (a few
    hundred ms at Pima scale): short enough to report progress and to stay far from any watchdog,
    long enough that launch overhead is invisible.  Chunking never changes the samples."""
    evals = {"hmc": kernel.params.get("l", 1), "mala": 1, "ul": 1, "rwmh": 1}[kernel.kind]
    work_per_kept = max(1, thin * evals * kernel.model.n * kernel.model.p)
    return int(max(1, min(iters, 2_000_000_000 // work_per_kept)))


def mcmc(init, kernel, thin=10, iters=10000, verb=True, *, seed=None, chunk=None, chain_offset=0, ll=None,
         group=0, mode="auto", return_info=False, summary_only=False, max_batches=16, precision="auto", plan_chains=0, plan_first=0):
    """Run a chain (or C chains): `mat[i]` = state after (i+1)*thin iterations (fit-np-hmc.py:89-103).

    Fused kernels run on the device; `init` of shape [p] returns a float64 `[iters, p]` matrix
    like the reference, `[C, p]` returns `[iters, C, p]` in the model's dtype.  RWMH/MALA start
    with the threaded log-density at -inf as the reference does (fit-np-mala.py:82) unless
    `ll=` is given.  `seed=None` draws the Philox key from NumPy's global RNG, so
    `np.random.seed(s)` before the call makes a run reproducible, like the reference.

    `precision="auto"` (the DEFAULT) lets HMC's l - 1 gradient evaluations strictly inside a trajectory run in reduced precision where such
    a kernel exists -- on a float64 model too (float64 state, end points and Metropolis test; float32 / 16-bit force inside the
    trajectory, at any chain count): such a run is NOT step-for-step comparable with the reference; `precision="full"` is (every
    evaluation in the model's dtype).  INTEGRATION.md section 3b has the numbers.

    `summary_only=True` (fused kernels): no sample matrix at all -- the kept samples are folded into on-device
    running statistics and the call returns a dict (mean, sd, rhat, ess, mcse, accept_rate, ...): what the
    reference computes from the full matrix afterwards (fit-np-hmc.py:113-117, analyse.R:17-19).

    `precision` (HMC): "auto" (DEFAULT) lets the L - 1 interior leapfrog gradients of a trajectory run on the matrix
    pipe from 16-bit operands (bf16 pieces; IEEE half precision for wide models whose design fits it) where such a kernel exists (the end-point value + gradient and the Metropolis test stay in the model's dtype, so
    the sampler stays exact; the acceptance rate is the only thing that can move); "full" keeps every evaluation in the
    model's dtype (step-for-step comparable with the float64 reference); see include/logreg_hip.h LR_PREC_*.  float64 models
    follow the same policy with more kept exact: position, momentum, end points, kinetic energies and the Metropolis test are
    float64, only the force inside the trajectory comes from float32 / 16-bit operands (4 - 6 x the all-float64 rate).
    `plan_chains`, `plan_first`: chain count to plan the kernel variant for and the global id of that run's first chain (a shard of
    a larger run passes the whole run's: its chains then run on the variants they have in the whole run, bit for bit).
    """
    if not isinstance(kernel, FusedKernel):
        return _mcmc_generic(init, kernel, thin, iters, verb)
    init = np.asarray(init, dtype=np.float64)
    single = init.ndim == 1
    if seed is None:
        seed = int(np.random.randint(0, 2**31 - 1))
    cs = ChainSet(kernel, init, seed, chain_offset=chain_offset, ll=ll, group=group, mode=mode, precision=precision,
                  plan_chains=plan_chains, plan_first=plan_first)
    m = kernel.model
    if chunk is None:
        chunk = _auto_chunk(kernel, cs.C, thin, iters)
    if summary_only:
        from .diagnostics import choose_batches
        batch, slots = choose_batches(iters, max_batches)
        cs.enable_stats(batch, slots)
        if verb:
            print(str(iters) + " iterations")
        done = 0
        while done < iters:
            k = min(chunk, iters - done)
            cs.advance(k, thin, keep=False)
            cs.sync()
            done += k
            if verb:
                print(str(done), end=" ", flush=True)
        if verb:
            print("\nDone.", flush=True)
        res = cs.stats_summary()
        res.update(accept_rate=float(cs.get_accepts().sum() / (cs.C * iters * thin)), batch=batch, seed=seed,
                   plan=cs.plan(), state=cs.get_state())
        return res
    mat = np.empty((iters, cs.C, m.p), dtype=m.np_dtype)
    if verb:
        print(str(iters) + " iterations")
    done = 0
    while done < iters:
        k = min(chunk, iters - done)
        out = cs.advance(k, thin, keep=True)
        cs.sync()
        mat[done:done + k] = out.to_host()
        out.free()
        done += k
        if verb:
            print(str(done), end=" ", flush=True)
    if verb:
        print("\nDone.", flush=True)
    res = mat[:, 0, :].astype(np.float64) if single else mat
    if return_info:
        info = {"accepts": cs.get_accepts(), "state": cs.get_state(), "ll": cs.get_ll(), "seed": seed,
                "plan": cs.plan(), "iterations": iters * thin}
        if getattr(kernel, "proposal", None):  # RWMH: how the proposal was recognised ("rwProposal" | "probed": mhKernel)
            info["proposal"] = kernel.proposal
        return res, info
    return res


def _mcmc_generic(init, kernel, thin, iters, verb):
    """The reference's driver for arbitrary Python kernels.  Kernels built by the generic
    `mhKernel` / `malaKernel` thread `ll` (fit-np-mala.py:80-95); `hmcKernel` / `ulKernel`
    kernels take and return the state only (fit-np-hmc.py:89-103)."""
    import inspect
    p = len(init)
    mat = np.zeros((iters, p))
    x = init
    try:
        nparams = len(inspect.signature(kernel).parameters)
    except (TypeError, ValueError):
        nparams = 1
    threaded = nparams >= 2
    ll = -np.inf
    if verb:
        print(str(iters) + " iterations")
    for i in range(iters):
        if verb:
            print(str(i), end=" ", flush=True)
        for j in range(thin):
            if threaded:
                x, ll = kernel(x, ll)
            else:
                x = kernel(x)
        mat[i, :] = x
    if verb:
        print("\nDone.", flush=True)
    return mat
