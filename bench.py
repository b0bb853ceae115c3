#!/usr/bin/env python3
"""bench.py -- headline benchmark: many-chain HMC for Bayesian logistic regression on MI355X.

Workload (BASELINE.json configs[1]): HMC, L = 50 leapfrog steps, 4096 chains per GPU, n = 200,
p = 8, synthetic design (X[:,0]=1, X[:,1:]~N(0,1), y~Bernoulli(sigma(X beta*)), seed 20240001),
unit mass matrix, eps = 0.1 (acceptance ~0.92 on this design), fp32 arithmetic.

A "step" is one fused launch that advances every chain by `thin` = 20 HMC iterations and writes
one kept sample per chain (i.e. one row block of the reference's mcmc() output).  Model data and
chain states are resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Multi-GPU: chains are sharded (weak scaling: 4096 per GPU, the
Philox counter carries the global chain id) and the kept samples are gathered to rank 0 over
RCCL inside the timed region.

At N = 1 the line also carries, outside the timed region:
  cpu_baseline   the float64 C oracle (OpenMP over chains) on a bounded sample of the same workload ("port")
  reference_cpu  the reference's own NumPy script, as measured in BASELINE.md (1 core)
  ess            ESS per kept draw from a separate 512-draw run (Geyer), scaled to the timed throughput
  extra.configs  BASELINE.json configs 1, 3, 4, 5 on this GPU (one GPU's shard where the config is multi-GPU), each with its
                 own roofline block, and "2_pima": the reference's own HMC run (Pima, eps=1e-3,
                 dmm=1/pre) at 4096 chains with like-for-like it/s and min-ESS/s
  extra.f64      the headline workload on the float64 instantiation (the reference computes in float64)
  extra.f64_wide config 5's shape on a float64 model (f64 matrix pipe)

`--dry-run` exercises the launch plumbing without a GPU (gloo instead of RCCL, no kernels): rank/world parsing,
shard offsets, the gather and the max-over-ranks reduction -- what tests/test_host_logic.py runs on CPU.
"""
from __future__ import annotations

import argparse
import ctypes as Ct
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_ROWS, N_PAR, LEAP, EPS, THIN = 200, 8, 50, 0.1, 20
CHAINS_PER_GPU = 4096
PREWARM_S = 0.15  # seconds of untimed load before the warm-up steps (GPU clock ramp), see main()
SEED = 42
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector == FP32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
MFMA16_PIPE = "16-bit MFMA (v_mfma_f32_16x16x32_f16 where the design fits f16 -- these synthetic designs do --, else _bf16), fp32 accumulate"
INTERIOR_NOTE = ("auto (the L-1 interior gradients on the matrix pipe from rows, beta and sigmoid weights in one f16 piece each -- bf16 rows x two "
                 "bf16 pieces of beta for a design outside the f16 range; end points exact)")
HBM_PEAK_GBS = 8000.0
# the reference's own script timed in BASELINE.md section 2 (Python/fit-np-hmc.py, Pima n=200 p=8, eps=1e-3 L=50)
REFERENCE_CPU = {"it_per_s": 1368.0, "grad_evals_per_s": 6.98e4, "min_ess_per_s": 24.8, "cores": 1,
                 "what": "unmodified Python/fit-np-hmc.py (NumPy), 1 chain, 1 core Xeon SPR 2.1 GHz, 146.2 s for 200 000 "
                         "HMC iterations: BASELINE.md section 2"}


def flops_per_grad_eval(n, p):  # F_g = 4np + 5n + 2p
    return 4 * n * p + 5 * n + 2 * p


def usable_cores() -> int:
    """CPU cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a container can see 256 logical CPUs and be allowed a handful)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(X, y, pscale, init, target_s=6.0):
    """Time the CPU oracle on bounded samples of the same workload, as SURVEY.md section 8(d)(ii) asks: the float64 C
    restatement (OpenMP over chains) at ALL usable cores and at ONE core, and the same C source compiled in IEEE float32
    (oracle/Makefile `f32`) likewise -- four legs of ~`target_s`/2..`target_s` seconds each (~20 s in all).  `value` is the
    float64 all-cores leg (the reference's arithmetic).  The oracle is the checker, used here only as the reported CPU
    baseline; every leg is `kind: "port"`."""
    from oracle import oracle as orc
    m = orc.OracleModel(X, y, pscale)
    # the thread count is passed explicitly (num_threads clause): torchrun exports OMP_NUM_THREADS=1 to its ranks
    all_threads = usable_cores()
    scale = np.ones(N_PAR)

    def leg(real, threads, budget):
        chains = 8 * threads
        st = np.tile(init, (chains, 1))

        def go(iters):
            t0 = time.perf_counter()
            if real == "f64":
                m.run("hmc", st, step=EPS, l=LEAP, scale=scale, thin=1, iters=iters, seed=SEED, keep=False, threads=threads)
            else:
                orc.run_f32(X, y, pscale, "hmc", st, step=EPS, l=LEAP, scale=scale, thin=1, iters=iters, seed=SEED, threads=threads)
            return time.perf_counter() - t0
        go(2)  # thread pool and caches warm
        probe = go(8) / 8
        iters = int(max(8, min(200000, budget / max(probe, 1e-6))))
        dt = go(iters)
        return {"value": chains * iters / dt, "unit": "chain-iterations/s", "cores": threads, "kind": "port", "arithmetic": real,
                "grad_evals_per_s": chains * iters * (LEAP + 1) / dt,
                "sample": f"{chains} chains x {iters} HMC iterations (L={LEAP}) of the same n={N_ROWS},p={N_PAR} workload, "
                          f"{'float64' if real == 'f64' else 'float32'} C oracle, {threads} thread(s), {dt:.1f} s"}
    all64 = leg("f64", all_threads, target_s)
    one64 = leg("f64", 1, target_s * 0.7)
    all32 = leg("f32", all_threads, target_s * 0.7)
    one32 = leg("f32", 1, target_s * 0.7)
    return {**all64, "all_cores": all64, "one_core": one64, "fp32": {"all_cores": all32, "one_core": one32},
            "note": "value = the float64 all-cores leg; one_core / fp32 rows: SURVEY.md section 8(d)(ii).  The reference's own "
                    "NumPy script (1 core) is `reference_cpu`."}


class Timer:
    """HIP events on the stream the kernels are launched on (torch.cuda.Event would see torch's stream only)."""

    def __init__(self, L, check, dev, stream):
        self.L, self.check, self.dev, self.stream = L, check, dev, stream
        self.e0, self.e1 = Ct.c_void_p(), Ct.c_void_p()
        check(L.lr_event_create(dev, Ct.byref(self.e0)))
        check(L.lr_event_create(dev, Ct.byref(self.e1)))

    def start(self):
        self.check(self.L.lr_event_record(self.dev, self.e0, self.stream))

    def stop_ms(self) -> float:
        self.check(self.L.lr_event_record(self.dev, self.e1, self.stream))
        ms = Ct.c_float()
        self.check(self.L.lr_event_elapsed_ms(self.dev, self.e0, self.e1, Ct.byref(ms)))
        return float(ms.value)


def _timed_chainset(la, timer, cs, iters, thin, repeats=3, warm=0):
    """Best-of-`repeats` HIP-event time (ms) of one advance(iters, thin, keep=False) after one warm-up iteration and `warm` untimed
    launches of the timed length (the stepwise configurations: a few ms each, so that the timed ones run at the clocks the GPU holds
    under that load -- the power-bound config 5 whole measures 9 % slower in the first milliseconds after an idle gap)."""
    cs.advance(1, thin, keep=False)
    for _ in range(warm):
        cs.advance(iters, thin, keep=False)
    cs.sync()
    best = None
    for _ in range(repeats):
        timer.start()
        cs.advance(iters, thin, keep=False)
        ms = timer.stop_ms()
        best = ms if best is None or ms < best else best
    return best


def extra_configs(la, L, check, dev, stream):
    """BASELINE.json configs 1, 3, 4, 5 on ONE GPU, timed with HIP events on the launch stream (bounded: a few ms of
    GPU time each).  Step sizes: config 3 the reference's; configs 4/5 the tuned ones of
    tests/golden/fullsize_cfg{4,5}.json (acceptance 0.75-0.85: the MH test does real work)."""
    timer = Timer(L, check, dev, stream)
    res = []
    X, y = la.load_pima()
    pre = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    bmap = np.array([-9.19131622, 0.09705401, 0.03112265, -0.00564495, -0.00062272, 0.0814371, 1.26032561, 0.03939102])
    m = la.LogReg(X, y, np.array([10.0, 1, 1, 1, 1, 1, 1, 1]), device=dev)
    # ---- config 1: the reference's fit-numpy.py run as it is -- ONE chain, RWMH, 10 000 kept x thin 1000 on Pima -- bounded
    # to 1 000 kept samples (10^6 iterations, ~0.3 s; the rate does not depend on the length: one launch, one wave)
    k1 = la.mhKernel(m.lpost, la.rwProposal(0.02 * np.array([10.0, 1, 1, 1, 1, 1, 5, 1])))
    cs = la.ChainSet(k1, bmap, seed=1, stream=stream)
    cs.advance(1, 1000, keep=False)
    cs.sync()
    a0 = int(cs.get_accepts().sum())
    timer.start()
    s1 = cs.advance(1000, 1000)
    ms = timer.stop_ms()
    res.append({"config": 1, "workload": "RWMH prop sd 0.02*[10,1,1,1,1,1,5,1] on Pima n=200 p=8, ONE chain, thin 1000, 1000 of the "
                "reference's 10 000 kept samples (fit-numpy.py:86)", "kernel_variant": cs.plan(), "chain_iterations_per_s": 1e6 / (ms * 1e-3),
                "accept_rate": float((int(cs.get_accepts().sum()) - a0) / 1e6), "launch_ms": ms, "reference_cpu_it_per_s": 6486.0,
                "posterior_mean": np.asarray(s1.to_host(), dtype=np.float64).reshape(-1, 8).mean(axis=0).round(4).tolist(),
                "note": "a single chain is one 64-lane wave on one SIMD of the chip: a latency figure (0.3 us per iteration), not a "
                        "throughput one; the many-chain rate of the same kernel family is config 3's"})
    # ---- config 2 as the REFERENCE runs it: fit-np-hmc.py:105-108 -- HMC eps=1e-3, l=50, dmm=1/pre, thin 20, from the MAP, on
    # Pima -- at 4096 chains, every evaluation in fp32 (the headline's variant): like-for-like it/s and min-ESS/s beside the
    # script's own 1 368 it/s and 24.8 ESS/s (same posterior, same tuning, same estimator)
    C2, KEPT = CHAINS_PER_GPU, 512
    k2 = la.hmcKernel(m.lpost, m.glp, eps=1e-3, l=LEAP, dmm=1.0 / pre)
    cs = la.ChainSet(k2, np.tile(bmap, (C2, 1)), seed=2, stream=stream, precision="full")
    # launches of the headline's own shape (THIN iterations, one kept sample each), so that a rocprofv3 trace of this command
    # keeps ONE population of launches for the headline kernel
    for _ in range(20):  # 400 iterations away from the common start (the reference starts at the MAP and keeps everything)
        cs.advance(1, THIN, keep=False)
    cs.sync()
    a0 = int(cs.get_accepts().astype(np.int64).sum())
    s2 = la.DeviceArray(dev, (KEPT, C2, 8), np.float32)
    timer.start()
    for i in range(KEPT):
        cs.advance(1, THIN, keep=True, out=s2.rows(i, i + 1))
    ms = timer.stop_ms()
    acc2 = (int(cs.get_accepts().astype(np.int64).sum()) - a0) / (C2 * KEPT * THIN)
    draws = np.asarray(s2.to_host(), dtype=np.float64)
    s2.free()
    ess2 = la.ess_pooled(draws, max_chains=256)  # Geyer per chain on 512 kept draws, 256 of the 4096 chains, scaled
    its2 = C2 * KEPT * THIN / (ms * 1e-3)
    res.append({"config": "2_pima", "workload": f"HMC eps=1e-3 L={LEAP} dmm=1/[100,1,1,1,1,1,25,1] thin {THIN} on Pima n=200 p=8 from the MAP "
                f"(fit-np-hmc.py:105-108), {C2} chains x {KEPT} kept draws", "kernel_variant": cs.plan(), "precision": "full (all fp32)",
                "chain_iterations_per_s": its2, "grad_evals_per_s": its2 * LEAP, "accept_rate": float(acc2), "launch_ms": ms,
                "min_ess_per_s": float(ess2.min() / (ms * 1e-3)), "ess_per_kept_draw": (ess2 / (C2 * KEPT)).round(4).tolist(),
                "ess_estimator": "Geyer initial-positive-sequence per chain (the estimator of BASELINE.md's 24.8 ESS/s), 256 of the "
                                 f"{C2} chains, scaled; the {KEPT} timed launches of {THIN} iterations themselves",
                "posterior_mean": draws.reshape(-1, 8).mean(axis=0).round(4).tolist(),
                "reference_cpu_it_per_s": REFERENCE_CPU["it_per_s"], "reference_min_ess_per_s": REFERENCE_CPU["min_ess_per_s"],
                "speedup_it_per_s": its2 / REFERENCE_CPU["it_per_s"],
                "speedup_min_ess_per_s": float(ess2.min() / (ms * 1e-3)) / REFERENCE_CPU["min_ess_per_s"]})
    # ---- config 3: MALA, thin 1000, 8192 chains = one GPU's shard of 65 536 (real Pima data)
    k = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=pre)
    C3 = 8192
    cs = la.ChainSet(k, np.tile(bmap, (C3, 1)), seed=3, stream=stream)
    ms = _timed_chainset(la, timer, cs, 2, 1000)
    its = C3 * 2000
    fg = flops_per_grad_eval(200, 8)
    ach = its * fg / (ms * 1e-3) / 1e12
    res.append({"config": 3, "workload": "MALA dt=1e-5 pre=[100,1,..,25,1] on Pima n=200 p=8, thin 1000, 8192 chains "
                "(one GPU's shard of 65 536)", "kernel_variant": cs.plan(), "chain_iterations_per_s": its / (ms * 1e-3),
                "grad_evals_per_s": its / (ms * 1e-3), "accept_rate": float(cs.get_accepts().sum() / (C3 * 7000)),
                "launch_ms": ms, "reference_cpu_it_per_s": 4930.0,
                "roofline": {"bound": "valu_fp32", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / PEAK_FP32_TFLOPS, "flops_per_iteration": fg,
                             "note": "1 fused value+gradient evaluation per iteration (F_g flops counted; the value's "
                                     "log and the Philox/Box-Muller work, ~60 % of the instructions, are not)"}})
    # ---- configs 4 and 5: stepwise engines at full size, tuned step sizes from the committed fixtures
    # ("5_whole": config 5 AS A WHOLE -- all 8192 chains on this one GPU instead of one eighth of them: the matrix-pipe roofline point)
    # ("5_quarter" / "5_half": what one GPU does when config 5's 8192 chains are split over 4 / 2 GPUs -- with 5 and "5_whole" the four
    #  shard sizes of a 1 / 2 / 4 / 8-GPU run of the configuration AS STATED, each measured on this one GPU)
    for cfg, C, label in ((4, 1024, 4), (5, 1024, 5), (5, 2048, "5_quarter"), (5, 4096, "5_half"), (5, 8192, "5_whole")):
        fix = json.load(open(os.path.join(REPO, "tests", "golden", f"fullsize_cfg{cfg}.json")))
        n, p = fix["n"], fix["p"]
        X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
        m = la.LogReg(X, y, np.array(fix["pscale"]), device=dev)
        k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
        rng = np.random.Generator(np.random.Philox(4000 + cfg))
        q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * rng.standard_normal((C, p))
        cs = la.ChainSet(k, q0, seed=5, stream=stream)
        iters = 4
        ms = _timed_chainset(la, timer, cs, iters, 1, warm=2)
        evals = iters * fix["l"]  # per chain: L evaluations per iteration (the carried gradient saves the L+1-th)
        per_eval_s = ms * 1e-3 / evals
        fg = flops_per_grad_eval(n, p)
        ach = C * fg / per_eval_s / 1e12
        acc = float(cs.get_accepts().sum() / (C * (5 * iters + 1)))  # 1 warm-up iteration + (2 warm + 3 timed) launches
        row = {"config": label, "workload": f"HMC L={fix['l']} eps={fix['eps']} unit mass, synthetic n={n} p={p}, {C} chains"
               + ({5: " (one GPU's shard of 8192 over 8 GPUs)", "5_quarter": " (one GPU's shard of 8192 over 4 GPUs)", "5_half": " (one GPU's shard of 8192 over 2 GPUs)",
                   "5_whole": " (BASELINE.json configs[4] as a whole on ONE GPU)"}.get(label, "")),
               "kernel_variant": cs.plan(),
               "interior_precision": INTERIOR_NOTE,
               "chain_iterations_per_s": C * iters / (ms * 1e-3), "grad_evals_per_s": C * evals / (ms * 1e-3),
               "accept_rate": acc, "us_per_evaluation_all_chains": per_eval_s * 1e6,
               "timing": "HIP events around 4 HMC iterations = 200 log-posterior-gradient evaluations of all chains (every "
                         "kernel launch of the stepwise engine and the boundaries between them included)"}
        # what the sigmoid alone costs: 2 transcendental VALU ops (quarter rate: 8 cycles per wave-instruction) per row and chain
        trans_floor_s = n * C * 2 / 64 * 8 / (1024 * 2.4e9)
        if cfg == 4:
            xbytes = 4 * n * (p + 1)
            row["roofline"] = {"bound": "valu_trans", "pipe": "interior steps: exp + rcp of the sigmoid on the vector ALU, multiply-adds "
                               "on the bf16 matrix pipe (lr_tall_mx.h); end points: fp32 vector ALU",
                               "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                               "flops_per_grad_eval": fg, "transcendental_floor_us": trans_floor_s * 1e6,
                               "frac_of_transcendental_floor": trans_floor_s / per_eval_s,
                               "x_pass_bytes": xbytes, "x_pass_GBps": xbytes / per_eval_s / 1e9,
                               "hbm_frac": xbytes / per_eval_s / (HBM_PEAK_GBS * 1e9),
                               "note": "SURVEY 8(d): report both; algorithmic flops are priced against the fp32 vector peak (the pipe "
                                       "the fp32 formulation needs), but the interior multiply-adds run on the matrix pipe, so the "
                                       "operative bound is the transcendental unit; 1024 chains share every X pass: HBM is idle"}
        else:
            row["roofline"] = {"bound": "mfma", "pipe": MFMA16_PIPE,
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                               "frac_of_fp32_peak": ach / PEAK_FP32_TFLOPS, "flops_per_grad_eval": fg,
                               "note": "algorithmic flops counted once (SURVEY 8(d)); at 1024 chains an evaluation is 2.15 GFLOP = "
                                       "0.9 us of the bf16 pipe: launch, prologue and epilogue dominate (DESIGN.md section 5)" if C == 1024 else
                                       "algorithmic flops counted once (SURVEY 8(d)); the interior steps run as ONE launch per trajectory with one chain "
                                       "tile per workgroup (k_wide_traj2_bf16<.., 1>): a step costs what streaming the 1 MB image through one CU costs" if C < 8192 else
                                       "algorithmic flops counted once (SURVEY 8(d)); 17.4 GFLOP per evaluation; the interior steps run as ONE "
                                       "launch per trajectory (k_wide_traj2_bf16: 32 chains per workgroup, one workgroup per CU); rows and beta in "
                                       "one f16 piece each: the MFMAs issued are the algorithmic ones; power-bound at 1300 W / 2.04 GHz (profiles/r5_cfg5_whole*.txt)"}
        if label == "5_whole":
            # the same workload under precision="bf16" (the caller's explicit request: bf16 rows x beta in ONE bf16 piece on the trajectory
            # kernel -- the same MFMA count as the default's f16 pieces, 8 significant bits instead of 11) with the acceptance it costs
            cb = la.ChainSet(k, q0, seed=5, stream=stream, precision="bf16")
            msb = _timed_chainset(la, timer, cb, iters, 1, warm=2)
            row["precision_bf16"] = {"us_per_evaluation_all_chains": msb * 1e-3 / evals * 1e6, "accept_rate": float(cb.get_accepts().sum() / (C * (5 * iters + 1))),
                                     "frac_bf16_peak": C * fg / (msb * 1e-3 / evals) / 1e12 / PEAK_BF16_TFLOPS,
                                     "note": "not the default: acceptance drops by ~0.02 (0.758 -> 0.738); still an exact sampler"}
        res.append(row)
    # config 5 as stated (8192 chains) on 1 / 2 / 4 / 8 GPUs, PROJECTED from the four shard sizes measured above on this one GPU: chains are
    # independent, so a run's time is its slowest GPU's -- the only cross-GPU step, the gather of the kept samples, is not in it
    by = {r["config"]: r for r in res}
    if all(k in by for k in (5, "5_quarter", "5_half", "5_whole")):
        t1 = by["5_whole"]["us_per_evaluation_all_chains"]
        res.append({"config": "5_strong_scaling_projection",
                    "what": "config 5's 8192 chains over N GPUs: us per evaluation of a GPU's shard, measured on ONE GPU per shard size; NOT a multi-GPU measurement",
                    "rows": [{"n_gpus": n, "chains_per_gpu": 8192 // n, "us_per_evaluation": by[k]["us_per_evaluation_all_chains"],
                              "speedup_over_one_gpu": t1 / by[k]["us_per_evaluation_all_chains"]}
                             for n, k in ((1, "5_whole"), (2, "5_half"), (4, "5_quarter"), (8, 5))]})
    return res


def default_policy_runs(la, L, check, dev, stream, kern, init, steps):
    """The headline workload under the library's DEFAULT precision policy (LR_PREC_AUTO): the L-1 interior leapfrog
    gradients on the bf16 matrix pipe (any deterministic force keeps the leapfrog map reversible and volume-preserving),
    end points and the MH test in fp32 -- the sampler stays exact, the acceptance rate is the check.  Reported beside
    `value`, never as it: `value` is the all-fp32 run.  Same timing as the headline (HIP events, `steps` launches of
    THIN iterations), same data, 4096 chains as the headline and 16 384 / 65 536 chains to show where the kernel goes."""
    timer = Timer(L, check, dev, stream)
    rows = []
    for C in (CHAINS_PER_GPU, 16384, 65536):
        rng = np.random.Generator(np.random.Philox(SEED + 77))
        q0 = init + 0.1 * 0.17 * rng.standard_normal((C, N_PAR))
        row = {"chains": C}
        # "full" only at the headline's own chain count: the same kernel and grid as the timed run, so the per-kernel
        # averages of a rocprofv3 trace of this command stay those of the headline launch (the all-fp32 kernel at the
        # larger chain counts is in profiles/r2_mfma_chain_grid.txt)
        for prec in (("auto", "full") if C == CHAINS_PER_GPU else ("auto",)):
            cs = la.ChainSet(kern, q0, seed=SEED, stream=stream, precision=prec)
            cs.advance(2, THIN, keep=False)
            cs.sync()
            a0 = cs.get_accepts().astype(np.int64).sum()
            timer.start()
            for _ in range(steps):
                cs.advance(1, THIN, keep=False)
            ms = timer.stop_ms()
            acc = (cs.get_accepts().astype(np.int64).sum() - a0) / (C * steps * THIN)
            its = C * steps * THIN / (ms * 1e-3)
            row[prec] = {"kernel_variant": cs.plan(), "chain_iterations_per_s": its, "ms_per_step": ms / steps,
                         "accept_rate": float(acc),
                         "algorithmic_TFLOPs": its * LEAP * flops_per_grad_eval(N_ROWS, N_PAR) / 1e12}
        rows.append(row)
    return {"note": "same workload, HIP-event timed, not part of `value`: precision='auto' (library default) against "
                    "precision='full' (what `value` uses)", "runs": rows}


PEAK_FP64_TFLOPS = 78.6  # MI355X vector FP64 (cdna_hip_programming.md section 1: the SIMD-16 ceiling of the fp64 pipe)


def f64_run(la, L, check, dev, stream, X, y, pscale, q0, steps):
    """The headline workload on the float64 instantiation of the kernels (the reference's NumPy arithmetic is float64:
    fit-np-hmc.py:18-19).  The f64 path has no packed math and no matrix pipe at this width (rows in LDS, 16 lanes per chain
    at 4096 chains: lr_plan.h): the number says what the dtype choice of `value` buys."""
    timer = Timer(L, check, dev, stream)
    m64 = la.LogReg(X, y, pscale, dtype="float64", device=dev)
    k64 = la.hmcKernel(m64.lpost, m64.glp, eps=EPS, l=LEAP, dmm=np.ones(N_PAR))
    C = q0.shape[0]
    n = max(2, min(steps, 20))
    res = {}
    for prec in ("full", "auto"):
        cs = la.ChainSet(k64, q0, seed=SEED, stream=stream, precision=prec)
        t_pre = time.perf_counter()  # the headline's pre-warm: untimed launches until the GPU holds its clocks (PREWARM_S of load)
        while time.perf_counter() - t_pre < PREWARM_S:  # (the timed launch shape: a kernel trace's average IS the per-step time)
            for _ in range(8):
                cs.advance(1, THIN, keep=False)
            cs.sync()
        a0 = cs.get_accepts().astype(np.int64).sum()
        timer.start()
        for _ in range(n):
            cs.advance(1, THIN, keep=False)
        ms = timer.stop_ms()
        its = C * n * THIN / (ms * 1e-3)
        res[prec] = {"kernel_variant": cs.plan(), "chain_iterations_per_s": its, "ms_per_step": ms / n,
                     "accept_rate": float((cs.get_accepts().astype(np.int64).sum() - a0) / (C * n * THIN))}
    # ... and with many chains, where the default policy moves to the matrix-core kernel with a float64 state (lr_mfma_f64.h)
    Cm = 4 * C
    qm = np.tile(q0, (4, 1))
    cs = la.ChainSet(k64, qm, seed=SEED, stream=stream)
    cs.advance(2, THIN, keep=False)
    cs.sync()
    nm = max(2, n // 4)
    timer.start()
    for _ in range(nm):
        cs.advance(1, THIN, keep=False)
    ms = timer.stop_ms()
    many = {"chains": Cm, "kernel_variant": cs.plan(), "chain_iterations_per_s": Cm * nm * THIN / (ms * 1e-3), "ms_per_step": ms / nm}
    tf = res["full"]["chain_iterations_per_s"] * LEAP * flops_per_grad_eval(N_ROWS, N_PAR) / 1e12
    return {"dtype": "f64", **res["full"], "default_policy_many_chains": many, "algorithmic_TFLOPs": tf, "peak": PEAK_FP64_TFLOPS, "frac_of_fp64_vector_peak": tf / PEAK_FP64_TFLOPS,
            "default_policy": {**res["auto"], "note": "precision='auto' on the float64 model (LR_MODE_MIXED): float64 end points, Metropolis "
                               "test, position and momentum; float32 force inside the trajectory"},
            "note": f"same workload, {C} chains, {n} launches of {THIN} iterations, HIP-event timed, not part of `value`; top level = precision='full'"}


def host_boundary_run(la, kern, q0, steps):
    """The drop-in boundary as the reference's caller meets it: `mcmc(init, kernel, thin, iters)` takes a HOST array and returns the HOST
    array of samples (fit-np-hmc.py:89-108) -- host -> device state, the fused launches, the kept samples device -> host (PCIe) into the
    array `mcmc` returns.  Wall clock around the whole call; never `value` (whose inputs and outputs are resident)."""
    C = q0.shape[0]
    best = None
    for _ in range(2):  # (the second call: clocks and allocator warm)
        t0 = time.perf_counter()
        out = la.mcmc(q0, kern, thin=THIN, iters=steps, verb=False, seed=SEED, precision="full")
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return {"chain_iterations_per_s": C * steps * THIN / best, "seconds": best, "kept_samples": steps, "host_bytes_in": int(q0.size * 8),
            "device_to_host_bytes": int(C * steps * N_PAR * 4), "returned_bytes": int(out.nbytes),
            "note": "mcmc() host array in, host array out: PCIe transfers of the start and of every kept sample included; wall clock, best of 2 calls"}


def f64_wide_run(la, L, check, dev, stream):
    """BASELINE config 5 (n = 4096, p = 128, HMC L = 50, 1024 chains) on LogReg(dtype="float64"): the reference's own arithmetic at
    the wide shape, on the f64 matrix pipe (lr_wide_f64.h), every evaluation exact (precision="full") and under the default policy;
    2 iterations = 100 evaluations, HIP-event timed."""
    timer = Timer(L, check, dev, stream)
    fix = json.load(open(os.path.join(REPO, "tests", "golden", "fullsize_cfg5.json")))
    n, p, C = fix["n"], fix["p"], 1024
    X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
    m = la.LogReg(X, y, np.array(fix["pscale"]), dtype="float64", device=dev)
    k = la.hmcKernel(m.lpost, m.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    rng = np.random.Generator(np.random.Philox(4005))
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * rng.standard_normal((C, p))
    iters = 2
    evals = iters * (fix["l"] + 1)  # end points included
    res = {}
    for prec in ("full", "auto"):
        cs = la.ChainSet(k, q0, seed=5, stream=stream, precision=prec)
        ms = _timed_chainset(la, timer, cs, iters, 1, repeats=2)
        res[prec] = {"kernel_variant": cs.plan(), "chain_iterations_per_s": C * iters / (ms * 1e-3), "us_per_evaluation_all_chains": ms * 1e3 / evals,
                     "accept_rate": float(cs.get_accepts().sum() / (C * (2 * iters + 1)))}
    tf = C * flops_per_grad_eval(n, p) / (res["full"]["us_per_evaluation_all_chains"] * 1e-6) / 1e12
    # ... and config 5 as a whole (8192 chains) under the default policy: float64 state on the two-tile trajectory kernel
    Cw = 8192
    qw = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * np.random.Generator(np.random.Philox(4005)).standard_normal((Cw, p))
    cw = la.ChainSet(k, qw, seed=5, stream=stream)
    msw = _timed_chainset(la, timer, cw, iters, 1, repeats=2, warm=1)
    res["auto"]["whole_8192_chains"] = {"kernel_variant": cw.plan(), "us_per_evaluation_all_chains": msw * 1e3 / evals,
                                        "chain_iterations_per_s": Cw * iters / (msw * 1e-3),
                                        "accept_rate": float(cw.get_accepts().sum() / (Cw * (3 * iters + 1))),
                                        "note": "interior steps: k_wide_traj2_bf16<.., double> (float64 position / momentum / kick / drift, 16-bit force); "
                                                "end points on the f64 matrix pipe (about a third of the time)"}
    return {"dtype": "f64", "workload": f"HMC L={fix['l']} eps={fix['eps']}, synthetic n={n} p={p}, {C} chains, float64 model", **res["full"],
            "algorithmic_TFLOPs": tf, "peak": PEAK_FP64_TFLOPS, "frac_of_fp64_matrix_peak": tf / PEAK_FP64_TFLOPS,
            "default_policy": {**res["auto"], "note": "precision='auto': interior gradients on the 16-bit matrix pipe (the float32 engine's row-split kernel on a float64 state: "
                               "position, momentum and the fused update float64), end points on the f64 pipe"},
            "note": "top level = precision='full': every evaluation on v_mfma_f64_16x16x4_f64; not part of `value`"}


def ess_per_draw(la, model, kern, q0, dev, plan, precision):
    """ESS per kept draw (thin 20) from a SEPARATE run of 256 chains x 512 kept draws, Geyer IPS per chain, on the kernel
    variant and precision policy of the timed run (the plan is pinned: 256 chains alone would be planned otherwise).
    ESS per draw is a property of (eps, L) and the posterior, not of the variant -- pinning it makes the line say so."""
    cs = la.ChainSet(kern, q0[:256], seed=SEED + 7, mode=plan["mode"], group=plan["group"], precision=precision)
    cs.advance(1, 200, keep=False)
    s = cs.advance(512, THIN).to_host()
    ess = la.ess_pooled(s, max_chains=None)
    return ess / (s.shape[0] * s.shape[1]), s.shape, cs.plan()


class Exchange:
    """The multi-process side of the bench -- the ONE data-path collective (gather of the thinned samples to rank 0:
    RCCL over xGMI on GPUs), the `summary_only` alternative (one all-reduce of 7p + 1 doubles of on-device statistics),
    the barrier and the scalar reductions of the timing contract, and the small all-gathers the self-check of the
    line needs.  One object for the real run (backend "nccl", CUDA tensors viewing the library's sample buffers in
    place) and for `--dry-run` / the CPU tests (backend "gloo", CPU tensors), so the CPU tests execute the very code
    the 8-GPU run does.  world = 1 without LOGREG_BENCH_FORCE_DIST: every method is a no-op / identity."""

    def __init__(self, backend, rank, world, local_rank, force=False):
        self.rank, self.world, self.dist, self.torch, self.dev = rank, world, None, None, "cpu"
        self.backend = backend
        self._bufs = {}
        if world > 1 or force:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if backend == "nccl":
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only kind this driver supports
                # one rank per GPU or nothing: device_count() does not initialise the GPU, every rank sees the same count and
                # leaves before the rendezvous, so a node with fewer GPUs than ranks ends the job at once with a non-zero code
                have = torch.cuda.device_count()
                local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))  # (a multi-node launch: ranks of THIS node only)
                if have < local_world or local_rank >= have:
                    print(f"bench.py: {local_world} ranks on this node but only {have} GPU(s) visible -- refusing to share GPUs "
                          f"between ranks (rank {rank})", file=sys.stderr, flush=True)
                    sys.exit(3)
                torch.cuda.set_device(local_rank)
                self.dev = f"cuda:{local_rank}"
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)

    @property
    def active(self):
        return self.dist is not None

    def tensor(self, samples):
        """This rank's sample buffer (a DeviceArray viewed in place on its GPU, or a CPU tensor / ndarray) as a tensor on
        the exchange's device."""
        if not self.active:
            return None
        if isinstance(samples, self.torch.Tensor):
            return samples
        return self.torch.as_tensor(samples, device=self.dev)

    def sync(self):
        if self.active and self.dev != "cpu":
            self.torch.cuda.synchronize()

    def barrier(self):
        """barrier + device synchronize: the bracket of a timed region"""
        if self.active:
            self.dist.barrier()
            self.sync()

    def gather(self, t, source=None):
        """Gather every rank's block `t` to rank 0 -> (this rank's wall seconds, the list of blocks on rank 0 / None);
        complete on return.  The receive buffers are allocated once per block shape.  `source`: the buffer `t` was made from --
        on the CPU exchange (gloo) a tensor of a GPU buffer is a snapshot, so it is taken again here (two gloo ranks sharing one GPU:
        tests/test_gpu_distributed.py); the RCCL exchange views the buffer in place and ignores it."""
        if not self.active:
            return 0.0, None
        if source is not None and self.dev == "cpu":
            t = self.tensor(source)
        key = (tuple(t.shape), t.dtype)
        if self.rank == 0 and key not in self._bufs:
            self._bufs[key] = [self.torch.empty_like(t) for _ in range(self.world)]
        bufs = self._bufs.get(key)
        t0 = time.perf_counter()
        self.dist.gather(t, bufs, dst=0)
        self.sync()
        return time.perf_counter() - t0, bufs

    def reduce(self, value, op="max"):
        """max / min (float) or sum (int) over the ranks"""
        if not self.active:
            return value
        ops = {"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN, "sum": self.dist.ReduceOp.SUM}
        t = self.torch.tensor([value], device=self.dev, dtype=self.torch.int64 if op == "sum" else self.torch.float64)
        self.dist.all_reduce(t, op=ops[op])
        return int(t.item()) if op == "sum" else float(t.item())

    def all_gather_i64(self, value):
        """every rank's integer, on every rank"""
        if not self.active:
            return [int(value)]
        t = self.torch.tensor([int(value)], device=self.dev, dtype=self.torch.int64)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [int(o.item()) for o in out]

    def all_gather_str(self, text, width=192):
        """every rank's (short) string, on every rank"""
        if not self.active:
            return [text]
        raw = text.encode()[:width].ljust(width, b"\0")
        t = self.torch.tensor(list(raw), device=self.dev, dtype=self.torch.uint8)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [bytes(o.cpu().tolist()).rstrip(b"\0").decode() for o in out]

    def all_reduce_stats(self, sums, n_chains_local):
        """The `summary_only` exchange: logreg_amd.distributed.reduce_stats (what mcmc_sharded(summary_only=True) does) --
        one all-reduce of the [7, p] statistics sums + the chain count -> (sums, chains, this rank's wall seconds)."""
        if not self.active:
            return sums, n_chains_local, 0.0
        from logreg_amd.distributed import reduce_stats
        t0 = time.perf_counter()
        tot, ctot = reduce_stats(sums, n_chains_local, device=None if self.dev == "cpu" else self.dev)
        self.sync()
        return tot, ctot, time.perf_counter() - t0

    def close(self):
        if self.active:
            self.dist.barrier()
            self.dist.destroy_process_group()


CHECK_CHAINS = 64  # chains of another rank's block that rank 0 re-runs for `gather_bitexact`


def self_check(ex, ident, kernel_ms, gathered, check_rank, rerun):
    """What makes an N > 1 line verifiable from the line alone (every rank calls this; rank 0 gets the dict):
      ranks_seen        all-reduce of ones: the ranks the collective library actually connected
      devices           every rank's GPU identity (lr_device_info: PCI bus id, uuid), `devices_distinct`
      kernel_ms_min/max this rank's average launch duration (HIP events on its launch stream), min / max over the ranks;
                        kernel_ms_per_rank: all of them (main() turns them into per-rank chain-iterations/s and HBM fractions)
      gather_bitexact   rank 0 re-runs, on its own GPU and outside the timed region, the first CHECK_CHAINS chains of
                        rank `check_rank`'s block -- same start, same global chain ids (chain_offset), planned for the
                        block's chain count (plan_chains), the same launches -- and compares the kept samples with what
                        the gather delivered for that rank, bit for bit: the global-chain-id contract and
                        the gather's block order shown on the hardware of the very run that is reported."""
    seen = ex.reduce(1, "sum")
    devices = ex.all_gather_str(ident)
    kmin, kmax = ex.reduce(kernel_ms, "min"), ex.reduce(kernel_ms, "max")
    per_rank_ns = ex.all_gather_i64(round(kernel_ms * 1e6))  # every rank's own average launch duration (HIP events), in ns
    if ex.rank != 0:
        return None
    res = {"ranks_seen": seen, "devices": devices, "devices_distinct": len(set(devices)) == len(devices),
           "kernel_ms_min": kmin, "kernel_ms_max": kmax, "kernel_ms_per_rank": [v / 1e6 for v in per_rank_ns]}
    if gathered is not None and rerun is not None:
        got = gathered[check_rank].cpu().numpy()
        ref = rerun(check_rank)
        k = ref.shape[1]
        res.update(gather_bitexact=bool(np.array_equal(got[:, :k, :], ref)), gather_checked_rank=check_rank,
                   gather_checked_chains=k,
                   gather_check="rank 0 re-ran chains [rank*C, rank*C + %d) of rank %d (chain_offset = rank*C, plan_chains = C, "
                                "the same launches) and compared with that rank's gathered block" % (k, check_rank))
    return res


def dist_configs(la, L, check, dev, stream, ex, rank, world, scale=1):
    """BASELINE.json configs 3 and 5 AS STATED -- multi-GPU: 8192 MALA chains per rank (65 536 at N = 8) with the RCCL
    gather of a kept block and, separately, the `summary_only` all-reduce of 7p + 1 doubles; 1024 wide-model HMC chains per
    rank (8192 at N = 8) with the gather -- each timed like `value`: barrier + device synchronise on both sides, wall
    time max over ranks, whole-job units / that time; plus each rank's kernel time (HIP events) min / max over ranks.
    Every rank runs this; rank 0 returns the rows.  `scale` > 1 divides chain counts and run lengths (CPU tests only:
    the rows are then marked `scaled_down` and are not measurements of the configs)."""
    timer = Timer(L, check, dev, stream)
    rows = []
    tag = {"scaled_down": scale} if scale != 1 else {}

    def timed(launch, exchange):
        """barrier | launch (HIP events) + sync + exchange | barrier -> (wall max over ranks, kernel ms min, max, exchange s max, payload)"""
        ex.barrier()
        t0 = time.perf_counter()
        timer.start()
        launch()
        kms = timer.stop_ms()  # synchronises on the stop event: the launches are complete
        xs, payload = exchange()
        ex.barrier()
        wall = ex.reduce(time.perf_counter() - t0, "max")
        return wall, ex.reduce(kms, "min"), ex.reduce(kms, "max"), ex.reduce(xs, "max"), payload

    # ---- config 3: MALA dt=1e-5 pre=[100,1,..,25,1] on Pima, thin 1000, 8192 chains per rank
    X, y = la.load_pima()
    pre = np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    bmap = np.array([-9.19131622, 0.09705401, 0.03112265, -0.00564495, -0.00062272, 0.0814371, 1.26032561, 0.03939102])
    m = la.LogReg(X, y, np.array([10.0, 1, 1, 1, 1, 1, 1, 1]), device=dev)
    k3 = la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=pre)
    C3, KEEP3, THIN3 = max(16, 8192 // scale), 4, max(10, 1000 // scale)
    cs = la.ChainSet(k3, np.tile(bmap, (C3, 1)), seed=3, chain_offset=rank * C3, stream=stream)
    out3 = la.DeviceArray(dev, (KEEP3, C3, 8), np.float32)
    t3 = ex.tensor(out3)
    cs.advance(1, THIN3, keep=False)
    cs.sync()
    ex.gather(t3)  # untimed first use of this block shape
    wall, kmin, kmax, xs, bufs = timed(lambda: cs.advance(KEEP3, THIN3, keep=True, out=out3), lambda: ex.gather(t3, out3))
    its = world * C3 * KEEP3 * THIN3
    fg = flops_per_grad_eval(200, 8)
    # the same chains' on-device statistics instead of their samples: 7p + 1 doubles per rank, one all-reduce
    cs.enable_stats(KEEP3 // 2, 2, pivot=bmap)

    def summary_exchange():  # device reduction over this rank's chains (lr_stats_reduce), then the all-reduce
        t0 = time.perf_counter()
        tot, ctot, _ = ex.all_reduce_stats(cs.stats_sums(), C3)
        return time.perf_counter() - t0, (tot, ctot)
    wall_s, kmin_s, kmax_s, xs_s, summ = timed(lambda: cs.advance(KEEP3, THIN3, keep=False), summary_exchange)
    if rank == 0:
        ach = its * fg / wall / world / 1e12
        blocks_ok = all(tuple(b.shape) == (KEEP3, C3, 8) and bool(np.isfinite(b.cpu().numpy()).all()) for b in bufs)
        rows.append({"config": 3, **tag, "n_gpus": world,
                     "workload": f"MALA dt=1e-5 pre=[100,1,..,25,1] on Pima n=200 p=8, thin {THIN3}, {C3} chains per GPU = "
                                 f"{world * C3} chains (BASELINE.json configs[2]: 65 536 over 8 GPUs), {KEEP3} kept samples",
                     "kernel_variant": cs.plan(), "chains_total": world * C3,
                     "with_gather": {"chain_iterations_per_s": its / wall, "wall_ms": wall * 1e3, "gather_ms": xs * 1e3,
                                     "gathered_bytes_per_rank": KEEP3 * C3 * 8 * 4, "blocks_ok": blocks_ok,
                                     "kernel_ms_min": kmin, "kernel_ms_max": kmax,
                                     "exchange": "gather of the kept [iters, C, p] block of every rank to rank 0 (RCCL)"},
                     "summary_only": {"chain_iterations_per_s": its / wall_s, "wall_ms": wall_s * 1e3, "allreduce_ms": xs_s * 1e3,
                                      "doubles_per_rank": 7 * 8 + 1, "chains_counted": int(summ[1]),
                                      "kernel_ms_min": kmin_s, "kernel_ms_max": kmax_s,
                                      "exchange": "lr_stats_reduce on every rank + ONE all-reduce of 7p + 1 doubles (no samples stored or moved)"},
                     "roofline": {"bound": "valu_fp32", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s per GPU",
                                  "frac": ach / PEAK_FP32_TFLOPS, "flops_per_iteration": fg,
                                  "note": "whole-job flops / wall (max over ranks, gather included) / GPUs"}})
    out3.free()
    # ---- config 5: HMC L=50 on synthetic n=4096 p=128, 1024 chains per rank
    fix = json.load(open(os.path.join(REPO, "tests", "golden", "fullsize_cfg5.json")))
    n, p, C5 = fix["n"], fix["p"], max(16, 1024 // scale)
    X, y, _ = la.synthetic_logreg(n, p, seed=fix["data_seed"], beta_sd=fix["beta_sd"])
    m5 = la.LogReg(X, y, np.array(fix["pscale"]), device=dev)
    k5 = la.hmcKernel(m5.lpost, m5.glp, eps=fix["eps"], l=fix["l"], dmm=np.array(fix["dmm"]))
    rng = np.random.Generator(np.random.Philox(4005 + 1000 * rank))
    q0 = np.array(fix["map"]) + np.array(fix["laplace_sd"]) * rng.standard_normal((C5, p))
    cs = la.ChainSet(k5, q0, seed=5, chain_offset=rank * C5, stream=stream)
    iters = 4 if scale == 1 else 1
    out5 = la.DeviceArray(dev, (iters, C5, p), np.float32)
    t5 = ex.tensor(out5)
    cs.advance(1, 1, keep=False)
    cs.sync()
    ex.gather(t5)
    a0 = int(cs.get_accepts().astype(np.int64).sum())
    wall, kmin, kmax, xs, bufs = timed(lambda: cs.advance(iters, 1, keep=True, out=out5), lambda: ex.gather(t5, out5))
    acc = ex.reduce(int(cs.get_accepts().astype(np.int64).sum()) - a0, "sum")
    if rank == 0:
        evals = iters * fix["l"]
        fg = flops_per_grad_eval(n, p)
        ach = world * C5 * evals * fg / wall / world / 1e12
        rows.append({"config": 5, **tag, "n_gpus": world,
                     "workload": f"HMC L={fix['l']} eps={fix['eps']} unit mass, synthetic n={n} p={p}, {C5} chains per GPU = {world * C5} "
                                 f"chains (BASELINE.json configs[4]: 8192 over 8 GPUs), {iters} iterations, every sample kept",
                     "kernel_variant": cs.plan(), "chains_total": world * C5,
                     "interior_precision": INTERIOR_NOTE,
                     "chain_iterations_per_s": world * C5 * iters / wall, "grad_evals_per_s": world * C5 * evals / wall,
                     "accept_rate": acc / (world * C5 * iters), "wall_ms": wall * 1e3, "gather_ms": xs * 1e3,
                     "gathered_bytes_per_rank": iters * C5 * p * 4, "kernel_ms_min": kmin, "kernel_ms_max": kmax,
                     "us_per_evaluation_all_chains_of_a_gpu": wall / evals * 1e6,
                     "blocks_ok": all(tuple(b.shape) == (iters, C5, p) for b in bufs),
                     "roofline": {"bound": "mfma", "pipe": MFMA16_PIPE,
                                  "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s per GPU", "frac": ach / PEAK_BF16_TFLOPS,
                                  "flops_per_grad_eval": fg, "note": "whole-job flops / wall (max over ranks, gather included) / GPUs"}})
    out5.free()
    return rows if rank == 0 else None


def dry_run(a, rank, world, local_rank):
    """Launch plumbing without a GPU and without any sampler: the same Exchange object and the same `self_check` as the
    real run on the gloo backend -- shard offsets, gather to rank 0, barrier, reductions, rank / device roll call, and the
    gather_bitexact re-run -- with a pure function of (global chain id, step) standing in for the sample buffer."""
    import torch
    ex = Exchange("gloo", rank, world, local_rank)
    C = a.chains

    def stand_in(r, chains):  # "samples" of chains [r*C, r*C + chains) at every step
        g = (r * C + np.arange(chains))[None, :, None]
        return (np.sin(0.001 * g + np.arange(a.steps)[:, None, None]) + 0.01 * np.arange(N_PAR)[None, None, :]).astype(np.float32)
    t = ex.tensor(torch.from_numpy(stand_in(rank, C)))
    ex.gather(t)  # untimed first use, as in the real run
    ex.barrier()
    t0 = time.perf_counter()
    gather_s, gathered = ex.gather(t)
    ex.barrier()
    wall = ex.reduce(time.perf_counter() - t0, "max")
    gather_s = ex.reduce(gather_s, "max")
    nacc = ex.reduce(rank + 1, "sum")
    chk = self_check(ex, f"pci=dry:{rank} uuid={rank:032x} name=dry-run cus=0", 0.1 * (rank + 1), gathered, min(1, world - 1),
                     lambda r: stand_in(r, min(CHECK_CHAINS, C)))
    if rank == 0:
        ok = world == 1 or ([tuple(g.shape) for g in gathered] == [(a.steps, C, N_PAR)] * world and
                            all(np.array_equal(g.numpy(), stand_in(r, C)) for r, g in enumerate(gathered)))
        print(json.dumps({"dry_run": True, "metric": "MCMC iterations/sec x chains for HMC (L=50) on n=200,p=8", "value": None,
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "chain_offsets": [r * C for r in range(world)],
                          "this_rank_offset": rank * C, "gather_ok": bool(ok), "gather_ms": gather_s * 1e3, "wall_ms": wall * 1e3,
                          "sum_over_ranks_ok": nacc == world * (world + 1) // 2, "scaling": "weak", **(chk or {})}), flush=True)
    ex.close()


# MAP of the headline design (BFGS on the float64 oracle, computed once offline)
INIT = np.array([-0.65920504, -0.18123564, -0.64985465, -0.19187958, -0.11223836, -0.51230749, -0.10401207, -0.8432688])


def headline_init(rank, C):
    """Start of rank `rank`'s chains: MAP + 0.1 * posterior-sd scale * N(0,1), from a
    generator keyed by the rank, so that any rank can regenerate any other's block (`gather_bitexact`)."""
    rng = np.random.Generator(np.random.Philox(SEED + 1000 * rank))
    return INIT + 0.1 * 0.17 * rng.standard_normal((C, N_PAR))


def self_launch(n):
    """`python bench.py --gpus N` as a PLAIN process (no WORLD_SIZE / RANK in the environment), N > 1: start the N ranks
    ourselves -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    <this script> <the same arguments>` as a CHILD process (never an exec), before this process has imported the package,
    torch or touched a GPU -- and hand back its exit code.  Rank 0's JSON line reaches stdout through the inherited pipe.
    The ranks themselves refuse to run when the node shows fewer than N GPUs (Exchange.__init__), so the code is non-zero
    then.  The torchrun entry (`WORLD_SIZE` set by the launcher) is untouched."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0]), *sys.argv[1:]]
    print(f"bench.py: --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--group", type=int, default=0, help="lanes per chain (0 = library's choice)")
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--precision", default="full", choices=["full", "auto", "bf16"],
                    help="headline run: 'full' = every gradient evaluation in fp32 (the default: `value` is an all-fp32 number); "
                         "'auto' = the library's default policy (HMC interior gradients on the bf16 matrix pipe)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs 1/3/4/5 sub-results")
    ap.add_argument("--host-boundary", action="store_true", help="also time the headline through mcmc(): host array in, host samples out (PCIe inclusive)")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only (gloo, no GPU work)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo: the CPU tests)")
    ap.add_argument("--scale", type=int, default=1, help="divide the chain counts / run lengths of the multi-GPU configs 3 and 5 "
                    "(CPU tests only; rows are marked scaled_down)")
    ap.add_argument("--prewarm", type=float, default=PREWARM_S, help="seconds of untimed load before the warm-up steps")
    a = ap.parse_args(argv)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} != --gpus {a.gpus}; using WORLD_SIZE", file=sys.stderr)
    if a.dry_run:
        return dry_run(a, rank, world, local_rank)
    # (LOGREG_BENCH_FORCE_DIST=1 runs the RCCL path at N = 1: tests/test_gpu_distributed.py)
    ex = Exchange(a.backend, rank, world, local_rank, force=os.environ.get("LOGREG_BENCH_FORCE_DIST") == "1")

    import logreg_amd as la
    from logreg_amd import _lib

    X, y, _ = la.synthetic_logreg(N_ROWS, N_PAR, seed=20240001)
    pscale = np.array([10.0] + [1.0] * (N_PAR - 1))
    init = INIT
    C = a.chains
    q0 = headline_init(rank, C)

    dev = local_rank if ex.active and a.backend == "nccl" else 0
    model = la.LogReg(X, y, pscale, dtype="float32", device=dev)
    kern = la.hmcKernel(model.lpost, model.glp, eps=EPS, l=LEAP, dmm=np.ones(N_PAR))
    L = _lib.load()
    from logreg_amd import build as lib_build
    dev_flags = lib_build.built_extra()
    stream = Ct.c_void_p()
    _lib.check(L.lr_stream_create(dev, Ct.byref(stream)))
    cs = la.ChainSet(kern, q0, seed=SEED, chain_offset=rank * C, group=a.group, mode=a.mode, stream=stream, precision=a.precision)
    plan = cs.plan()
    out = la.DeviceArray(dev, (a.steps, C, N_PAR), np.float32)

    def one_step(i, keep):
        cs.advance(1, THIN, keep=keep, out=out.rows(i % a.steps, i % a.steps + 1))

    # Untimed pre-warm: the GPU needs some tens of milliseconds of load before it holds its clocks (measured: the same
    # 50 timed steps run at 0.417-0.424 ms after 5 warm-up steps = 2 ms of load, at 0.389-0.391 ms after 200 or 1000), and
    # `value` is meant to be the sustained rate whatever W the caller picks.  Same launches as the timed ones.
    t_pre, n_pre = time.perf_counter(), 0
    while time.perf_counter() - t_pre < a.prewarm:
        for _ in range(20):
            one_step(n_pre, True)
            n_pre += 1
        cs.sync()
    for i in range(a.warmup):
        one_step(i, True)
    cs.sync()

    timer = Timer(L, _lib.check, dev, stream)
    tout = ex.tensor(out)
    ex.gather(tout)  # untimed first use: RCCL sets up its p2p channels
    acc0 = cs.get_accepts().astype(np.int64).sum()
    ex.barrier()  # barrier + device synchronize on both sides of the timed region
    t0 = time.perf_counter()
    timer.start()
    for i in range(a.steps):
        one_step(i, True)
    _lib.check(L.lr_event_record(dev, timer.e1, stream))
    cs.sync()
    gather_s, gathered = ex.gather(tout, out)  # the ONE exchange of the path: RCCL gather of the thinned samples to rank 0
    ex.barrier()
    t1 = time.perf_counter()
    ms = Ct.c_float()
    _lib.check(L.lr_event_elapsed_ms(dev, timer.e0, timer.e1, Ct.byref(ms)))
    kernel_ms_total = float(ms.value)
    wall = ex.reduce(t1 - t0, "max")
    gather_s = ex.reduce(gather_s, "max")
    acc = ex.reduce(int(cs.get_accepts().astype(np.int64).sum() - acc0), "sum")

    check_block = None
    if ex.active:
        launches_before = ex.all_gather_i64(n_pre + a.warmup)  # the pre-warm is time-bounded: every rank's own count

        def rerun(r):
            k = min(CHECK_CHAINS, C)
            c2 = la.ChainSet(kern, headline_init(r, C)[:k], seed=SEED, chain_offset=r * C, group=a.group, mode=a.mode,
                             stream=stream, precision=a.precision, plan_chains=C, plan_first=r * C)
            for _ in range(launches_before[r]):
                c2.advance(1, THIN, keep=False)
            buf = la.DeviceArray(dev, (a.steps, k, N_PAR), np.float32)
            for i in range(a.steps):
                c2.advance(1, THIN, keep=True, out=buf.rows(i, i + 1))
            c2.sync()
            res = buf.to_host()
            buf.free()
            return res
        check_block = self_check(ex, _lib.device_info(dev), kernel_ms_total / a.steps, gathered, min(1, world - 1), rerun)
    dist_rows = None
    if ex.active and not a.no_extra:
        dist_rows = dist_configs(la, L, _lib.check, dev, stream, ex, rank, world, a.scale)

    if rank == 0:
        iters_total = world * C * a.steps * THIN
        grad_evals = iters_total * LEAP  # executed: the gradient at the current state is carried, L per iteration
        kern_s = kernel_ms_total / 1e3 / a.steps  # average launch duration from HIP events on the launch stream
        fg = flops_per_grad_eval(N_ROWS, N_PAR)
        achieved = C * THIN * LEAP * fg / kern_s / 1e12
        alg_bytes = C * N_PAR * 4 * 3 + C * 4 + N_ROWS * N_PAR * 4  # state r/w + sample + accepts + X once
        value = iters_total / wall
        line = {
            "metric": "MCMC iterations/sec x chains for HMC (L=50) on n=200,p=8",
            "value": value,
            "unit": "chain-iterations/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * wall / a.steps,
            # the exchange's share of the timed region (max over ranks; 0 without a process group): separates compute from
            # the gather in a 1 -> 8 GPU curve
            "gather_ms": gather_s * 1e3,
            "prewarm": {"seconds": a.prewarm, "steps": n_pre, "note": "untimed launches before the W warm-up steps, until the GPU holds its clocks"},
            "higher_is_better": True,
            "scaling": "weak",
            # BASELINE.md's number for this metric: the reference's own fit-np-hmc.py, 1 368 it/s x 1 chain (1 core)
            "vs_baseline": value / REFERENCE_CPU["it_per_s"],
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "HMC L=50 eps=0.1 unit-mass, n=200 p=8 synthetic logistic regression, "
                                   f"{C} chains/GPU, thin {THIN} (BASELINE.json configs[1])",
                       "chains_per_gpu": C, "thin": THIN, "leapfrog_steps": LEAP,
                       "precision": {"full": "full: every log-posterior / gradient evaluation in fp32",
                                     "auto": "auto (library default): L-1 interior leapfrog gradients on the bf16 matrix pipe, "
                                             "end-point value + gradient and the MH test in fp32",
                                     "bf16": "bf16 interior gradients forced"}[a.precision],
                       "kernel_variant": plan, "parallelism": f"chains sharded x{world}" + (" + RCCL gather" if world > 1 else "")},
            "grad_evals_per_s": grad_evals / wall,
            "accept_rate": acc / iters_total,
            "reference_cpu": REFERENCE_CPU,
            # compute-bound on the fp32 VECTOR ALU (MFMA busy = 0 in the PMC passes): priced against the dense fp32
            # peak, 157.3 TFLOP/s (the guide's figure for v_pk_fma_f32 and for fp32-input MFMA alike)
            "roofline": {"bound": "valu_fp32", "pipe": "fp32 vector ALU (v_pk_fma_f32) + transcendental unit",
                         "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": None,
                         "kernel_ms": kern_s * 1e3,
                         # SURVEY.md section 8(d)'s own estimate of what the VALU + transcendental mix allows
                         # (1.0e10 grad evals/s/GPU), reported beside the fraction of the nominal peak
                         "frac_of_survey_ceiling": grad_evals / wall / world / 1.0e10,
                         "flops_per_grad_eval": fg, "grad_evals_per_launch": C * THIN * LEAP,
                         "algorithmic_hbm_bytes_per_launch": alg_bytes,
                         "hbm_GBps_algorithmic": alg_bytes / kern_s / 1e9,
                         "note": "X lives in VGPRs for the whole launch: the path is compute-bound on the fp32 vector "
                                 "ALU, not HBM-bound (8 TB/s: see hbm_frac) and not on the matrix cores; DESIGN.md section 5"},
        }
        if model.debug_opts():  # A/B switches (LOGREG_DEBUG_OPTS) in effect: not a default run
            line["debug_opts"] = model.debug_opts()
        if dev_flags:  # a development build (e.g. -DLR_STAMPS) is not what `value` is meant to measure: say so in the line
            line["development_build_flags"] = dev_flags
        if check_block is not None:
            line["multi_gpu"] = check_block
            # north_star: "chains/sec and achieved-HBM-fraction reported at 1/2/4/8 GPUs" -- per rank, from the rank's own HIP-event time
            per = []
            for r, kms in enumerate(check_block.get("kernel_ms_per_rank", [])):
                ks = max(kms, 1e-9) / 1e3
                per.append({"rank": r, "kernel_ms": kms, "chain_iterations_per_s": C * THIN / ks, "grad_evals_per_s": C * THIN * LEAP / ks,
                            "valu_frac": C * THIN * LEAP * fg / ks / 1e12 / PEAK_FP32_TFLOPS,
                            "hbm_GBps_algorithmic": alg_bytes / ks / 1e9, "hbm_frac": alg_bytes / ks / (HBM_PEAK_GBS * 1e9)})
            line["per_rank"] = per
        # the weak-scaling row of the headline (fixed chains per GPU): what a 1 / 2 / 4 / 8 GPU series of these lines is a curve of
        line["weak_scaling"] = {"chains_per_gpu": C, "n_gpus": world, "chains_total": world * C, "chain_iterations_per_s": value,
                                "per_gpu": value / world, "kernel_only_per_gpu": C * THIN / kern_s, "gather_share": gather_s / wall if wall > 0 else 0.0,
                                "note": "efficiency is the driver's to compute from its own N = 1, 2, 4, 8 runs; no 1 -> 8 GPU curve has been "
                                        "measured on hardware by this repo"}
        # HBM traffic of the same launch from the committed rocprofv3 PMC passes (profiles/), if they
        # were taken for this kernel variant and shape
        for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
            try:
                tr = json.load(open(os.path.join(REPO, "profiles", name)))
                if tr["kernel_variant"] == plan and tr["chains"] == C and tr["thin"] == THIN:
                    line["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = f"profiles/{name} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
                    line["roofline"]["hbm_GBps_measured"] = tr["hbm_bytes_per_launch"] / kern_s / 1e9
                    line["roofline"]["hbm_frac"] = tr["hbm_bytes_per_launch"] / kern_s / (HBM_PEAK_GBS * 1e9)
                    break
            except (OSError, KeyError, ValueError):
                pass
        if not a.no_ess:
            eff, shape, ess_plan = ess_per_draw(la, model, kern, q0, dev, plan, a.precision)
            draws_per_s = world * C * a.steps / wall
            line["min_ess_per_s"] = float(eff.min() * draws_per_s)
            line["ess"] = {"ess_per_kept_draw_min": float(eff.min()), "ess_per_kept_draw_max": float(eff.max()),
                           "kept_draws_per_s": draws_per_s,
                           "estimator": f"Geyer initial-positive-sequence per chain on a separate run of {shape[1]} chains x "
                                        f"{shape[0]} kept draws (thin {THIN}), outside the timed region; min over the {N_PAR} "
                                        "parameters; ESS/s = ESS per kept draw x kept draws/s of the timed run",
                           "kernel_variant": ess_plan, "precision": a.precision,
                           "not_comparable_with": "reference_cpu.min_ess_per_s (24.8): that is Pima at eps=1e-3, dmm=1/pre; this is the "
                                                  "synthetic headline design at eps=0.1, unit mass.  The like-for-like pair is "
                                                  "extra.configs[config=2_pima]"}
        if dist_rows is not None:
            # N > 1 (or the forced one-rank process group): BASELINE configs 3 and 5 as stated, across the ranks
            line["extra"] = {"configs": dist_rows}
        elif world == 1 and not a.no_extra:
            f64 = f64_run(la, L, _lib.check, dev, stream, X, y, pscale, q0, a.steps)
            # TOP LEVEL: the same workload in the reference's own arithmetic (Python/fit-np-hmc.py:17-19 computes in float64; `value`
            # above is the float32 instantiation the contract prescribes) -- HIP-event timed, same chains, same launches
            line["reference_arithmetic"] = {
                "dtype": "f64", "value": f64["chain_iterations_per_s"], "unit": "chain-iterations/s", "ms_per_step": f64["ms_per_step"],
                "frac": f64["frac_of_fp64_vector_peak"], "peak": PEAK_FP64_TFLOPS, "peak_unit": "TFLOP/s (fp64 vector)",
                "kernel_variant": f64["kernel_variant"], "accept_rate": f64["accept_rate"],
                "policy_auto_value": f64["default_policy"]["chain_iterations_per_s"], "policy_auto_kernel_variant": f64["default_policy"]["kernel_variant"],
                "ratio_to_value": f64["chain_iterations_per_s"] / value,
                "note": "precision='full': every log-posterior / gradient evaluation float64 (k_chain_f64x, lr_f64x.h); policy_auto_value: the "
                        "float64 model's default policy (float64 state, end points and Metropolis test; float32 force inside the trajectory)"}
            if a.host_boundary:  # the PCIe-inclusive rate of the same workload.  Opt-in: mcmc() runs its 200 kept samples as
                # ONE launch of the headline kernel, which would drag that kernel's AVERAGE duration in a rocprofv3 trace of this command
                # away from the per-step time the roofline is computed from
                line["host_boundary"] = host_boundary_run(la, kern, q0, a.steps)
            line["extra"] = {"configs": extra_configs(la, L, _lib.check, dev, stream),
                             "default_policy": default_policy_runs(la, L, _lib.check, dev, stream, kern, init, a.steps),
                             "f64": f64,
                             "f64_wide": f64_wide_run(la, L, _lib.check, dev, stream)}
    # the process group is closed first: at N > 1 the other ranks leave, and rank 0 times the CPU oracle on an otherwise idle
    # host (they would spin in a barrier otherwise) before it prints the ONE line.  Whatever happens in the teardown or in the CPU
    # legs, the measured line is printed: the failure is recorded in it.
    try:
        ex.close()
    except Exception as e:  # noqa: BLE001 -- a hung / failed RCCL teardown must not lose the measurement
        if rank == 0:
            line["teardown_error"] = repr(e)[:300]
    if rank == 0:
        try:
            if not a.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(X, y, pscale, init)
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"error": repr(e)[:300]}
        finally:
            print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
