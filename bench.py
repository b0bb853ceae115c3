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
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_ROWS, N_PAR, LEAP, EPS, THIN = 200, 8, 50, 0.1, 20
CHAINS_PER_GPU = 4096
SEED = 42
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector == FP32-input MFMA peak
HBM_PEAK_GBS = 8000.0


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


def cpu_baseline(X, y, pscale, init, target_s=12.0):
    """Time the CPU oracle (float64 C restatement, OpenMP over chains) on a bounded sample of the
    same workload.  The oracle is the checker, used here only as the reported CPU baseline."""
    from oracle.oracle import OracleModel, max_threads
    m = OracleModel(X, y, pscale)
    threads = min(max_threads(), usable_cores())
    chains = 8 * threads
    st = np.tile(init, (chains, 1))
    t0 = time.perf_counter()
    m.run("hmc", st, step=EPS, l=LEAP, scale=np.ones(N_PAR), thin=1, iters=4, seed=SEED, keep=False, threads=threads)
    probe = (time.perf_counter() - t0) / 4
    iters = int(max(4, min(200000, target_s / max(probe, 1e-6))))
    t0 = time.perf_counter()
    m.run("hmc", st, step=EPS, l=LEAP, scale=np.ones(N_PAR), thin=1, iters=iters, seed=SEED, keep=False, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": chains * iters / dt, "unit": "chain-iterations/s", "cores": threads, "kind": "port",
            "sample": f"{chains} chains x {iters} HMC iterations (L={LEAP}) of the same n={N_ROWS},p={N_PAR} workload, "
                      f"float64 C oracle, {dt:.1f} s",
            "grad_evals_per_s": chains * iters * (LEAP + 1) / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--group", type=int, default=0, help="lanes per chain (0 = library's choice)")
    ap.add_argument("--mode", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if rank == 0:
            print(f"warning: WORLD_SIZE={world} != --gpus {a.gpus}; using WORLD_SIZE", file=sys.stderr)
    dist = None
    if world > 1 or os.environ.get("LOGREG_BENCH_FORCE_DIST") == "1":  # the env var exercises the RCCL path at N=1
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import logreg_amd as la
    from logreg_amd import _lib

    X, y, _ = la.synthetic_logreg(N_ROWS, N_PAR, seed=20240001)
    pscale = np.array([10.0] + [1.0] * (N_PAR - 1))
    # MAP of this design (BFGS on the float64 oracle, computed once offline); chains start at
    # MAP + 0.1*N(0,1)*posterior-sd-scale, as SURVEY.md section 8(d) config 2 prescribes
    init = np.array([-0.65920504, -0.18123564, -0.64985465, -0.19187958, -0.11223836, -0.51230749, -0.10401207,
                     -0.8432688])
    C = a.chains
    rng = np.random.Generator(np.random.Philox(SEED + 1000 * rank))
    q0 = init + 0.1 * 0.17 * rng.standard_normal((C, N_PAR))

    dev = local_rank if dist is not None else 0
    model = la.LogReg(X, y, pscale, dtype="float32", device=dev)
    kern = la.hmcKernel(model.lpost, model.glp, eps=EPS, l=LEAP, dmm=np.ones(N_PAR))
    L = _lib.load()
    import ctypes as Ct
    stream = Ct.c_void_p()
    _lib.check(L.lr_stream_create(dev, Ct.byref(stream)))
    cs = la.ChainSet(kern, q0, seed=SEED, chain_offset=rank * C, group=a.group, mode=a.mode, stream=stream)
    plan = cs.plan()
    out = la.DeviceArray(dev, (a.steps, C, N_PAR), np.float32)

    def one_step(i, keep):
        cs.advance(1, THIN, keep=keep, out=out.rows(i % a.steps, i % a.steps + 1))

    for i in range(a.warmup):
        one_step(i, True)
    cs.sync()

    ev0, ev1 = Ct.c_void_p(), Ct.c_void_p()
    _lib.check(L.lr_event_create(dev, Ct.byref(ev0)))
    _lib.check(L.lr_event_create(dev, Ct.byref(ev1)))
    gathered = None
    if dist is not None:
        import torch
        tout = torch.as_tensor(out, device=f"cuda:{dev}")
        gathered = [torch.empty_like(tout) for _ in range(world)] if rank == 0 else None
        dist.gather(tout, gathered, dst=0)  # untimed warm-up: RCCL sets up its p2p channels on first use
        dist.barrier()
        torch.cuda.synchronize()
    acc0 = cs.get_accepts().astype(np.int64).sum()
    t0 = time.perf_counter()
    _lib.check(L.lr_event_record(dev, ev0, stream))
    for i in range(a.steps):
        one_step(i, True)
    _lib.check(L.lr_event_record(dev, ev1, stream))
    cs.sync()
    if dist is not None:
        dist.gather(tout, gathered, dst=0)  # RCCL gather of the thinned samples to rank 0
        torch.cuda.synchronize()
        dist.barrier()
    t1 = time.perf_counter()
    ms = Ct.c_float()
    _lib.check(L.lr_event_elapsed_ms(dev, ev0, ev1, Ct.byref(ms)))
    wall = t1 - t0
    acc = cs.get_accepts().astype(np.int64).sum() - acc0
    if dist is not None:
        tw = torch.tensor([wall], device=f"cuda:{dev}", dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        ta = torch.tensor([acc], device=f"cuda:{dev}", dtype=torch.int64)
        dist.all_reduce(ta)
        acc = int(ta.item())

    if rank == 0:
        iters_total = world * C * a.steps * THIN
        grad_evals = iters_total * LEAP  # executed: the gradient at the current state is carried, L per iteration
        kern_s = ms.value / 1e3 / a.steps  # average launch duration from HIP events on the launch stream
        fg = flops_per_grad_eval(N_ROWS, N_PAR)
        achieved = C * THIN * LEAP * fg / kern_s / 1e12
        alg_bytes = C * N_PAR * 4 * 3 + C * 4 + N_ROWS * N_PAR * 4  # state r/w + sample + accepts + X once
        line = {
            "metric": "MCMC iterations/sec x chains for HMC (L=50) on n=200,p=8",
            "value": iters_total / wall,
            "unit": "chain-iterations/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * wall / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "HMC L=50 eps=0.1 unit-mass, n=200 p=8 synthetic logistic regression, "
                                   f"{C} chains/GPU, thin {THIN} (BASELINE.json configs[1])",
                       "chains_per_gpu": C, "thin": THIN, "leapfrog_steps": LEAP,
                       "kernel_variant": plan, "parallelism": f"chains sharded x{world}" + (" + RCCL gather" if world > 1 else "")},
            "grad_evals_per_s": grad_evals / wall,
            "accept_rate": acc / iters_total,
            # compute-bound: priced against the dense fp32 peak (157.3 TFLOP/s: the fp32-input MFMA peak and the
            # fp32 vector-ALU peak are the same number and the same multipliers); `pipe` says which one runs
            "roofline": {"bound": "mfma", "pipe": "fp32 vector ALU (v_pk_fma_f32) + transcendental unit",
                         "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": None,
                         "kernel_ms": kern_s * 1e3,
                         # SURVEY.md section 8(d)'s own estimate of what the VALU + transcendental mix allows
                         # (1.0e10 grad evals/s/GPU), reported beside the fraction of the nominal peak
                         "frac_of_survey_ceiling": grad_evals / wall / world / 1.0e10,
                         "flops_per_grad_eval": fg, "grad_evals_per_launch": C * THIN * LEAP,
                         "algorithmic_hbm_bytes_per_launch": alg_bytes,
                         "hbm_GBps_algorithmic": alg_bytes / kern_s / 1e9,
                         "note": "X lives in VGPRs for the whole launch: the path is compute-bound on the fp32 "
                                 "multipliers (dense fp32 peak 157.3 TFLOP/s, shared by v_pk_fma_f32 and fp32-input "
                                 "MFMA), not HBM-bound (8 TB/s: see hbm_frac); see DESIGN.md section 5"},
        }
        # HBM traffic of the same launch from the committed rocprofv3 PMC passes (profiles/), if they
        # were taken for this kernel variant and shape
        try:
            tr = json.load(open(os.path.join(REPO, "profiles", "r1_traffic.json")))
            if tr["kernel_variant"] == plan and tr["chains"] == C and tr["thin"] == THIN:
                line["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = "profiles/r1_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
                line["roofline"]["hbm_GBps_measured"] = tr["hbm_bytes_per_launch"] / kern_s / 1e9
                line["roofline"]["hbm_frac"] = tr["hbm_bytes_per_launch"] / kern_s / (HBM_PEAK_GBS * 1e9)
        except (OSError, KeyError, ValueError):
            pass
        if not a.no_ess:
            samples = out.to_host()  # rank 0's chains, [steps, C, p]
            ess = la.ess_pooled(samples, max_chains=64)
            line["min_ess_per_s"] = float(world * ess.min() / wall)
            line["ess_note"] = "Geyer IPS per chain, summed over chains (64-chain subsample scaled), rank 0 x n_gpus"
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(X, y, pscale, init)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
