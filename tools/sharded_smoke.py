#!/usr/bin/env python3
"""Exercise logreg_amd.distributed.mcmc_sharded over RCCL (run with torchrun on a GPU box; works
with --nproc-per-node 1).  Checks the gathered samples against a single-process run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import logreg_amd as la
from logreg_amd.distributed import mcmc_sharded

rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
if os.environ.get("SHARDED_SMOKE_BACKEND") == "gloo":  # several ranks sharing GPU 0 (a one-GPU test box), the exchange on the CPU
    lr = 0
    dist.init_process_group("gloo")
else:
    torch.cuda.set_device(lr)
    dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
X, y = la.load_pima()
pre = np.array([100., 1, 1, 1, 1, 1, 25, 1])
init = np.tile([-9.19131622, 0.09705401, 0.03112265, -0.00564495, -0.00062272, 0.0814371, 1.26032561, 0.03939102], (1000, 1))

def make_kernel(dev):
    m = la.LogReg(X, y, [10, 1, 1, 1, 1, 1, 1, 1], device=dev)
    return la.malaKernel(m.lpost, m.glp, dt=1e-5, pre=pre)

out = mcmc_sharded(init, make_kernel, thin=5, iters=7, seed=11, local_device=lr, plan="global")
if rank == 0:
    ref = la.mcmc(init, make_kernel(lr), thin=5, iters=7, seed=11, verb=False)
    got = out.cpu().numpy()
    assert got.shape == (7, 1000, 8), got.shape
    assert np.array_equal(got, ref), "sharded run differs from the single-process run"
    print("sharded_smoke ok: world", world, "gathered", got.shape, "bit-exact vs single process")
# the no-samples path: on-device statistics of every rank's shard, one all-reduce of 7p + 1 doubles over RCCL
summ = mcmc_sharded(init, make_kernel, thin=5, iters=8, seed=11, summary_only=True, max_batches=4, plan="global", local_device=lr)
if rank == 0:
    ref = la.mcmc(init, make_kernel(lr), thin=5, iters=8, seed=11, verb=False).astype(np.float64)
    flat = ref.reshape(-1, 8)
    assert summ["n"] == flat.shape[0]
    assert np.allclose(summ["mean"], flat.mean(0), rtol=1e-9) and np.allclose(summ["sd"], flat.std(0, ddof=1), rtol=1e-6)
    assert np.allclose(summ["rhat"], la.split_rhat(ref), rtol=1e-6)
    print("sharded_smoke ok: summary_only over", summ["chains"], "chains: mean", np.round(summ["mean"], 4))
dist.barrier(); dist.destroy_process_group()
