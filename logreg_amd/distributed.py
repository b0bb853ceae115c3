"""Multi-GPU execution: independent chain blocks, one process per GPU, one gather at the end.

The reference has no distributed code on this path; chains never interact (`mcmc` carries only
`(x, ll)`, Python/fit-np-mala.py:80-95), so the path shards trivially:

* rank r of G owns the contiguous chain block `shard_bounds(C, G, r)`; the model data and kernel
  parameters are replicated (a few KB);
* the Philox counter uses the GLOBAL chain id (`chain_offset`), so the samples are identical for
  every G (bit-exact with the single-GPU run);
* the only exchange is a gather of the thinned samples `[iters, C_r, p]` to rank 0 --
  `torch.distributed.gather` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
  tests).  With 7 direct xGMI links per GPU every sender has its own link into rank 0.

* when only pooled posterior summaries are wanted, `reduce_moments` exchanges (2p + 1) float64 sufficient
  statistics per rank with one all-reduce instead of moving any samples.

`run_sharded` takes the per-rank compute as a callable so the sharding/gather logic is testable
without a GPU (tests inject the CPU oracle); `mcmc_sharded` binds it to the fused HIP kernels.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_chains: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_chains), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(local, n_chains: int, dst: int = 0, group=None):
    """Gather per-rank sample blocks `[iters, C_r, p]` (torch tensors, ragged in C_r) into
    `[iters, C, p]` on rank `dst`; returns None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(n_chains, world, r) for r in range(world)]
    cmax = max(hi - lo for lo, hi in sizes)
    iters, c_local, p = local.shape
    assert c_local == sizes[rank][1] - sizes[rank][0], (c_local, sizes[rank])
    if c_local < cmax:  # pad ragged shards so one fixed-size gather serves all ranks
        pad = torch.zeros((iters, cmax - c_local, p), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=1)
    local = local.contiguous()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:, : hi - lo, :] for b, (lo, hi) in zip(bufs, sizes)], dim=1)


def reduce_moments(local, group=None):
    """Pooled posterior mean and SD (ddof = 1, as `scipy.stats.describe` in fit-np-hmc.py:113-117) over the
    samples of ALL ranks from one all-reduce of sufficient statistics: count, sum and sum of squares per
    parameter, accumulated in float64.  `local` is this rank's `[iters, C_r, p]` block (tensor on the rank's
    device, or ndarray).  Every rank gets `{"n", "mean", "sd"}`; no samples move."""
    import torch
    import torch.distributed as dist
    t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(local))
    x = t.reshape(-1, t.shape[-1]).to(torch.float64)
    # centre on a common pivot (rank 0's first sample) so that sum-of-squares cancellation stays harmless
    pivot = x[0].clone() if x.shape[0] else torch.zeros(t.shape[-1], dtype=torch.float64, device=t.device)
    dist.broadcast(pivot, src=0, group=group)
    d = x - pivot
    stats = torch.cat([torch.tensor([float(x.shape[0])], dtype=torch.float64, device=t.device), d.sum(0), (d * d).sum(0)])
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    p = t.shape[-1]
    n, s1, s2 = stats[0], stats[1:1 + p], stats[1 + p:]
    mean_d = s1 / n
    var = (s2 - n * mean_d * mean_d) / (n - 1)
    return {"n": int(n.item()), "mean": (pivot + mean_d).cpu().numpy(), "sd": var.clamp_min(0).sqrt().cpu().numpy()}


def run_sharded(init, run_block, n_chains: int | None = None, dst: int = 0, group=None, device=None):
    """Shard `init [C, p]` over the ranks of the initialised process group, call
    `run_block(init_block, chain_offset) -> ndarray | tensor [iters, C_r, p]` on each rank and
    gather the results on `dst`."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    init = np.asarray(init)
    C = init.shape[0] if n_chains is None else n_chains
    lo, hi = shard_bounds(C, world, rank)
    out = run_block(init[lo:hi], lo)
    if not isinstance(out, torch.Tensor):
        out = torch.as_tensor(np.ascontiguousarray(out))
    if device is not None:
        out = out.to(device)
    return gather_samples(out, C, dst=dst, group=group)


def mcmc_sharded(init, make_kernel, thin=10, iters=10000, seed=0, dst=0, group=None, local_device=None, **kw):
    """Many-chain `mcmc` across all ranks (one process per GPU, launched with torchrun).

    `make_kernel(device) -> FusedKernel` builds the rank's model + kernel on its own GPU.
    Returns the gathered `[iters, C, p]` tensor on rank `dst` (on that rank's GPU), else None."""
    import os
    import torch
    import torch.distributed as dist
    from .kernels import ChainSet
    if local_device is None:
        local_device = int(os.environ.get("LOCAL_RANK", dist.get_rank(group)))
    kernel = make_kernel(local_device)

    def run_block(block, chain_offset):
        cs = ChainSet(kernel, block, seed, chain_offset=chain_offset, **kw)
        out = cs.advance(iters, thin)
        cs.sync()
        t = torch.as_tensor(out, device=f"cuda:{local_device}").clone()  # own the memory beyond `out`
        return t
    return run_sharded(init, run_block, dst=dst, group=group)
