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

* `reduce_stats` goes one step further: the statistics themselves are accumulated ON THE DEVICE while sampling
  (`ChainSet.enable_stats`), reduced over the rank's chains by `lr_stats_reduce`, and 7p + 1 doubles per rank are
  all-reduced: posterior mean / sd / split-R-hat / ESS of 65 536 chains without ever storing `[iters, C, p]`.

`run_sharded` takes the per-rank compute as a callable so the sharding/gather logic is testable
without a GPU (tests inject the CPU oracle); `mcmc_sharded` binds it to the fused HIP kernels through a
`ChainSet` factory that the CPU tests replace as well.
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
    if local.is_cuda and dist.get_backend(group) != "nccl":  # a CPU exchange (gloo: tests, ranks sharing one GPU) moves host tensors
        local = local.cpu()
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


def reduce_stats(sums, n_chains_local: int, group=None, device=None):
    """All-reduce (SUM) the chain-pooled statistics sums of every rank's shard -- the `[7, p]` array of
    `ChainSet.stats_sums()` (device accumulators reduced over the rank's chains by `lr_stats_reduce`) -- plus the
    chain count: 7p + 1 float64 per rank cross xGMI, no samples.  Every rank gets `(sums_total, n_chains_total)`;
    feed them to `diagnostics.summary_from_sums`.  All ranks must have used the same pivot and batch length."""
    import torch
    import torch.distributed as dist
    flat = np.concatenate([np.asarray(sums, dtype=np.float64).ravel(), [float(n_chains_local)]])
    t = torch.as_tensor(flat)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    flat = t.cpu().numpy()
    return flat[:-1].reshape(np.shape(sums)), int(round(flat[-1]))


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
    if hi == lo:  # more ranks than chains: the callable's output shape is unknown here (mcmc_sharded handles this case itself)
        raise ValueError(f"run_sharded: rank {rank} of {world} has no chains ({C} chains in all); use mcmc_sharded or fewer ranks")
    out = run_block(init[lo:hi], lo)
    if not isinstance(out, torch.Tensor):
        out = torch.as_tensor(np.ascontiguousarray(out))
    if device is not None:
        out = out.to(device)
    return gather_samples(out, C, dst=dst, group=group)


def _as_tensor(arr, device):
    """Samples of one shard as a torch tensor: a DeviceArray is viewed in place on its GPU (and cloned, so the
    tensor owns its memory); anything else goes through NumPy (CPU tests)."""
    import torch
    if hasattr(arr, "__cuda_array_interface__"):
        return torch.as_tensor(arr, device=f"cuda:{device}").clone()
    return torch.as_tensor(np.ascontiguousarray(arr))


def mcmc_sharded(init, make_kernel, thin=10, iters=10000, seed=0, dst=0, group=None, local_device=None,
                 chunk=None, summary_only=False, max_batches=16, plan="local", chainset_factory=None, **kw):
    """Many-chain `mcmc` across all ranks (one process per GPU, launched with torchrun).

    `make_kernel(device) -> FusedKernel` builds the rank's model + kernel on its own GPU.  Rank r runs the
    contiguous chain block `shard_bounds(C, world, r)` with its GLOBAL chain ids in the Philox counter, in chunked
    launches like `mcmc()`; an empty shard (more ranks than chains) runs nothing and still takes part in the
    collective.

    Returns the gathered `[iters, C, p]` tensor on rank `dst` (on that rank's GPU), else None -- or, with
    `summary_only=True`, on EVERY rank the posterior summary dict of all chains from one all-reduce of the
    on-device statistics (`reduce_stats`); no samples are stored or moved.

    `plan="global"` makes every rank plan for all C chains (`lr_run_opts.plan_chains = C`): the kernel variant, the
    row slicing of the stepwise engine and of its reduced-precision interior kernels and the trajectory-kernel choice
    are then the ones a single GPU running all C chains would make -- for this kernel family and this `precision` --
    so the output is bit-identical to the one-GPU run (summation order is a property of those choices); the default
    "local" lets every rank plan for its own shard size (statistically identical, fastest).
    `chainset_factory(kernel, block, seed, chain_offset=..., **kw)` defaults to `ChainSet` (tests inject a CPU one).
    """
    import os
    import torch
    import torch.distributed as dist
    from .diagnostics import choose_batches, summary_from_sums
    from .kernels import ChainSet, _auto_chunk
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if local_device is None:
        local_device = int(os.environ.get("LOCAL_RANK", rank))
    factory = ChainSet if chainset_factory is None else chainset_factory
    init = np.asarray(init, dtype=np.float64)
    C, p = init.shape
    lo, hi = shard_bounds(C, world, rank)
    kernel = make_kernel(local_device)
    if plan == "global":
        kw = dict(kw, plan_chains=C)
    elif plan != "local":
        raise ValueError("plan must be 'local' or 'global'")
    cs = factory(kernel, init[lo:hi], seed, chain_offset=lo, **kw) if hi > lo else None
    if chunk is None:
        chunk = _auto_chunk(kernel, hi - lo, thin, iters) if hasattr(kernel, "params") else iters
    pivot = init[0]

    if summary_only:
        batch, slots = choose_batches(iters, max_batches)
        sums = np.zeros((7, p))
        if cs is not None:
            cs.enable_stats(batch, slots, pivot=pivot)
            done = 0
            while done < iters:
                k = min(chunk, iters - done)
                cs.advance(k, thin, keep=False)
                cs.sync()
                done += k
            sums = cs.stats_sums()
        dev = f"cuda:{local_device}" if dist.get_backend(group) == "nccl" else None
        tot, ctot = reduce_stats(sums, hi - lo, group=group, device=dev)
        acc = np.array([float(cs.get_accepts().sum()) if cs is not None else 0.0])
        t = torch.as_tensor(acc)
        t = t.to(dev) if dev else t
        dist.all_reduce(t, group=group)
        res = summary_from_sums(tot, ctot, iters, batch, pivot)
        res.update(accept_rate=float(t.item()) / (C * iters * thin), batch=batch)
        return res

    if cs is None:
        local = np.zeros((iters, 0, p), dtype=np.float32)
    else:
        outs, done = [], 0
        while done < iters:  # chunked like mcmc(): bounded launches, identical samples
            k = min(chunk, iters - done)
            outs.append(_as_tensor(cs.advance(k, thin), local_device))
            cs.sync()
            done += k
        local = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
    if not isinstance(local, torch.Tensor):
        local = torch.as_tensor(local)
        if dist.get_backend(group) == "nccl":
            local = local.to(f"cuda:{local_device}")
    return gather_samples(local, C, dst=dst, group=group)
