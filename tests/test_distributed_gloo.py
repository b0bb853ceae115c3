"""world_size-2 gloo test of the multi-GPU path's host logic (sharding, global chain ids, ragged
gather).  The per-rank compute is injected: here the CPU oracle stands in for the HIP kernels
(tests may use the oracle), which also proves shard-invariance of the Philox stream end to end."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, C, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import OracleModel
    from logreg_amd.data import load_pima
    from logreg_amd.distributed import run_sharded
    X, y = load_pima()
    m = OracleModel(X, y, [10, 1, 1, 1, 1, 1, 1, 1])
    dmm = 1.0 / np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    init = np.load(os.path.join(tmp, "init.npy"))

    def run_block(block, chain_offset):
        return m.run("hmc", block, step=1e-3, l=5, scale=dmm, thin=2, iters=3, seed=31, chain_offset=chain_offset)["out"]
    out = run_sharded(init, run_block)
    # the no-samples-moved alternative: pooled mean / SD from one all-reduce of sufficient statistics
    from logreg_amd.distributed import reduce_moments, shard_bounds
    lo, hi = shard_bounds(C, world, rank)
    mom = reduce_moments(run_block(init[lo:hi], lo))
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), out.numpy())
        np.savez(os.path.join(tmp, "moments.npz"), **mom)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("C", [10, 7])
def test_sharded_run_equals_single_process_run(tmp_path, C):
    import torch.multiprocessing as mp
    from oracle.oracle import OracleModel
    from logreg_amd.data import load_pima
    X, y = load_pima()
    m = OracleModel(X, y, [10, 1, 1, 1, 1, 1, 1, 1])
    dmm = 1.0 / np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    rng = np.random.default_rng(0)
    init = np.array([-9.19, 0.097, 0.031, -0.0056, -0.0006, 0.0814, 1.26, 0.0394]) + 0.01 * rng.standard_normal((C, 8))
    np.save(tmp_path / "init.npy", init)
    mp.spawn(_worker, args=(2, _free_port(), C, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    ref = m.run("hmc", init, step=1e-3, l=5, scale=dmm, thin=2, iters=3, seed=31)["out"]
    assert got.shape == (3, C, 8)
    np.testing.assert_array_equal(got, ref)  # bit-exact: chain ids are global
    mom = np.load(tmp_path / "moments.npz")
    flat = ref.reshape(-1, 8)
    assert int(mom["n"]) == flat.shape[0]
    np.testing.assert_allclose(mom["mean"], flat.mean(0), rtol=1e-12)
    np.testing.assert_allclose(mom["sd"], flat.std(0, ddof=1), rtol=1e-9)


# ------------------------------------------------------------------------------------------------
# mcmc_sharded ITSELF (chunked launches, global chain ids, ragged and empty shards, gather, and the
# statistics all-reduce) with a CPU ChainSet injected through its factory argument: same interface as
# logreg_amd.kernels.ChainSet, the float64 oracle as the engine.
class _OracleChainSet:
    def __init__(self, kernel, block, seed, chain_offset=0, **kw):
        self.k, self.seed, self.chain_offset = kernel, seed, chain_offset
        self.state = np.array(block, dtype=np.float64)
        self.iter_offset = 0
        self.acc = np.zeros(self.state.shape[0], dtype=np.uint64)
        self.kept, self.batch, self.pivot = None, 0, None
        self.launches = 0
        self.kw = kw

    def enable_stats(self, batch, slots, pivot=None):
        self.kept, self.batch, self.pivot = [], batch, np.asarray(pivot)

    def advance(self, iters, thin, keep=True, stats=None):
        r = self.k.oracle.run("hmc", self.state, thin=thin, iters=iters, seed=self.seed, chain_offset=self.chain_offset,
                              iter_offset=self.iter_offset, **self.k.kw)
        self.state, self.iter_offset = r["state"], self.iter_offset + iters * thin
        self.acc += r["accepts"]
        self.launches += 1
        if self.kept is not None:
            self.kept.append(r["out"])
        return r["out"].astype(np.float32) if keep else None

    def sync(self):
        pass

    def get_accepts(self):
        return self.acc

    def stats_sums(self):
        from logreg_amd.diagnostics import batch_sums
        return batch_sums(np.concatenate(self.kept), self.batch, self.pivot)


def _worker_mcmc_sharded(rank, world, port, C, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import types
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import OracleModel
    from logreg_amd.data import load_pima
    from logreg_amd.distributed import mcmc_sharded
    X, y = load_pima()
    dmm = 1.0 / np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    init = np.load(os.path.join(tmp, "init.npy"))

    def make_kernel(dev):  # what FusedKernel exposes to mcmc_sharded: kind, params, model.n/.p
        return types.SimpleNamespace(kind="hmc", params={"l": 5}, model=types.SimpleNamespace(n=200, p=8),
                                     oracle=OracleModel(X, y, [10, 1, 1, 1, 1, 1, 1, 1]), kw=dict(step=1e-3, l=5, scale=dmm))
    made = []

    def factory(*a, **k):
        made.append(_OracleChainSet(*a, **k))
        return made[-1]
    out = mcmc_sharded(init, make_kernel, thin=2, iters=8, seed=31, chunk=3, chainset_factory=factory)
    assert (made[0].launches == 3) if made else (C < world and rank >= C)  # 8 kept samples in chunks of 3, 3, 2
    assert all("plan_chains" not in cs.kw for cs in made)  # plan="local" (the default): every rank plans for its own shard
    n0 = len(made)
    again = mcmc_sharded(init, make_kernel, thin=2, iters=8, seed=31, chunk=3, chainset_factory=factory, plan="global", precision="auto")
    # plan="global": every shard plans for ALL C chains (lr_run_opts.plan_chains), whatever else the caller passed on
    assert all(cs.kw == {"plan_chains": C, "precision": "auto"} for cs in made[n0:])
    assert (again is None) == (rank != 0) and (rank != 0 or np.array_equal(again.numpy(), out.numpy()))
    summ = mcmc_sharded(init, make_kernel, thin=2, iters=8, seed=31, summary_only=True, max_batches=4, chainset_factory=factory)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), out.numpy())
    else:
        assert out is None
    np.savez(os.path.join(tmp, f"summary{rank}.npz"), **{k: v for k, v in summ.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("C", [7, 1])
def test_mcmc_sharded_driver_on_cpu(tmp_path, C):
    """world_size 2; C = 7: ragged shards (4 + 3); C = 1: rank 1's shard is EMPTY and still joins the collectives."""
    import torch.multiprocessing as mp
    from oracle.oracle import OracleModel
    from logreg_amd.data import load_pima
    from logreg_amd.diagnostics import split_rhat
    X, y = load_pima()
    m = OracleModel(X, y, [10, 1, 1, 1, 1, 1, 1, 1])
    dmm = 1.0 / np.array([100.0, 1, 1, 1, 1, 1, 25, 1])
    rng = np.random.default_rng(1)
    init = np.array([-9.19, 0.097, 0.031, -0.0056, -0.0006, 0.0814, 1.26, 0.0394]) + 0.01 * rng.standard_normal((C, 8))
    np.save(tmp_path / "init.npy", init)
    mp.spawn(_worker_mcmc_sharded, args=(2, _free_port(), C, str(tmp_path)), nprocs=2, join=True)
    ref = m.run("hmc", init, step=1e-3, l=5, scale=dmm, thin=2, iters=8, seed=31)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == (8, C, 8)
    np.testing.assert_array_equal(got, ref["out"].astype(np.float32))  # chunked + sharded = monolithic, bit for bit
    flat = ref["out"].reshape(-1, 8)
    for rank in range(2):  # every rank holds the summary of ALL chains
        s = np.load(tmp_path / f"summary{rank}.npz")
        assert int(s["n"]) == flat.shape[0] and int(s["chains"]) == C
        np.testing.assert_allclose(s["mean"], flat.mean(0), rtol=1e-12)
        np.testing.assert_allclose(s["sd"], flat.std(0, ddof=1), rtol=1e-8)
        np.testing.assert_allclose(s["rhat"], split_rhat(ref["out"]), rtol=1e-8)
        assert float(s["accept_rate"]) == pytest.approx(ref["accepts"].sum() / (C * 16))


def _c_exchange_worker(rank, world, port, tmp):
    """One rank = one process: the C-level exchange of the ABI (lr_comm_* / lr_gather / lr_allreduce_sum_f64, here the CPU test double's
    shared-memory implementation) next to torch.distributed (gloo) on the SAME blocks."""
    import ctypes as C
    import time
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch
    import torch.distributed as dist
    import twin
    L = twin.install()
    from logreg_amd import _lib
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the communicator id travels from rank 0 to the others as RCCL's does on the GPU: here a file
    idf = os.path.join(tmp, "comm.id")
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        _lib.check(L.lr_comm_unique_id(ident))
        with open(idf + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(idf + ".tmp", idf)
    else:
        while not os.path.exists(idf):
            time.sleep(0.01)
        ident = (C.c_ubyte * 128).from_buffer_copy(open(idf, "rb").read())
    comm = C.c_void_p()
    _lib.check(L.lr_comm_create(ident, rank, world, 0, C.byref(comm)))
    rng = np.random.default_rng(100 + rank)
    ok = True
    for root in range(world):  # the thinned samples of this rank's chain block: [iters][chains per rank][p] float32
        block = rng.standard_normal((3, 40, 8)).astype(np.float32)
        recv = np.full((world,) + block.shape, np.nan, dtype=np.float32)
        _lib.check(L.lr_gather(comm, block.ctypes.data, recv.ctypes.data if rank == root else None, block.nbytes, root, None))
        t = torch.from_numpy(block)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == root else None
        dist.gather(t, bufs, dst=root)
        if rank == root:
            ok &= all(np.array_equal(recv[r], bufs[r].numpy()) for r in range(world))  # every rank's block, in rank order, bit for bit
    # the statistics all-reduce: 7 p + 1 doubles per rank (lr_stats_reduce's sums + the chain count)
    sums = rng.standard_normal(57) * 10.0 ** rng.integers(-3, 6, 57)
    mine = sums.copy()
    _lib.check(L.lr_allreduce_sum_f64(comm, mine.ctypes.data, mine.size, None))
    t = torch.from_numpy(sums.copy())
    dist.all_reduce(t)
    gathered = [torch.empty(57, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(sums))
    in_rank_order = gathered[0].numpy().copy()
    for r in range(1, world):
        in_rank_order += gathered[r].numpy()
    ok &= np.array_equal(mine, in_rank_order)                   # the sum in rank order: the same bits on every rank
    ok &= np.allclose(mine, t.numpy(), rtol=1e-14, atol=0)      # torch's all_reduce (its own order) to rounding
    bad_root = L.lr_gather(comm, block.ctypes.data, None, block.nbytes, world, None)
    ok &= bad_root < 0
    _lib.check(L.lr_comm_destroy(comm))
    res = torch.tensor([int(ok)])
    dist.all_reduce(res, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(os.path.join(tmp, "c_exchange_ok"), "w") as f:
            f.write(str(int(res.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_c_level_exchange_between_processes_matches_torch_distributed(tmp_path, world):
    """`lr_comm_create` at world > 1 over a file-passed id, `lr_gather` (every root) and `lr_allreduce_sum_f64` --
    the exchange a plain-C client of the ABI makes on the GPU through RCCL -- run between PROCESSES on the CPU test double and compared
    with torch.distributed's gather / all_reduce of the same blocks: the same block order, the root alone receives, sums in rank order."""
    import torch.multiprocessing as mp
    mp.spawn(_c_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert open(os.path.join(str(tmp_path), "c_exchange_ok")).read() == "1"
