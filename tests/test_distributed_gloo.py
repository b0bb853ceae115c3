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
