"""The real sharded path -- logreg_amd.distributed.mcmc_sharded on the HIP kernels with torch.distributed's RCCL
backend -- launched the way the driver launches bench.py: a fresh child process under torch.distributed.run.
One GPU is all a test box has, so N = 1; the N > 1 logic (ragged/empty shards, global chain ids, chunked launches,
the statistics all-reduce) is covered by the world-size-2 gloo tests in test_distributed_gloo.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_mcmc_sharded_over_rccl_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(REPO, "tools", "sharded_smoke.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "bit-exact vs single process" in r.stdout and "summary_only over 1000 chains" in r.stdout


def test_mcmc_sharded_with_three_ranks_sharing_the_gpu_over_gloo():
    """mcmc_sharded at world 3 with the HIP kernels in every rank (1000 chains: ragged shards of 334 / 333 / 333), the exchange on
    gloo: gathered samples bit-equal to the single-process run (plan="global"), statistics all-reduce equal to NumPy on that run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SHARDED_SMOKE_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(REPO, "tools", "sharded_smoke.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "world 3" in r.stdout and "bit-exact vs single process" in r.stdout and "summary_only over 1000 chains" in r.stdout


def _bench(extra_env, *launcher, extra=False):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, *launcher, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "1",
           "--no-ess", "--no-cpu-baseline"] + ([] if extra else ["--no-extra"])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rccl_branch_runs_on_a_gpu_under_torchrun():
    """bench.py's N > 1 code -- `nccl` process group, the library's sample buffer viewed in place as a CUDA tensor, the RCCL
    gather of the thinned samples inside the timed region, the max / sum reductions -- launched exactly as the driver
    launches it for N > 1 (a fresh child under `python -m torch.distributed.run`), on the one GPU a test box has
    (LOGREG_BENCH_FORCE_DIST=1 keeps the process group at world size 1).  Checked against the plain single-process run
    of the same command: same workload, the same acceptance rate to Monte-Carlo error (same seed and chain ids), throughput of the same order (wall clock on a possibly shared host)."""
    plain = _bench({})
    dist = _bench({"LOGREG_BENCH_FORCE_DIST": "1"}, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), extra=True)
    for d in (plain, dist):
        assert d["n_gpus"] == 1 and d["steps"] == 40 and d["scaling"] == "weak" and d["dtype"] == "f32"
        assert d["config"]["kernel_variant"] == {"mode": "reg", "group": 16, "rows_per_lane": 13}
        assert np.isfinite(d["value"]) and d["value"] > 1e8  # north_star: >= 1e8 gradient evaluations/s is 2e6 of these
        assert d["roofline"]["frac"] > 0.3 and d["roofline"]["kernel_ms"] > 0
        # HIP-event time of the fused launch (insensitive to what the host's other tenants do; 0.383 ms on every box so far): the bound that
        # does guard the kernel -- the wall-clock comparisons below are sanity factors only
        assert d["roofline"]["kernel_ms"] < 0.48 and d["roofline"]["frac"] > 0.40, d["roofline"]
    assert dist["config"]["parallelism"] == "chains sharded x1" and dist["gather_ms"] > 0 and plain["gather_ms"] == 0
    # (same seed and chain ids; the timed window starts after a TIME-bounded pre-warm, so the two runs may sit 20 launches apart in the
    #  chains' history -- equal to the Monte-Carlo error of 3.3e6 proposals, not to the bit: round 6 saw 0.92223 against 0.92232)
    assert abs(dist["accept_rate"] - plain["accept_rate"]) < 2e-3, (dist["accept_rate"], plain["accept_rate"])
    # 40 timed steps = 15 ms (round 5; with 5 steps = 2 ms the closing RCCL barrier alone, inside the timed region, decided the
    # comparison): the gather (5 MB device-to-device at N = 1) is inside the timed region of the distributed run
    # (wall clock with barriers on a possibly shared host: a sanity factor, not a performance claim -- measured 0.95 - 1.0 on a quiet box)
    assert dist["value"] > 0.6 * plain["value"] * (1 - dist["gather_ms"] / (40 * dist["ms_per_step"])), (plain["value"], dist["value"])
    # the self-check block of a multi-process line, produced on the hardware of this very run
    mg = dist["multi_gpu"]
    assert mg["ranks_seen"] == 1 and len(mg["devices"]) == 1 and mg["devices"][0].startswith("pci=") and "uuid=" in mg["devices"][0]
    assert mg["devices_distinct"] and 0 < mg["kernel_ms_min"] == mg["kernel_ms_max"]
    assert abs(mg["kernel_ms_max"] - dist["roofline"]["kernel_ms"]) < 1e-6
    # rank 0 re-ran 64 chains of the checked block (at one rank: its own) planned for 4096 chains; bit-identical to the gather
    assert mg["gather_bitexact"] is True and mg["gather_checked_chains"] == 64 and mg["gather_checked_rank"] == 0
    # BASELINE configs 3 and 5 through the multi-process code (8192 MALA chains / 1024 wide-model chains per rank)
    rows = {r["config"]: r for r in dist["extra"]["configs"]}
    assert set(rows) == {3, 5} and "scaled_down" not in rows[3]
    c3, c5 = rows[3], rows[5]
    assert c3["chains_total"] == 8192 and c3["with_gather"]["blocks_ok"] and c3["with_gather"]["gathered_bytes_per_rank"] == 4 * 8192 * 32
    assert c3["with_gather"]["chain_iterations_per_s"] > 1e9 and c3["summary_only"]["chain_iterations_per_s"] > 1e9  # (6.6e9 on a quiet box: wall clock)
    assert c3["summary_only"]["chains_counted"] == 8192 and c3["roofline"]["frac"] > 0.05
    assert c5["chains_total"] == 1024 and c5["blocks_ok"] and 0.6 < c5["accept_rate"] < 0.9
    # (wall clock around ~250 host-enqueued launches and the gather: a sanity bound -- on a box whose host cores were busy with other
    #  tenants' jobs this row measured 215 - 367 us per evaluation where it measures 9 - 11 on a quiet one; the performance figure of
    #  this workload is bench.py's HIP-event-timed extra.configs[5])
    assert c5["us_per_evaluation_all_chains_of_a_gpu"] < 2000 and c5["roofline"]["frac"] > 0
    # ... and the event-timed figure of config 3 with a bound that means something: ONE fused launch of 4000 iterations x 8192 chains (4.96 ms
    # at 6.6e9 chain-iterations/s) -- event time of a single launch does not see the host.  (Config 5's row is ~250 host-enqueued launches of
    # 9 us: its event time stretches with the host like its wall clock, so it keeps the sanity bound above.)
    assert c3["with_gather"]["kernel_ms_max"] < 7.0 and c3["summary_only"]["kernel_ms_max"] < 7.0, c3


def test_bench_with_two_ranks_sharing_the_gpu_over_gloo():
    """The N = 2 flow of bench.py with real kernels in BOTH ranks: two processes under torch.distributed.run on the one GPU of a test
    box, the exchange on gloo (RCCL refuses two ranks on one device).  Rank 1's chain block (chain_offset = C, its own pre-warm count),
    the gather, and the line's self-check: rank 0 re-runs 64 chains of RANK 1's block and finds them bit-equal; the line says the two
    ranks shared a device."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
           "--warmup", "2", "--no-ess", "--no-cpu-baseline", "--scale", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    mg = d["multi_gpu"]
    assert d["n_gpus"] == 2 and mg["ranks_seen"] == 2 and mg["devices_distinct"] is False
    assert mg["gather_checked_rank"] == 1 and mg["gather_bitexact"] is True
    assert d["config"]["kernel_variant"] == {"mode": "reg", "group": 16, "rows_per_lane": 13}
    rows = {row["config"]: row for row in d["extra"]["configs"]}
    assert rows[3]["n_gpus"] == 2 and rows[5]["n_gpus"] == 2 and rows[3]["scaled_down"] == 8


def test_c_level_exchange_over_rccl_at_one_rank():
    """include/logreg_hip.h lr_comm_* / lr_gather / lr_allreduce_sum_f64 on the real library: RCCL loaded at first use (the library has no
    link-time dependency on it), a communicator of one rank on the box's GPU, the gather of a device block to the root and the in-place
    float64 sum -- in a child process, so that the ROCm RCCL this entry loads does not meet the copy torch carries in another test's
    process.  (More ranks need more GPUs: the multi-GPU bench is the driver's.)"""
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import logreg_amd as la
from logreg_amd import _lib
L = _lib.load()
ident = (C.c_ubyte * 128)()
_lib.check(L.lr_comm_unique_id(ident))
assert any(ident)
comm = C.c_void_p()
_lib.check(L.lr_comm_create(ident, 0, 1, 0, C.byref(comm)))
src = la.DeviceArray(0, (4, 96, 8), np.float32)
dst = la.DeviceArray(0, (1, 4, 96, 8), np.float32)
host = np.random.default_rng(1).standard_normal((4, 96, 8)).astype(np.float32)
_lib.check(L.lr_memcpy_h2d(0, src.ptr, host.ctypes.data, host.nbytes, None))
st = C.c_void_p(); _lib.check(L.lr_stream_create(0, C.byref(st)))
_lib.check(L.lr_gather(comm, src.ptr, dst.ptr, host.nbytes, 0, st))
sums = la.DeviceArray(0, (57,), np.float64)
hs = np.linspace(1, 2, 57)
_lib.check(L.lr_memcpy_h2d(0, sums.ptr, hs.ctypes.data, hs.nbytes, None))
_lib.check(L.lr_allreduce_sum_f64(comm, sums.ptr, 57, st))
_lib.check(L.lr_stream_sync(0, st))
assert np.array_equal(dst.to_host()[0], host), "gather"
assert np.array_equal(sums.to_host(), hs), "allreduce at world 1"
assert L.lr_gather(comm, src.ptr, dst.ptr, host.nbytes, 3, st) < 0
_lib.check(L.lr_comm_destroy(comm))
print("exchange ok")
""" % REPO
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "exchange ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
