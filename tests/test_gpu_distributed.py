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


def _bench(extra_env, *launcher):
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, *launcher, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1", "--no-extra",
           "--no-ess", "--no-cpu-baseline"]
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
    of the same command: same workload, same acceptance rate (same seed and chain ids), throughput within 15 %."""
    plain = _bench({})
    dist = _bench({"LOGREG_BENCH_FORCE_DIST": "1"}, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port()))
    for d in (plain, dist):
        assert d["n_gpus"] == 1 and d["steps"] == 5 and d["scaling"] == "weak" and d["dtype"] == "f32"
        assert d["config"]["kernel_variant"] == {"mode": "reg", "group": 16, "rows_per_lane": 13}
        assert np.isfinite(d["value"]) and d["value"] > 1e8  # north_star: >= 1e8 gradient evaluations/s is 2e6 of these
        assert d["roofline"]["frac"] > 0.3 and d["roofline"]["kernel_ms"] > 0
    assert dist["config"]["parallelism"] == "chains sharded x1" and dist["gather_ms"] > 0 and plain["gather_ms"] == 0
    assert dist["accept_rate"] == plain["accept_rate"]
    # 5 timed steps = 2 ms: the gather (0.66 MB device-to-device at N = 1) is inside the timed region of the distributed run
    assert dist["value"] > 0.85 * plain["value"] * (1 - dist["gather_ms"] / (5 * dist["ms_per_step"])), (plain["value"], dist["value"])
