"""The real sharded path -- logreg_amd.distributed.mcmc_sharded on the HIP kernels with torch.distributed's RCCL
backend -- launched the way the driver launches bench.py: a fresh child process under torch.distributed.run.
One GPU is all a test box has, so N = 1; the N > 1 logic (ragged/empty shards, global chain ids, chunked launches,
the statistics all-reduce) is covered by the world-size-2 gloo tests in test_distributed_gloo.py."""
import os
import socket
import subprocess
import sys

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
