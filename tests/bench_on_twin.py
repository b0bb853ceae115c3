"""TEST INFRASTRUCTURE: bench.py's real main() -- ChainSets, Exchange, self_check, dist_configs -- on the CPU test double of
the C ABI (tests/host/lr_cpu_twin.c, injected by tests/twin.py) over the gloo backend, launched under torch.distributed.run
by tests/test_host_logic.py.  The product has no switch that loads the twin: the injection lives here, under tests/.

    python -m torch.distributed.run --nproc-per-node 2 ... tests/bench_on_twin.py --gpus 2 --chains 96 --steps 3 ...
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("OMP_NUM_THREADS", "1")

import twin  # noqa: E402

twin.install()
import bench  # noqa: E402
import torch  # noqa: E402


def host_view(self, samples):
    """The twin's "device" memory is host memory: view a DeviceArray in place as a CPU tensor (the GPU run views it as a
    CUDA tensor through __cuda_array_interface__)."""
    if isinstance(samples, torch.Tensor):
        return samples
    raw = (ctypes.c_byte * samples.nbytes).from_address(samples.ptr)
    return torch.from_numpy(np.frombuffer(raw, dtype=samples.dtype).reshape(samples.shape))


bench.Exchange.tensor = host_view
sys.exit(bench.main(sys.argv[1:] + ["--backend", "gloo", "--no-ess", "--no-cpu-baseline"]) or 0)
