"""logreg_amd -- many-chain MCMC for Bayesian logistic regression on AMD MI355X (gfx950).

Drop-in for the hot path of darrenjw/logreg's NumPy scripts (Python/fit-numpy.py,
fit-np-mala.py, fit-np-hmc.py, fit-np-ul.py): the same `ll / lprior / lpost / glp` closures and
`mhKernel / malaKernel / hmcKernel / ulKernel / mcmc` constructors, executed by hand-written HIP
kernels through the C ABI in include/logreg_hip.h.  There is no CPU fallback.

    from logreg_amd import LogReg, hmcKernel, mcmc, load_pima
    X, y = load_pima()
    model = LogReg(X, y, pscale=[10, 1, 1, 1, 1, 1, 1, 1])
    kern = hmcKernel(model.lpost, model.glp, eps=1e-3, l=50, dmm=1 / pre)
    out = mcmc(init, kern, thin=20)                 # [10000, 8], like the reference
    out = mcmc(np.tile(init, (4096, 1)), kern, thin=20, iters=1000)   # [1000, 4096, 8]
"""
from ._lib import LogregHipError, device_count  # noqa: F401
from .data import load_pima, load_pima_parquet, synthetic_logreg  # noqa: F401
from .diagnostics import describe, ess_geyer, ess_per_param, ess_pooled, split_rhat, summarise  # noqa: F401
from .kernels import (ChainSet, FusedKernel, hmcKernel, malaKernel, mcmc, mhKernel, rwProposal,  # noqa: F401
                      ulKernel)
from .model import DeviceArray, LogReg  # noqa: F401
from .optimize import find_map, overdispersed_init  # noqa: F401
from .output import print_summary, read_parquet, to_frame, write_parquet  # noqa: F401

__all__ = ["LogReg", "DeviceArray", "ChainSet", "FusedKernel", "mhKernel", "malaKernel", "hmcKernel", "ulKernel",
           "rwProposal", "mcmc", "load_pima", "load_pima_parquet", "synthetic_logreg", "summarise", "describe",
           "ess_geyer", "ess_per_param", "ess_pooled", "split_rhat", "device_count", "LogregHipError", "find_map", "overdispersed_init", "write_parquet",
           "read_parquet", "to_frame", "print_summary"]
