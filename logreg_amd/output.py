"""Output step of the reference scripts (Python/fit-np-hmc.py:110-117): the sample matrix as a
DataFrame with columns b0..b{p-1} written to parquet, and the `scipy.stats.describe` summary.
Many-chain output [iters, C, p] is written chain-major with an extra `chain` column.
"""
from __future__ import annotations

import numpy as np

from .diagnostics import describe


def column_names(p: int):
    return [f"b{j}" for j in range(p)]


def to_frame(out):
    """`out` [iters, p] -> DataFrame b0..; [iters, C, p] -> DataFrame chain, draw, b0.. (needs pandas)."""
    import pandas as pd
    out = np.asarray(out)
    if out.ndim == 2:
        return pd.DataFrame(out, columns=column_names(out.shape[1]))
    iters, C, p = out.shape
    flat = np.transpose(out, (1, 0, 2)).reshape(C * iters, p)
    df = pd.DataFrame(flat, columns=column_names(p))
    df.insert(0, "draw", np.tile(np.arange(iters), C))
    df.insert(0, "chain", np.repeat(np.arange(C), iters))
    return df


def write_parquet(out, path: str):
    """odf = pd.DataFrame(out, columns=[b0..]); odf.to_parquet(path)   (fit-np-hmc.py:111-112)."""
    to_frame(out).to_parquet(path)
    return path


def read_parquet(path: str):
    import pandas as pd
    df = pd.read_parquet(path)
    cols = [c for c in df.columns if c.startswith("b")]
    if "chain" in df.columns:
        C = int(df["chain"].max()) + 1
        iters = len(df) // C
        return np.transpose(df[cols].to_numpy().reshape(C, iters, len(cols)), (1, 0, 2))
    return df[cols].to_numpy()


def print_summary(out):
    """The reference's closing prints (fit-np-hmc.py:113-117)."""
    d = describe(out)
    print("Posterior summaries:")
    print(d)
    print("\nMean: " + str(d["mean"]))
    print("Variance: " + str(d["variance"]))
    return d
