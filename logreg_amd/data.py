"""Data block of the reference scripts (Python/fit-np-hmc.py:12-19): read the Pima training set,
build X with an intercept column prepended and y in {0,1}.

`load_pima()` reads the reference's own `pima.data` text format (headerless, space-separated,
7 numbers + Yes|No per line -- the format C/fit-bayes.c:54-67 parses); `load_pima_parquet()`
reads `pima.parquet` exactly like the Python scripts do.  A copy of the 200 rows (R's public
MASS::Pima.tr, written by the reference's R/create-dataset.R:6-12) ships with the package so
the GPU box needs neither the reference checkout nor pandas.
"""
from __future__ import annotations

import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PIMA_TXT = os.path.join(_HERE, "data", "Pima.tr.txt")
COLUMNS = ["npreg", "glu", "bp", "skin", "bmi", "ped", "age"]


def load_pima(path: str | None = None):
    """-> (X [n, 8] float64 with intercept column, y [n] float64 in {0,1})."""
    path = path or PIMA_TXT
    rows, ys = [], []
    with open(path) as f:
        for ln, line in enumerate(f):
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if len(tok) != 8 or tok[7] not in ("Yes", "No"):
                raise ValueError(f"{path}:{ln + 1}: expected 7 numbers and Yes|No, got {line!r}")
            rows.append([float(t) for t in tok[:7]])
            ys.append(1.0 if tok[7] == "Yes" else 0.0)
    X = np.asarray(rows, dtype=np.float64)
    X = np.hstack((np.ones((X.shape[0], 1)), X))
    return X, np.asarray(ys, dtype=np.float64)


def load_pima_parquet(path: str):
    """Same data block as the reference scripts, from pima.parquet (needs pandas + pyarrow)."""
    import pandas as pd
    df = pd.read_parquet(path)
    n = df.shape[0]
    y = pd.get_dummies(df["type"])["Yes"].to_numpy(dtype="float32").astype(np.float64)
    X = df.drop(columns="type").to_numpy()
    X = np.hstack((np.ones((n, 1)), X))
    return X, y


def synthetic_logreg(n: int, p: int, seed: int = 20240001, beta_sd: float = 0.5):
    """Synthetic design of BASELINE.json's throughput configs:
    X[:,0]=1, X[:,1:]~N(0,1), beta*~N(0,beta_sd^2), y~Bernoulli(sigma(X beta*))."""
    rng = np.random.Generator(np.random.Philox(seed))
    X = rng.standard_normal((n, p))
    X[:, 0] = 1.0
    beta = rng.standard_normal(p) * beta_sd
    y = (rng.random(n) < 1.0 / (1.0 + np.exp(-X @ beta))).astype(np.float64)
    return X, y, beta
