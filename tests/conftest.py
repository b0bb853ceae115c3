import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pima():
    d = load_golden("pima_xy.json")
    return np.array(d["X"]), np.array(d["y"])


@pytest.fixture(scope="session")
def pscale():
    return np.array([10.0, 1, 1, 1, 1, 1, 1, 1])


@pytest.fixture(scope="session")
def oracle_model(pima, pscale):
    from oracle.oracle import OracleModel
    X, y = pima
    return OracleModel(X, y, pscale)


@pytest.fixture(scope="session")
def map_beta():
    return np.array(load_golden("map.json")["map"])
