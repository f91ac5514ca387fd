import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    g = os.path.join(HERE, "golden")
    return {
        "k2": np.load(os.path.join(g, "kernels2d.npz")),
        "k3": np.load(os.path.join(g, "kernels3d.npz")),
        "sweeps": np.load(os.path.join(g, "sweeps.npz")),
        "solves": json.load(open(os.path.join(g, "solves.json"))),
    }
