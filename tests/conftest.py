import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name))

        def json(self, name):
            import json
            return json.load(open(os.path.join(GOLDEN, name)))
    return G()


@pytest.fixture(scope="session")
def mips():
    from topsy_amd import kernel_lut
    return kernel_lut.kernel_mips()


def make_cloud(n, seed=0, h_scale=1.0):
    """Seeded test cloud with a wide range of smoothing lengths (sub-pixel ... several hundred px)."""
    rs = np.random.RandomState(seed)
    pos = (rs.normal(size=(n, 3)) * np.array([30.0, 20.0, 40.0])).astype(np.float32)
    h = (np.exp(rs.uniform(np.log(0.02), np.log(60.0), size=n)) * h_scale).astype(np.float32)
    m = rs.uniform(0.5, 2.0, size=n).astype(np.float32)
    q = rs.normal(size=n).astype(np.float32)
    rgb = rs.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    return pos, h, m, q, rgb
