import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return _load


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    _o.build()
    _o.load()
    return _o
