import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "bench: timing prints without assertions on speed (run by hand with -m bench on a GPU box; not part of -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen, hence importlib)."""
    return importlib.import_module("endoscopydepthestimation-pytorch_amd")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load
