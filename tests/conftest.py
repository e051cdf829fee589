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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def ref_literals():
    return load_golden("ref_test_literals.npz")


@pytest.fixture(scope="session")
def ref_leaf():
    return load_golden("ref_leaf.npz")


@pytest.fixture(scope="session")
def ref_hierarchical():
    return load_golden("ref_hierarchical.npz")


@pytest.fixture(scope="session")
def ref_slavcheva():
    return load_golden("ref_slavcheva.npz")


@pytest.fixture(scope="session")
def ref_config1():
    return load_golden("ref_config1.npz")


@pytest.fixture(scope="session")
def ref_tsdf():
    return load_golden("ref_tsdf.npz")


@pytest.fixture(scope="session")
def ref_ewa():
    return load_golden("ref_ewa.npz")
