import os
import sys
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """the C oracle; built on demand here (CPU box), prebuilt on the GPU box."""
    so = os.path.join(ROOT, "oracle", "libescat_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import chain_oracle
    return chain_oracle.lib()


@pytest.fixture(scope="session")
def sg():
    return np.load(os.path.join(GOLDEN, "scatter_goldens.npz"))


@pytest.fixture(scope="session")
def bg():
    return np.load(os.path.join(GOLDEN, "builder_goldens.npz"))


@pytest.fixture(scope="session")
def r3g():
    """round-3 goldens: the reference's vector track on a batch of column_dot graphs, fp32 and fp16-stored blocks"""
    return np.load(os.path.join(GOLDEN, "round3_goldens.npz"))


@pytest.fixture(scope="session")
def ig():
    return np.load(os.path.join(GOLDEN, "infogain_goldens.npz"))
