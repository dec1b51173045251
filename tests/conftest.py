import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_native():
    """Build the product library and the oracle if they are missing (make is incremental)."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "hnsw_rs_amd", "libhnsw_mi355x.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hnsw_rs_amd", "csrc"), "-s", "-j4"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])


_build_native()

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def testdata():
    """The reference's own test-data (GloVe-50d, 1000 store rows + 100 queries) as binary fixtures."""
    store = np.load(os.path.join(GOLDEN, "testdata_store.npy"))
    queries = np.load(os.path.join(GOLDEN, "testdata_queries.npy"))
    assert store.shape == (1000, 50) and queries.shape == (100, 50)
    return store, queries


@pytest.fixture(scope="session")
def gpu_available():
    import hnsw_rs_amd as H
    return H.device_count() > 0
