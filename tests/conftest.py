import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session", autouse=True)
def _blas_threads_within_quota():
    """numpy's BLAS starts one thread per VISIBLE CPU (256 on the bench host) whatever the container's CPU quota
    (16 there): an oracle matmul then time-shares 16 CPUs between 256 threads, and a process pool of such oracles is
    slower than one process.  Keep BLAS inside the quota for the whole session."""
    try:
        from threadpoolctl import threadpool_limits
        from acoss_amd.utils import effective_cpus
    except ImportError:
        yield
        return
    with threadpool_limits(limits=effective_cpus()):
        yield
