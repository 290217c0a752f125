import os
import sys

# (one BLAS thread: numpy's 64 spinning OpenBLAS workers exhaust the test hosts' cgroup CPU quota, and a throttled process sees
# random stalls of tens of milliseconds -- bench.py does the same)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds (if needed) and loads the fp64 C++ CPU oracle."""
    from oracle import binding

    binding.build()
    return binding
