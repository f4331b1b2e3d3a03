import os
import sys

import pytest

# the oracle's OpenMP regions: a 256-thread GPU-box host gains nothing beyond a few threads here
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
