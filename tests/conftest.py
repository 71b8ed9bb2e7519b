import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The GPU box has 256 hardware threads: torch-CPU / OpenMP with that many threads on these small tensors
# is pathologically slow (the oracle took minutes).  Cap the CPU side of the tests.
_NT = str(min(16, os.cpu_count() or 1))
os.environ.setdefault("OMP_NUM_THREADS", _NT)
os.environ.setdefault("MKL_NUM_THREADS", _NT)
try:
    import torch
    torch.set_num_threads(int(_NT))
except Exception:  # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
