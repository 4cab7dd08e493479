import os
import sys
import glob
import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def rel_err(a, b):
    """max|a-b| / max|b| (the fp64 parity measure; tolerance per north_star is 1e-12)."""
    a = np.asarray(a)
    b = np.asarray(b)
    s = np.abs(b).max()
    return float(np.abs(a - b).max() / (s if s > 0 else 1.0))


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
