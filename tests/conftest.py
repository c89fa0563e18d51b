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
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def tables():
    return dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))


def kabsch_free_rmsd(a, b):
    """Plain (no superposition) RMSD over the non-zero backbone atoms of [..., 37, 3] arrays."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = ((a[..., :5, :] - b[..., :5, :]) ** 2).sum(-1)
    return float(np.sqrt(d.mean()))
