"""pytest configuration: the ``gpu`` marker and shared fixture helpers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The library honours its test / developer switches (every GMMVB_* but GMMVB_ESTEP_PRUNE and GMMVB_MSTEP_SPARSE) only under
# GMMVB_DEBUG (csrc/workspace.h: dev_env): the suite's variants are such switches.
os.environ.setdefault("GMMVB_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    """The parity metric of BASELINE.md section 3: max|a-b| / max|b| per array."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def mat_functionals(a):
    """What tests/golden/make_golden_large.py stores of a [K, D, D] array instead of the array itself: the first two
    matrices, every diagonal, three fixed (seeded) projections and the log-determinants."""
    a = np.asarray(a, dtype=float)
    v = np.random.default_rng(7).standard_normal((a.shape[-1], 3))
    return dict(head=a[:2], diag=np.diagonal(a, axis1=1, axis2=2), proj=a @ v, logabsdet=np.linalg.slogdet(a)[1])
