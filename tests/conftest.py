import os
import sys

import numpy as np
import pytest

# guard bands behind every device workspace, verified whenever a separator syncs with the host
os.environ.setdefault("SSSPY_AMD_WS_CANARY", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a committed golden fixture (tests/golden/<name>.npz) into a dict."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as data:
        return {k: data[k] for k in data.files}


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    denom = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (denom if denom > 0 else 1.0)


def rel_err_up_to_phase(a, b, name):
    """Relative error after removing a unit-modulus factor per (bin, source): the pairwise updates
    (IP2 / ISS2) inherit the arbitrary phase of 2x2 eigenvectors until scale restoration
    ("eigenvector phase ambiguity").  `name` is "demix_filter" (.., F, N, N:
    one phase per row) or "output" (.., N, F, T: one phase per source and bin)."""
    a, b = np.asarray(a), np.asarray(b)
    inner = np.sum(a * b.conj(), axis=-1, keepdims=True)
    mag = np.abs(inner)
    phase = np.where(mag > 0, inner / np.where(mag > 0, mag, 1), 1)
    return rel_err(a * phase.conj(), b)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def option(value):
    """A golden's meta entry that is a bool in most fixtures and a keyword (e.g. "projection_back")
    in some."""
    v = np.asarray(value)
    return str(v) if v.dtype.kind in "US" else bool(v)
