import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "di-hpc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never silently pass on a box without a GPU."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def rel_err(ref, got):
    """max |ref-got| / max(1, |ref|): the tolerance form SURVEY.md 7.5 prescribes (north_star: <=1e-5 rel)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    assert ref.shape == got.shape, (ref.shape, got.shape)
    if ref.size == 0:
        return 0.0
    return float(np.max(np.abs(ref - got) / np.maximum(1.0, np.abs(ref))))
