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


# ---------------------------------------------------------------------------------------------------------------------
# Gradients are compared RELATIVE TO THE TENSOR'S OWN SCALE (VERDICT r02 "weak" #1): gradients of mean-reduced losses are
# O(1/(T*B)) or O(1/B), so the max(1,|ref|) denominator of rel_err above would accept errors of several percent of the
# tensor's maximum (and an all-zero gradient at T=256,B=16384).  The reference's own tests check gradients relatively
# (tests/testbase.py:8-11 mean relative error, tests/test_vtrace.py:55-60).
_PROBE = []          # (test id, tensor name, err, scale): dumped to gpurun_out/parity_probe.json at session end


def grad_err(ref, got, name=""):
    """max |ref - got| / max |ref| plus a magnitude assert.  A reference gradient that is identically zero (e.g. a
    clipped-away sample set) demands an identically zero result."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    assert ref.shape == got.shape, (ref.shape, got.shape)
    if ref.size == 0:
        return 0.0
    assert np.isfinite(got).all(), "gradient is not finite"
    scale = float(np.max(np.abs(ref)))
    if scale == 0.0:
        assert not got.any(), "reference gradient is identically zero, result is not"
        return 0.0
    gmax = float(np.max(np.abs(got)))
    assert 0.5 * scale < gmax < 2.0 * scale, f"gradient has the wrong magnitude: max|got| {gmax:.3e} vs max|ref| {scale:.3e}"
    err = float(np.max(np.abs(ref - got))) / scale
    _PROBE.append((os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], name, err, scale))
    if os.environ.get("HPC_RLL_GRAD_PROBE_ONLY") == "1":   # measurement runs (tests/tools): record every case, assert nothing
        return 0.0
    return err


def pytest_sessionfinish(session, exitstatus):
    import torch
    # only sessions that ran tests on a GPU write the record (a CPU-tier run used to overwrite the GPU session's file)
    if not _PROBE or not torch.cuda.is_available():
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    worst = {}
    for tid, name, err, scale in _PROBE:
        key = tid + ("::" + name if name else "")
        rec = worst.setdefault(key, {"n": 0, "max_err": 0.0, "min_scale": scale, "max_scale": scale})
        rec["n"] += 1
        rec["max_err"] = max(rec["max_err"], err)
        rec["min_scale"] = min(rec["min_scale"], scale)
        rec["max_scale"] = max(rec["max_scale"], scale)
    with open(os.path.join(out, "parity_probe.json"), "w") as f:
        json.dump({"what": "max|d| / max|ref| of every gradient assert of this pytest session (tests/conftest.py: grad_err)",
                   "asserts": len(_PROBE), "worst_overall": max(e for _, _, e, _ in _PROBE), "per_test": worst}, f, indent=1)
