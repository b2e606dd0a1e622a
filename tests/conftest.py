import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(params=["plain", "batch_affine"])
def fixed_msm_mode(request, monkeypatch):
    """Both variants of the fixed-generator MSM: the plain XYZZ kernel and the batch-affine pairing
    (csrc/batch_affine.hpp), forced regardless of batch size.  Read by bpp_verifier_create."""
    monkeypatch.setenv("BPP_AMD_BATCH_AFFINE", "0" if request.param == "plain" else "2")
    return request.param
