import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own headers compiled in place; container only."""
    from oracle_lib import REF_SO, Reference
    if not os.path.isdir("/root/reference") and not os.path.exists(REF_SO):
        pytest.skip("reference tree not available on this machine")
    return Reference()


@pytest.fixture(scope="session")
def vectors():
    return json.load(open(os.path.join(GOLDEN, "vectors.json")))


@pytest.fixture(scope="session")
def refdata():
    return json.load(open(os.path.join(GOLDEN, "reference.json")))


@pytest.fixture(scope="session")
def patterns():
    return np.load(os.path.join(GOLDEN, "patterns.npz"))


@pytest.fixture(scope="session")
def full_patterns():
    """WHOLE bundled test images as NV12 (reference encoder) + sha256 of the reference headers' decode."""
    meta = json.load(open(os.path.join(GOLDEN, "patterns_full.json")))
    return meta["images"], np.load(os.path.join(GOLDEN, "patterns_full.npz"))


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def bt709():
    """The product: C-ABI library through the Python host mirror.  Fails loudly
    (no CPU fallback) if the HIP library is missing."""
    import metalbt709decoder_amd as pkg
    return pkg


@pytest.fixture(scope="session")
def pass2():
    """Decode + pass 2 computed by the reference's own inlines (tests/golden/make_golden.py)."""
    return json.load(open(os.path.join(GOLDEN, "pass2.json")))
