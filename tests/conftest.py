import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref), or None where it did not travel."""
    from oracle.oracle import load_ref
    return load_ref()


@pytest.fixture(scope="session")
def checker(ref, oracle):
    """What the differential GPU tests compare with: the compiled, unmodified reference where it travelled
    (oracle/_ref/libedlib_ref.so rides along to the GPU box), the C99 restatement otherwise."""
    return ref if ref is not None else oracle


@pytest.fixture(scope="session")
def engine():
    """The product library; on the GPU box it must load and see a device (no fallback)."""
    import edlib_amd
    assert edlib_amd.device_count() >= 1, "no HIP device visible: " + edlib_amd.last_error()
    return edlib_amd
